#!/bin/bash
# round 5, visit H: selection by extraction in the one-launch small-batch search: parity, phase stamps, latency
mkdir -p gpurun_out
python -m pytest tests/test_scan_gpu.py tests/test_random_shapes_gpu.py tests/test_bench_parity_gpu.py tests/test_index_gpu.py tests/test_scan_form_selection_gpu.py -x -q 2>&1 | tail -4
QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_smallprobe.so QK_SMALL_CLOCK=1 python scripts/small_clock.py 2>&1 | grep k_search_small | tail -6 | tee gpurun_out/r05h_small_clock.txt
LAT_NO_CPU=1 python scripts/latency_probe.py 2>/dev/null | tee gpurun_out/r05h_latency_probe.json
