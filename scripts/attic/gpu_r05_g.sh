#!/bin/bash
# round 5, visit G: k_assign_pf without in-loop spills (parity + timing), phase stamps of the small-batch search, dynamic test rerun
mkdir -p gpurun_out
python -m pytest tests/test_assign_pf_gpu.py tests/test_kmeans_gpu.py tests/test_sharded_maintenance_gpu.py -x -q 2>&1 | tail -3
python scripts/stress_assign_pf.py 40 7 2>&1 | tail -2
python scripts/kmeans_probe.py 2>/dev/null | tee gpurun_out/r05g_kmeans_probe.jsonl
python scripts/assign_probe.py 2>/dev/null | tee gpurun_out/r05g_assign_probe.jsonl
QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_smallprobe.so QK_SMALL_CLOCK=1 python scripts/small_clock.py 2>&1 | grep k_search_small | tail -8 | tee gpurun_out/r05g_small_clock.txt
(time python -m pytest tests/test_dynamic_workload_10m_gpu.py -x -q) 2>&1 | tail -8
