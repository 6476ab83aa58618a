#!/bin/bash
# round 5, visit B: fused prep in the nearest-centroid kernel: parity suites + headline
mkdir -p gpurun_out
python -m pytest tests/test_scan_gpu.py tests/test_bench_parity_gpu.py tests/test_random_shapes_gpu.py tests/test_index_gpu.py tests/test_group_gpu.py tests/test_workers_gpu.py tests/test_kmeans_gpu.py -x -q 2>&1 | tail -5
timeout 900 python bench.py --steps 20 --warmup 3 --no-extra --no-pmc > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err; echo "bench rc=$?"; grep -E "phases|PARITY" gpurun_out/r05b_bench.err | tail -3
python -c "
import json; d=json.load(open('gpurun_out/r05b_bench.json')); print(d['value'], d['ms_per_step'], d['phases_ms'], d['roofline']['frac'], d['timed_groups']['min'], d['timed_groups']['max'])"
