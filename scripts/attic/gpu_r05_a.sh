#!/bin/bash
# round 5, visit A: workers / group tests, the group's orchestration cost, the default bench with the new blocks, one-process N = 2 run
mkdir -p gpurun_out
python -m pytest tests/test_workers_gpu.py tests/test_group_gpu.py -x -q 2>&1 | tail -3
python scripts/group_probe.py > gpurun_out/r05_group_probe.jsonl 2> gpurun_out/group_probe.err; tail -5 gpurun_out/r05_group_probe.jsonl; tail -3 gpurun_out/group_probe.err
timeout 1500 python bench.py --steps 20 --warmup 3 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"; tail -25 gpurun_out/r05a_bench.err
timeout 600 python bench.py --gpus 2 --single-process --steps 20 --warmup 3 --nvec-sharded 2000000 --nlist-sharded 1024 > gpurun_out/r05a_bench_group2.json 2> gpurun_out/r05a_bench_group2.err; echo "group bench rc=$?"; tail -5 gpurun_out/r05a_bench_group2.err
