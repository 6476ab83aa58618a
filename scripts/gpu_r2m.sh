#!/bin/bash
O=gpurun_out/r2m; mkdir -p $O
(timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_bench_parity_gpu.py -m gpu -x -q -k "small or configs0") > $O/pytest_small.log 2>&1; tail -3 $O/pytest_small.log
LAT_NO_CPU=1 python scripts/latency_probe.py > $O/latency.log 2>&1; tail -2 $O/latency.log
QUAKE_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --nvec-sharded 2000000 --nlist-sharded 1024 --batch-sharded 256 --steps 20 --warmup 3 --settle 5 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; grep -E "bench\]|rror|fault" $O/bench_2rank_gloo.err | tail -8; cat $O/bench_2rank_gloo.json | head -c 1800
