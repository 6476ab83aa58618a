#!/bin/bash
# round-2: product build -- rocprofv3 passes (headline + hard), default bench, 2-rank functional run of the configs[3] path over gloo
O=gpurun_out/r2l; mkdir -p $O
(time python bench.py) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
bash scripts/gpu_profile.sh r2l/prof_headline --no-extra --steps 100 > $O/prof_headline.log 2>&1
bash scripts/gpu_profile.sh r2l/prof_hard --no-extra --steps 100 --manifold 10 > $O/prof_hard.log 2>&1
grep -E "k_scan|k_merge|k_dense|k_seed|k_group|k_prep" $O/prof_headline/summary.txt | head -12
grep -E "k_scan|k_merge" $O/prof_hard/summary.txt | head -8
QUAKE_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --nvec-sharded 2000000 --nlist-sharded 1024 --batch-sharded 256 --steps 20 --warmup 3 --settle 5 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -5 $O/bench_2rank_gloo.err; cat $O/bench_2rank_gloo.json | head -c 1500
