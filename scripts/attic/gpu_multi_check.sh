#!/bin/bash
# functional check of bench.py's N>1 path on ONE GPU: two ranks share cuda:0 over the gloo backend
cd $GRAFT_REPO_ROOT
QUAKE_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 5 --warmup 2 --nvec 2000000 --nlist 1024 --no-cpu 2>&1 | grep -v amdgpu.ids | tail -15
