#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench.  Usage: bash scripts/gpu_round.sh [tag]
cd $GRAFT_REPO_ROOT
TAG=${1:-run}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > gpurun_out/${TAG}_smoke.log
cat gpurun_out/${TAG}_smoke.log
( timeout 600 python bench.py --nvec 1000000 --nlist 1024 --steps 10 2> gpurun_out/${TAG}_bench_small.err | tail -3 ) > gpurun_out/${TAG}_bench_small.json
tail -12 gpurun_out/${TAG}_bench_small.err; cat gpurun_out/${TAG}_bench_small.json
( timeout 1200 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -3 ) > gpurun_out/${TAG}_bench.json
tail -14 gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.json
