#!/bin/bash
# round 5, final build: the driver's sequence on one box (whole -m gpu suite, smoke(), default bench line) + the recall-target probe
# at three candidate fractions + the seeded stress of the recall-target search
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
M=gpurun_out/r5f
(time python -m pytest tests -m gpu -x -q) > $M/r05_pytest_gpu.log 2>&1; tail -4 $M/r05_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python bench.py > $M/r05_bench.json 2> $M/bench.err); echo "bench rc=$?"; tail -2 $M/bench.err
for f in 0.02 0.05 0.2; do APS_FRACTION=$f python scripts/aps_probe.py 10000000 4096 0.8 0.9 0.99 2>/dev/null | grep "^{" | sed "s/^{/{\"initial_search_fraction\": $f, /"; done > $M/r05_aps_probe.jsonl
cut -c1-260 $M/r05_aps_probe.jsonl
timeout 900 python scripts/stress_aps.py 1500 5000 2>/dev/null | tail -1 > $M/r05_stress_aps.json; cat $M/r05_stress_aps.json
python scripts/coarse_probe.py 4096,16384,65536 1,32,100,204,400 2>/dev/null > $M/r05_coarse_probe.jsonl
bash scripts/gpu_shape_sweep.sh 2>/dev/null | grep "^{" > $M/r05_shape_sweep.jsonl
