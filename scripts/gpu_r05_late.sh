#!/bin/bash
# round 5, after the recall-target / store changes: smoke, the default bench line, the recall-target probe, the hot dynamic workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
M=gpurun_out/r5q
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python bench.py > $M/r05_bench.json 2> $M/bench.err); echo "bench rc=$?"; tail -2 $M/bench.err
python scripts/aps_probe.py 10000000 4096 0.8 0.9 0.99 > $M/r05_aps_probe.jsonl 2>/dev/null; cat $M/r05_aps_probe.jsonl | cut -c1-200
python scripts/dynamic_workload.py 10000000 128 60 hot > $M/r05_dynamic_workload_hot_10M.json 2> $M/dyn10.err
timeout 2400 python scripts/dynamic_workload.py 50000000 128 60 hot > $M/r05_dynamic_workload_hot_50M.json 2> $M/dyn50.err
ls -la $M
