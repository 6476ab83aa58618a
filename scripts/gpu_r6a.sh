#!/bin/bash
# coarse step: prefiltered top-k (qk_dense_pf.hip) per-kernel times, one size at a time
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp
for nl in 4096 65536; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r6a/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6a/trace -- python $GRAFT_REPO_ROOT/scripts/coarse_probe.py $nl 32 2> /dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r6a/trace -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:7]:
    print($nl, r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg")
PY
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r6a/trace
