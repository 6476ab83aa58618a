#!/bin/bash
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/tp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/scripts/scan_probe.py 10000000 4096 ${1:-32} > $OUT/o.log 2> $OUT/e.log
cd $REPO
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    if r["Name"].startswith(("k_", "void k_", "__amd")):
        print(f"{r['Name'][:50]:50s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f}")
PY
rm -rf $OUT/trace
