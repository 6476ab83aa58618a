#!/bin/bash
# rocprofv3 per-kernel averages of the recall-target search (scripts/aps_probe.py, one target)
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/aps_stats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
APS_NO_CPU=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $REPO/scripts/aps_probe.py 10000000 4096 0.9 > $OUT/stdout.log 2> $OUT/stderr.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Name"].startswith(("k_", "void k_"))]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"{r['Name'][:44]:44s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
tail -1 $OUT/stdout.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -delete
