#!/bin/bash
# rocprofv3 per-kernel averages of bench.py on BASELINE.json configs[2] (10M x 768, IP, k = 100)
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/c3_stats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $REPO/bench.py --dim 768 --metric ip --k 100 --no-cpu > $OUT/stdout.log 2> $OUT/stderr.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(t in n for t in ("k_scan", "k_merge", "k_group", "k_seed", "k_dense", "k_select", "k_prep", "k_argmin", "k_merge_wide")):
        print(f"{n[:44]:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f}")
PY
find $OUT -name "*kernel_trace.csv" -delete
