#!/bin/bash
# rocprofv3 kernel stats of the bench for two settings of the dynamic-tail parameters (per-kernel averages)
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "0 8" "20 16"; do
  set -- $cfg
  OUT=$REPO/gpurun_out/ab_$1_$2
  rm -rf $OUT; mkdir -p $OUT
  QK_SCAN_DYN_PCT=$1 QK_SCAN_DYN_CHUNK=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $REPO/bench.py --no-cpu > $OUT/stdout.log 2> $OUT/stderr.log
  echo "== pct=$1 chunk=$2"
  python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(t in n for t in ("k_scan", "k_merge", "k_group", "k_seed", "k_dense", "k_select", "k_prep")):
        print(f"{n[:40]:40s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:8.2f} max_us={float(r['MaxNs'])/1e3:8.2f}")
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
