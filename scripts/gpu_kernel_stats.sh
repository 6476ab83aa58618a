#!/bin/bash
# per-kernel totals of any command (run on the GPU box):  bash scripts/gpu_kernel_stats.sh <tag> python scripts/assign_probe.py ...
REPO=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$REPO/gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $REPO && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- "$@" ) > $OUT/stdout.log 2> $OUT/stderr.log
grep "^{" $OUT/stdout.log
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
tot = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    t = tot[r["Kernel_Name"][:90]]
    t[0] += 1; t[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k:90s} calls={n:5d} avg_us={ns / n / 1e3:10.2f} total_ms={ns / 1e6:9.3f}")
PY
rm -rf $OUT/trace
