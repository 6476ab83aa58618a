#!/bin/bash
# the launches of the LAST call of a probe in start order, with gaps (run on the GPU box):
#   bash scripts/gpu_call_trace.sh <tag> <n_last> python scripts/aps_probe.py ...
REPO=$GRAFT_REPO_ROOT
TAG=$1; shift
NLAST=$1; shift
OUT=$REPO/gpurun_out/ctrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $REPO && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- "$@" ) > $OUT/stdout.log 2> $OUT/stderr.log
grep "^{" $OUT/stdout.log
python - <<PY > $OUT/timeline.txt
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Grid_Size", ""), r.get("Workgroup_Size", "")) for r in csv.DictReader(open(f))))
rows = rows[-$NLAST:]
t0 = rows[0][0]; prev = None
for s, e, k, g, w in rows:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{gap:7.1f} gap  {(e - s) / 1e3:9.1f} us  {k:70s} grid={g} wg={w}")
    prev = e
PY
cat $OUT/timeline.txt
rm -rf $OUT/trace
