#!/bin/bash
# kernel sequence (durations, gaps) of the last search steps of the bench workload at a given nprobe:
#   bash scripts/gpu_trace_search.sh <nprobe> [extra bench.py args]
# (bench.py --traffic-probe = build + 8 searches; the trace's last 3 searches are printed)
REPO=$GRAFT_REPO_ROOT
NP=$1; shift
OUT=$REPO/gpurun_out/search_trace_np$NP
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $REPO/bench.py --traffic-probe --nprobe $NP "$@" > $OUT/stdout.log 2> $OUT/stderr.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the searches: from the last k_prep_queries-or-first-kernel boundaries; print the last 24 kernels
prev_end = None
for r in rows[-24:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{r['Kernel_Name'][:64]:64s} dur_us={(e - s) / 1e3:8.2f} gap_us={gap:7.2f} grid={r.get('Grid_Size_X','')} wg={r.get('Workgroup_Size_X','')}")
    prev_end = e
PY
rm -rf $OUT/trace
