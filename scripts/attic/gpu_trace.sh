#!/bin/bash
REPO=$GRAFT_REPO_ROOT
TAG=${1:-tr}; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --no-cpu "$@" > $OUT/trace_stdout.log 2> $OUT/trace_stderr.log
cd $REPO
python scripts/summarize_prof.py $OUT | head -40
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last search step: find last k_merge and print the preceding ~14 kernels with gaps
idx = [i for i, r in enumerate(rows) if "k_merge<" in r["Kernel_Name"]][-1]
prev_end = None
for r in rows[idx - 13: idx + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{r['Kernel_Name'][:60]:60s} dur_us={(e - s) / 1e3:8.2f} gap_us={gap:7.2f} grid={r.get('Grid_Size_X','')} wg={r.get('Workgroup_Size_X','')} vgpr={r.get('VGPR_Count','')} lds={r.get('LDS_Block_Size','')}")
    prev_end = e
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete
