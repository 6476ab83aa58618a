#!/bin/bash
# kernel sequence (durations and gaps) of the last search step of scripts/phase_probe.py at the given batch sizes
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/step_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for Q in "$@"; do
  rm -rf $OUT/trace_$Q
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$Q -- python $REPO/scripts/phase_probe.py $Q > $OUT/stdout_$Q.log 2> $OUT/stderr_$Q.log
  python - <<PY > $OUT/seq_$Q.txt
import csv, glob
f = glob.glob("$OUT/trace_$Q/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
t0 = None
for r in rows[-40:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{r['Kernel_Name'][:70]:70s} dur_us={(e - s) / 1e3:8.2f} gap_us={gap:7.2f} grid={r.get('Grid_Size_X','')} wg={r.get('Workgroup_Size_X','')} lds={r.get('LDS_Block_Size','')}")
    prev_end = e
PY
  cat $OUT/stdout_$Q.log; cat $OUT/seq_$Q.txt
  rm -rf $OUT/trace_$Q
done
