#!/bin/bash
# round 6: where the step at d = 128, k = 100 goes (nprobe 8 and 32): kernel trace + stats of scripts/step_ab.py
R=$GRAFT_REPO_ROOT; M=$R/gpurun_out/r6k; mkdir -p $M
cd /tmp && export TMPDIR=/tmp
for np in 8 32; do
  rm -rf /tmp/k100_$np
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k100_$np -- python $R/scripts/step_ab.py $np 100 > $M/step_np$np.log 2>&1
  f=$(find /tmp/k100_$np -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
print("== nprobe $np, k = 100")
for r in rows[:12]:
    if any(t in r["Name"] for t in ("k_", "qk")): print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f} total_ms={float(r["TotalDurationNs"])/1e6:9.2f}')
PY
done | tee $M/r06_k100_kernel_stats.txt
