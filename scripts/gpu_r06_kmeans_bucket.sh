#!/bin/bash
# round 6, late: the bucketing in front of the k-means mean update (k_rs_hist / k_rs_scan / k_rs_scatter) with 16 rows in flight per
# lane and no device-scope atomics -- parity first, then the update's time and its per-kernel split
R=$GRAFT_REPO_ROOT; M=$R/gpurun_out/r6k; mkdir -p $M
cd $R
if [ "$1" != "profile-only" ]; then
timeout 600 python -m pytest tests/test_kmeans_gpu.py tests/test_maintenance_gpu.py -m gpu -x -q 2>&1 | tail -n 4 | tee $M/r06_kmeans_bucket_pytest.log
for i in 1 2 3; do timeout 300 python scripts/kmeans_probe.py 2>/dev/null; done | tee $M/r06_kmeans_bucket_probe.jsonl
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kmprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kmprof -- python $R/scripts/kmeans_probe.py > $M/kmprof.log 2>&1
f=$(find /tmp/kmprof -name "*kernel_stats.csv" | head -1)
python - <<PY | tee $M/r06_kmeans_bucket_kernel_stats.txt
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows:
    if any(t in r["Name"] for t in ("k_rs_", "k_accumulate", "k_segment", "k_assign", "k_finalize", "fill", "Memset", "memset")): print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f} min_us={float(r["MinNs"])/1e3:9.2f} max_us={float(r["MaxNs"])/1e3:9.2f}')
PY
