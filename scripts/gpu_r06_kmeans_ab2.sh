#!/bin/bash
# round 6, late: rows per wave of the bucketing again, now that its loops keep 16 rows in flight (512 / 1024 / 2048), then the bench line
R=$GRAFT_REPO_ROOT; M=$R/gpurun_out/r6k; mkdir -p $M
cd $R
for rep in 1 2; do for lib in product wr512 wr2048; do
  if [ $lib = product ]; then timeout 300 python scripts/kmeans_probe.py 2>/dev/null; else QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_$lib.so timeout 300 python scripts/kmeans_probe.py 2>/dev/null; fi | sed "s/^{/{\"lib\": \"$lib\", /"
done; done | tee $M/r06_kmeans_wave_rows_ab2.jsonl
timeout 900 python bench.py > $M/r06_bench_after_bucketing.json 2> $M/bench_err.log; tail -c 600 $M/bench_err.log
python - <<PY
import json
b=json.loads(open("$M/r06_bench_after_bucketing.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], json.dumps(b["build"]))
PY
