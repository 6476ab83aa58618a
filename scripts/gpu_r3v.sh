#!/bin/bash
# nprobe 8 / 16 on the mixture: actual HBM traffic of the row-per-lane launch (FETCH_SIZE) and the waves' end-time distribution
O=$GRAFT_REPO_ROOT/gpurun_out/r3v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for np in 8 16; do
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_np$np -- python $GRAFT_REPO_ROOT/bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 30 --settle 30 > $O/pmc_np$np.json 2> $O/pmc_np$np.err
  python - <<PY
import csv,glob,json
f=glob.glob("/tmp/pmc_np$np/**/*counter_collection.csv", recursive=True)[0]
tot={}; n={}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]!="FETCH_SIZE": continue
    k=r["Kernel_Name"][:40]; tot[k]=tot.get(k,0)+float(r["Counter_Value"]); n[k]=n.get(k,0)+1
b=json.loads(open("$O/pmc_np$np.json").read().strip().splitlines()[-1])
for k in tot:
    if "k_scan" in k: print("np$np", k, "launches", n[k], "FETCH KB avg", round(tot[k]/n[k],1), "traffic GB (x1024x2)", round(tot[k]/n[k]*2048/1e9,3), "unique GB", b["roofline"]["algorithmic_bytes_per_launch"]/1e9)
PY
done
cd $GRAFT_REPO_ROOT
for np in 8 16; do
  QK_SCAN_WAVE_CLOCK=1 timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 3 --warmup 2 --settle 0 > $O/wc_np$np.json 2> $O/wc_np$np.err
  grep -E "k_scan waves|decile|k_scan_rl\]" $O/wc_np$np.err | tail -13
done
