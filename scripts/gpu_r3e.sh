#!/bin/bash
# with the flat merge: is the row-per-lane form / a dynamic tail worth it at nprobe 1 now?
O=gpurun_out/r3e; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout 600 python bench.py --no-extra --no-cpu --inflight 1 --steps 100 --settle 100 > $O/b_${name}.json 2> $O/b_${name}.err
}
run base QK_NOTHING=1
run rl QK_SCAN_RL=1
run dyn20 QK_SCAN_DYN_PCT=20
run dyn35 QK_SCAN_DYN_PCT=35
run dyn35c8 QK_SCAN_DYN_PCT=35 QK_SCAN_DYN_CHUNK=8
run base2 QK_NOTHING=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms']['merge'])
    except Exception as e: print(f,'ERR',e)
PY
