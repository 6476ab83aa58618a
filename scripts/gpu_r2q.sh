#!/bin/bash
# k_scan_rl phase probes at nprobe 32 / 8: bit 1 = no epilogue, bit 2 = no MFMA chain (results wrong: timing only)
O=gpurun_out/r2q; mkdir -p $O
for np in 32 8; do
for pr in 0 1 2 3; do
  QK_SCAN_RL=1 QK_SCAN_RL_PROBE=$pr timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 30 --settle 30 > $O/b_np${np}_p${pr}.json 2> $O/b_np${np}_p${pr}.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2q/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms']['merge'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
