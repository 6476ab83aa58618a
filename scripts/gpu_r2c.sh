#!/bin/bash
# round-2 probe: where the row-per-lane scan spends its time (attribution modes + wave clocks)
O=gpurun_out/r2c; mkdir -p $O
for np in 8 32; do
  for pr in 0 1 2 3; do
    QK_SCAN_RL=1 QK_SCAN_RL_PROBE=$pr timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/bench_np${np}_probe${pr}.json 2> $O/bench_np${np}_probe${pr}.err
  done
  QK_SCAN_RL=1 QK_SCAN_WAVE_CLOCK=1 timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 2 --warmup 1 --settle 2 > $O/clock_np${np}.json 2> $O/clock_np${np}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
for np in 8 32; do grep -E "k_scan launch|k_scan waves|decile|k_scan params" $O/clock_np${np}.err | tail -14; done
