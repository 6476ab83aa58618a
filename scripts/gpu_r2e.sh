#!/bin/bash
O=gpurun_out/r2e; mkdir -p $O
(timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_index_gpu.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
(QK_SCAN_RL=1 timeout 900 python -m pytest tests/test_scan_gpu.py -m gpu -x -q) > $O/pytest_rl1.log 2>&1; tail -3 $O/pytest_rl1.log
LAT_NO_CPU=1 python scripts/latency_probe.py > $O/latency.log 2>&1; tail -3 $O/latency.log
LAT_NO_CPU=1 QK_SMALL_MAX_Q=64 python scripts/latency_probe.py > $O/latency_q64.log 2>&1; tail -2 $O/latency_q64.log
LAT_NO_CPU=1 QK_SMALL_CLOCK=1 python scripts/latency_probe.py 2>&1 | grep k_search_small | awk 'NR%40==1' | head -12
for pr in 0 1 4 5; do
  for np in 8 32; do
    QK_SCAN_RL=1 QK_SCAN_RL_PROBE=$pr timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/b_probe${pr}_np${np}.json 2> $O/b_probe${pr}_np${np}.err
  done
done
for np in 8 32; do QK_SCAN_RL=1 QK_SCAN_WAVE_CLOCK=1 timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 2 --warmup 1 --settle 2 > $O/clock_np${np}.json 2> $O/clock_np${np}.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2e/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
for np in 8 32; do grep -E "k_scan_rl\]|k_scan waves" $O/clock_np${np}.err | tail -2; done
