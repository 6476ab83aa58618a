#!/bin/bash
# broadcast-A form of k_scan_rl (probe build): parity, then the nprobe sweep forced on the row-per-lane form + wave clocks
O=gpurun_out/r2p; mkdir -p $O
(QK_SCAN_RL=1 timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_bench_parity_gpu.py -m gpu -x -q -k "not configs2") > $O/pytest_rl1.log 2>&1; tail -3 $O/pytest_rl1.log
run() { name=$1; shift
  for np in 1 2 4 8 16 32; do
    env "$@" timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/b_${name}_np${np}.json 2> $O/b_${name}_np${np}.err
  done
  env "$@" timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --steps 50 --settle 50 > $O/b_${name}_hard.json 2> $O/b_${name}_hard.err
}
run rl QK_SCAN_RL=1
run auto QK_SCAN_RL=-1
for np in 8 32; do QK_SCAN_RL=1 QK_SCAN_WAVE_CLOCK=1 timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 2 --warmup 1 --settle 2 > $O/clock_np${np}.json 2> $O/clock_np${np}.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2p/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms']['merge'])
    except Exception as e: print(f,'ERR',e)
PY
for np in 8 32; do grep -E "k_scan_rl\]|k_scan waves" $O/clock_np${np}.err | tail -2; done
