#!/bin/bash
# round-2 probe: row-per-lane scan -- parity with the kernel forced on, then A/B at nprobe 1 / 8 / 32 and the hard workload
O=gpurun_out/r2b; mkdir -p $O
(QK_SCAN_RL=1 timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_index_gpu.py tests/test_bench_parity_gpu.py -m gpu -x -q) > $O/pytest_rl1.log 2>&1
tail -5 $O/pytest_rl1.log
for np in 1 8 32; do
  for rl in 0 1; do
    QK_SCAN_RL=$rl timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 100 > $O/bench_np${np}_rl${rl}.json 2> $O/bench_np${np}_rl${rl}.err
  done
done
for rl in 0 1; do
  QK_SCAN_RL=$rl timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --steps 100 > $O/bench_hard_rl${rl}.json 2> $O/bench_hard_rl${rl}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2b/bench_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['config']['nprobe'], r['config']['recall_at_k'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms'])
    except Exception as e: print(f,'ERR',e)
PY
