#!/bin/bash
O=gpurun_out/r3a; mkdir -p $O
(QK_SCAN_RL=1 QK_RANDOM_SHAPES=100 timeout 900 python -m pytest tests/test_random_shapes_gpu.py tests/test_scan_gpu.py -m gpu -x -q -k "not kmeans and not aps") > $O/pytest_rl1.log 2>&1; tail -2 $O/pytest_rl1.log
for np in 4 8 16 32; do
  QK_SCAN_RL=1 timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_rl_np${np}.json 2> $O/b_rl_np${np}.err
done
timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_rl_hard.json 2> $O/b_rl_hard.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
