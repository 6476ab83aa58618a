#!/bin/bash
# metric as a template parameter (k_scan, k_scan_rl, k_dense_ord, k_dense_argmin, k_assign): parity, coarse probe, nprobe sweep
O=gpurun_out/r2r; mkdir -p $O
(timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_bench_parity_gpu.py tests/test_kmeans_gpu.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
(QK_SCAN_RL=1 timeout 900 python -m pytest tests/test_scan_gpu.py -m gpu -x -q) > $O/pytest_rl1.log 2>&1; tail -2 $O/pytest_rl1.log
python scripts/coarse_probe.py > $O/coarse.jsonl 2> $O/coarse.err; grep '"nprobe": 1,' $O/coarse.jsonl; grep '"nprobe": 32,' $O/coarse.jsonl
for np in 1 2 4 8 16 32; do
  timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/b_auto_np${np}.json 2> $O/b_auto_np${np}.err
done
timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --steps 50 --settle 50 > $O/b_auto_hard.json 2> $O/b_auto_hard.err
for np in 16 32; do
  QK_SCAN_RL=1 timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --steps 50 --settle 50 > $O/b_rl_np${np}.json 2> $O/b_rl_np${np}.err
done
timeout 600 python bench.py --dim 768 --metric ip --k 100 --no-extra --no-cpu > $O/b_c3.json 2> $O/b_c3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2r/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms'])
    except Exception as e: print(f,'ERR',e)
PY
