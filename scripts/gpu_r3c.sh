#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
(QK_RANDOM_SHAPES=300 timeout 1200 python -m pytest tests/test_random_shapes_gpu.py tests/test_scan_gpu.py tests/test_bench_parity_gpu.py tests/test_index_gpu.py tests/test_aps_gpu.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_hard.json 2> $O/b_hard.err
for np in 1 8 16; do timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_np${np}.json 2> $O/b_np${np}.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms'])
    except Exception as e: print(f,'ERR',e)
PY
