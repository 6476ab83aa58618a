#!/bin/bash
# pass width of the row-per-lane scan: 32 (pools k+32) vs 48 (pools k+16) queries per pass; probe build
O=gpurun_out/r3h; mkdir -p $O
(timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_random_shapes_gpu.py tests/test_bench_parity_gpu.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
run() { name=$1; np=$2; shift; shift
  env "$@" timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_${name}.json 2> $O/b_${name}.err
}
for np in 4 8 16 32; do
  run np${np}_qb32 $np QK_SCAN_RL_QB=32 QK_SCAN_RL=1
  run np${np}_qb48 $np QK_SCAN_RL_QB=48 QK_SCAN_RL=1
done
run np16_qb40 16 QK_SCAN_RL_QB=40 QK_SCAN_RL=1
env QK_SCAN_RL_QB=32 timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_hard_qb32.json 2> $O/b_hard_qb32.err
timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_hard_rule.json 2> $O/b_hard_rule.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms_avg'], r['roofline'].get('hbm',r['roofline'])['frac'], r['phases_ms']['merge'], r['config']['recall_at_k'])
    except Exception as e: print(f,'ERR',e)
PY
