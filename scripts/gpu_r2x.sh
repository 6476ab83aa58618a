#!/bin/bash
# team form of the row-per-lane scan (probe build): parity (fuzz + scan suite, forced RL), then the nprobe sweep with / without teams
O=gpurun_out/r2x; mkdir -p $O
(QK_SCAN_RL=1 QK_RANDOM_SHAPES=300 timeout 900 python -m pytest tests/test_random_shapes_gpu.py tests/test_scan_gpu.py tests/test_bench_parity_gpu.py -m gpu -x -q -k "not configs2 and not kmeans and not aps") > $O/pytest_rl1.log 2>&1; tail -5 $O/pytest_rl1.log
(QK_RANDOM_SHAPES=300 timeout 900 python -m pytest tests/test_random_shapes_gpu.py -m gpu -x -q -k "search_bit_exact") > $O/pytest_auto.log 2>&1; tail -2 $O/pytest_auto.log
run() { name=$1; shift
  for np in 4 8 16 32; do
    env "$@" timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_${name}_np${np}.json 2> $O/b_${name}_np${np}.err
  done
  env "$@" timeout 600 python bench.py --manifold 10 --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_${name}_hard.json 2> $O/b_${name}_hard.err
}
run team QK_SCAN_RL=1 QK_SCAN_RL_TEAM=1
run noteam QK_SCAN_RL=1 QK_SCAN_RL_TEAM=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2x/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms']['merge'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
PY
