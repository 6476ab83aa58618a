#!/bin/bash
# workgroup-cooperative row-per-lane passes (QK_SCAN_RL_TEAM=1, probe build): parity, then nprobe 8 / 16 / 32 / hard against the product form
O=gpurun_out/r4f; mkdir -p $O
(QK_SCAN_RL_TEAM=1 QK_SCAN_RL=1 QK_RANDOM_SHAPES=300 timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_random_shapes_gpu.py -m gpu -q -x 2>&1 | tail -15) > $O/pytest.log; tail -6 $O/pytest.log | cut -c1-200
run() { name=$1; shift; args=$1; shift
  env "$@" timeout 600 python bench.py $args --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_${name}.json 2> $O/b_${name}.err || tail -3 $O/b_${name}.err
}
for np in 4 8 16 32; do
  run np${np}_base "--nprobe $np" QK_SCAN_RL=1
  run np${np}_team "--nprobe $np" QK_SCAN_RL=1 QK_SCAN_RL_TEAM=1
done
run hard_base "--manifold 10" QK_X=1
run hard_team "--manifold 10" QK_SCAN_RL_TEAM=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4f/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms_avg'], r['roofline'].get('hbm',r['roofline'])['frac'], r['phases_ms']['merge'], r['config']['recall_at_k'])
    except Exception as e: print(f,'ERR',e)
PY
