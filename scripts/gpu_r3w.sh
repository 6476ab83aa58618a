#!/bin/bash
# row-per-lane scan: 3 waves per CU with 64-query passes vs 4 waves with 32 (probe build)
O=gpurun_out/r3w; mkdir -p $O
run() { name=$1; np=$2; shift; shift
  env "$@" timeout 600 python bench.py --nprobe $np --no-extra --no-cpu --inflight 1 --steps 50 --settle 50 > $O/b_${name}.json 2> $O/b_${name}.err || tail -3 $O/b_${name}.err
}
for np in 8 16 32; do
  run np${np}_w4q32 $np QK_SCAN_RL=1
  run np${np}_w3q64 $np QK_SCAN_RL=1 QK_SCAN_RL_WAVES=3 QK_SCAN_RL_QB=64
  run np${np}_w3q32 $np QK_SCAN_RL=1 QK_SCAN_RL_WAVES=3
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3w/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['kernel_ms_avg'], r['roofline'].get('hbm',r['roofline'])['frac'], r['phases_ms']['merge'], r['config']['recall_at_k'])
    except Exception as e: print(f,'ERR',e)
PY
(QK_SCAN_RL_WAVES=3 QK_SCAN_RL_QB=64 QK_RANDOM_SHAPES=300 timeout 900 python -m pytest tests/test_random_shapes_gpu.py tests/test_scan_gpu.py -m gpu -q -x 2>&1 | tail -3)
