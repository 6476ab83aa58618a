#!/bin/bash
# fused group+seed launch, prep-side zeroing: parity + headline bench A/B
O=gpurun_out/r2s; mkdir -p $O
(timeout 1200 python -m pytest tests/test_scan_gpu.py tests/test_bench_parity_gpu.py tests/test_index_gpu.py tests/test_aps_gpu.py tests/test_store_dynamic_gpu.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in base nofuse; do
  e=""; [ $v = nofuse ] && e="QK_NO_GROUP_SEED=1"
  env $e timeout 600 python bench.py --no-extra --no-cpu --steps 100 --settle 50 > $O/b_${v}.json 2> $O/b_${v}.err
  env $e timeout 600 python bench.py --nprobe 8 --no-extra --no-cpu --steps 50 --settle 50 > $O/b_${v}_np8.json 2> $O/b_${v}_np8.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2s/b_*.json')):
    try:
        r=json.load(open(f)); print(f.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['phases_ms'])
    except Exception as e: print(f,'ERR',e)
PY
