#!/bin/bash
# flat id map in the store: store / index / stream tests, dynamic workload at 10M
O=gpurun_out/r3i; mkdir -p $O
(timeout 1200 python -m pytest tests/test_store_dynamic_gpu.py tests/test_index_gpu.py tests/test_random_index_streams_gpu.py tests/test_sharded_maintenance_gpu.py tests/test_maintenance_gpu.py -m gpu -q -x 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
python scripts/dynamic_workload.py 10000000 128 60 > $O/dynamic_10M.json 2> $O/dynamic_10M.err; tail -2 $O/dynamic_10M.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3i/dynamic_10M.json').read().strip().splitlines()[-1])
for k,v in r['results'].items(): print(k, json.dumps({a:b for a,b in v.items() if not isinstance(b,(list,dict))}))
PY
