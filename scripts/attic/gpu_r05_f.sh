#!/bin/bash
# round 5, visit F: deferred hit recording + counters: hot workload at 2M, the 10M test, maintenance / sharded-maintenance suites
mkdir -p gpurun_out
python -m pytest tests/test_maintenance_gpu.py tests/test_sharded_maintenance_gpu.py tests/test_index_gpu.py tests/test_random_index_streams_gpu.py -x -q 2>&1 | tail -3
python scripts/dynamic_workload.py 2000000 128 60 hot > gpurun_out/r05f_dyn_hot_2M.json 2> gpurun_out/r05f_dyn_hot_2M.err; echo "rc=$?"; tail -3 gpurun_out/r05f_dyn_hot_2M.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05f_dyn_hot_2M.json'))
for k, v in d['results'].items():
    print(k, {kk: vv for kk, vv in v.items() if kk != 'slow_ops'})
    for s in v['slow_ops'][:8]:
        print('   slow', s)
PY
(time python -m pytest tests/test_dynamic_workload_10m_gpu.py -x -q) 2>&1 | tail -15
