#!/bin/bash
# round 5, visit E: the workload that must split (2M), store counters / phases in the records; quick suites
mkdir -p gpurun_out
python -m pytest tests/test_store_dynamic_gpu.py tests/test_maintenance_gpu.py tests/test_index_gpu.py -x -q 2>&1 | tail -3
python scripts/dynamic_workload.py 2000000 128 60 hot > gpurun_out/r05e_dyn_hot_2M.json 2> gpurun_out/r05e_dyn_hot_2M.err; echo "rc=$?"; tail -5 gpurun_out/r05e_dyn_hot_2M.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05e_dyn_hot_2M.json'))
print(d['thresholds_ns'])
print(d['device_latency_grid']['ns'])
for k, v in d['results'].items():
    print(k, {kk: vv for kk, vv in v.items() if kk != 'slow_ops'})
    for s in v['slow_ops'][:6]:
        print('   slow', s)
PY
