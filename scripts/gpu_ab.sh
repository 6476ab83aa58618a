#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu --steps 40 2>&1 | grep -E "\"value\"" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('auto', r['value'], r['roofline']['achieved'], r['phases_ms'])"
done
bash scripts/gpu_probe.sh 2>&1 | grep -v amdgpu | tail -6
