#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  QK_SCAN_MODE=1 timeout 300 python scripts/scan_probe.py 10000000 4096 1 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/mode1 /"
  QK_SCAN_TAU0=1 timeout 300 python scripts/scan_probe.py 10000000 4096 1 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/tau0 /"
  timeout 300 python scripts/scan_probe.py 10000000 4096 1 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/full /"
done
