#!/bin/bash
cd $GRAFT_REPO_ROOT
for mode in 2 1 0; do for wpc in 6 8; do
QK_SCAN_MODE=$mode QK_SCAN_WAVES_PER_CU=$wpc timeout 300 python scripts/scan_probe.py 10000000 4096 1 2>&1 | grep -v amdgpu.ids | tail -1
done; done
