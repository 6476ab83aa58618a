#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for wpc in 5 6 7 8 10; do
  for P in 1 32; do
    QK_SCAN_WAVES_PER_CU=$wpc timeout 300 python scripts/scan_probe.py 10000000 4096 $P 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/wpc$wpc /"
  done
done
done
