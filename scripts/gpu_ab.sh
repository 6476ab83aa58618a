#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for side in 0 1; do
  for P in 1 32; do
    QK_SEED_SIDE_STREAM=$side timeout 300 python scripts/scan_probe.py 10000000 4096 $P 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/side$side /"
  done
done
QK_NO_SEED=1 timeout 300 python scripts/scan_probe.py 10000000 4096 1 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/noseed /"
done
