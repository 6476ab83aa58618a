#!/bin/bash
# the recall-target search under each scan form (run on the GPU box): feedback on / static rule / walk only / tile form only
cd $GRAFT_REPO_ROOT
export APS_ONLY=1 APS_NO_CPU=1
echo "feedback";           python scripts/aps_probe.py 10000000 4096 0.9 | grep "^{"
echo "static (mixed)";     APS_FEEDBACK=0 python scripts/aps_probe.py 10000000 4096 0.9 | grep "^{"
echo "walk only";          APS_FEEDBACK=0 QK_SCAN_HOT_MIN=0 QK_SCAN_RL=1 python scripts/aps_probe.py 10000000 4096 0.9 | grep "^{"
echo "tile only";          APS_FEEDBACK=0 QK_SCAN_RL=0 python scripts/aps_probe.py 10000000 4096 0.9 | grep "^{"
