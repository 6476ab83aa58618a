#!/bin/bash
# schedule of the recall-target search's rounds (run on the GPU box; needs scripts/build_variant.sh apsprobe qk_aps.hip -DQK_PROBES): first round / cap per later round
cd $GRAFT_REPO_ROOT
export APS_ONLY=1 APS_NO_CPU=1
for fc in "2 80" "3 80" "4 80" "6 80" "8 80" "4 48" "4 64" "2 48"; do
  set -- $fc
  echo -n "{\"first\": $1, \"cap\": $2} "; QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_apsprobe.so APS_FEEDBACK=0 QK_APS_FIRST=$1 QK_APS_CH=$2 python scripts/aps_probe.py 10000000 4096 0.9 | grep "^{"
done
