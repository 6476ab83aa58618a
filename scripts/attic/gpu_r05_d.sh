#!/bin/bash
mkdir -p gpurun_out
QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_denseprobe.so python scripts/prep_ab.py 2>/dev/null | tee gpurun_out/r05d_prep_ab.json
export QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_apsprobe.so QK_APS_FIRST=4 QK_APS_CH=80 APS_NO_CPU=1
bash scripts/gpu_kernel_stats.sh r05d_aps python scripts/aps_probe.py 10000000 4096 0.9 > gpurun_out/r05d_aps_stats.log 2>&1
tail -40 gpurun_out/r05d_aps_stats.log
