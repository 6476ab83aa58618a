#!/bin/bash
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe_nosgb.so
for pr in 16 24 48 56; do
echo "== probe $pr"
QK_SCAN_RL=1 QK_SCAN_RL_PROBE=$pr QK_SCAN_TAU0=1 QK_SCAN_WAVE_CLOCK=1 python scripts/nprobe_sweep.py --nprobes 32 --steps 2 --tag clock 2>&1 | grep -E "k_scan_rl hot|k_scan waves" | tail -2
done
