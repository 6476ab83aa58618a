#!/bin/bash
# mixed work sequence: phases of the hot items (wave clock probe)
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
for np in 16 32; do
for pr in 0 16 24; do
echo "== nprobe $np probe $pr"
QK_SCAN_RL=1 QK_SCAN_RL_PROBE=$pr QK_SCAN_WAVE_CLOCK=1 python scripts/nprobe_sweep.py --nprobes $np --steps 2 --tag clock 2>&1 | grep -E "k_scan_rl hot|k_scan waves" | tail -2
done; done
