#!/bin/bash
# mixed work sequence, attribution: per-wave sequence alone (probe 4), hot items alone (16), hot items without chains (24)
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5c
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 8,16,32 --steps 30 --tag $tag > gpurun_out/r5c/$tag.jsonl 2> gpurun_out/r5c/$tag.err; }
run full
run cold_only QK_SCAN_RL_PROBE=4
run hot_only QK_SCAN_RL_PROBE=16
run hot_nochain QK_SCAN_RL_PROBE=24
run full_tau0 QK_SCAN_TAU0=1
run cold_only_tau0 QK_SCAN_RL_PROBE=4 QK_SCAN_TAU0=1
run hot_only_tau0 QK_SCAN_RL_PROBE=16 QK_SCAN_TAU0=1
run hot_nochain_tau0 QK_SCAN_RL_PROBE=24 QK_SCAN_TAU0=1
cat gpurun_out/r5c/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'])
"
