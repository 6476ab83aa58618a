#!/bin/bash
# mixed work sequence: where does the time go?  bound 0 (nothing passes the filter), item size, block width, threshold
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5b
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 8,16,32 --steps 30 --tag $tag > gpurun_out/r5b/$tag.jsonl 2> gpurun_out/r5b/$tag.err; }
run rl_tau0 QK_SCAN_HOT_MIN=0 QK_SCAN_TAU0=1
run mixed_tau0 QK_SCAN_TAU0=1
run mixed_u512 QK_SCAN_HOT_UNIT=512
run mixed_u1024 QK_SCAN_HOT_UNIT=1024
run mixed_hq64 QK_SCAN_HOT_HQ=64
run mixed_min65 QK_SCAN_HOT_MIN=65
cat gpurun_out/r5b/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['corpus'], r['nprobe'], r['kernel'], 'scan_ms', r['scan_ms'], 'hbm', r['hbm_frac_unique'], 'mfma', r['mfma_frac'], 'roof', r['frac_of_binding_roof'], 'step', r['step_ms'])
"
QK_SCAN_RL=1 QK_SCAN_WAVE_CLOCK=1 python scripts/nprobe_sweep.py --nprobes 16 --steps 2 --tag clock 2>&1 | grep -A14 "k_scan launch" | tail -45
