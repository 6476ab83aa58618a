#!/bin/bash
# mixed work sequence: ranges of whole 8-tile groups, item size, cost weight, hot threshold
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe_nosgb.so
mkdir -p gpurun_out/r5h
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 8,16,32 --steps 30 --tag $tag $EXTRA > gpurun_out/r5h/$tag.jsonl 2> gpurun_out/r5h/$tag.err; }
EXTRA=--parity run u512_w24
EXTRA=
run u256_w24 QK_SCAN_HOT_UNIT=256
run u384_w24 QK_SCAN_HOT_UNIT=384
run u512_w16 QK_SCAN_HOT_W10=16
run u512_w32 QK_SCAN_HOT_W10=32
run u512_min17 QK_SCAN_HOT_MIN=17
run u512_min25 QK_SCAN_HOT_MIN=25
run u512_min49 QK_SCAN_HOT_MIN=49
run hot_only QK_SCAN_RL_PROBE=16
cat gpurun_out/r5h/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'], 'roof', r['frac_of_binding_roof'], r.get('ids_equal'), r.get('dist_bits_equal'))
"
QK_SCAN_RL=1 QK_SCAN_RL_PROBE=16 QK_SCAN_WAVE_CLOCK=1 python scripts/nprobe_sweep.py --nprobes 32 --steps 2 --tag clock 2>&1 | grep -E "k_scan_rl hot" | tail -1
