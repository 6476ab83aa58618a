#!/bin/bash
# bf16 prefilter, exact chains per flagged row tile: threshold sweep + counters
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5j
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 4,8,16,32 --steps 30 --tag $tag $EXTRA > gpurun_out/r5j/$tag.jsonl 2> gpurun_out/r5j/$tag.err; }
EXTRA=--parity run min33
run min5 QK_SCAN_HOT_MIN=5
EXTRA=
run min9 QK_SCAN_HOT_MIN=9
run min17 QK_SCAN_HOT_MIN=17
run min5_noexact QK_SCAN_HOT_MIN=5 QK_SCAN_RL_PROBE=64
cat gpurun_out/r5j/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'], 'hbm', r['hbm_frac_unique'], 'roof', r['frac_of_binding_roof'], r.get('ids_equal'), r.get('dist_bits_equal'))
"
for np in 8 32; do
QK_SCAN_HOT_MIN=5 QK_SCAN_RL=1 QK_SCAN_WAVE_CLOCK=1 python scripts/nprobe_sweep.py --nprobes $np --steps 2 --tag clock 2>&1 | grep -E "k_scan_rl hot|k_scan_rl prefilter|k_scan waves|decile" | tail -13
done
