#!/bin/bash
# hot items with the bf16 prefilter
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5i
timeout 300 python -m pytest tests/test_scan_mixed_gpu.py -x -q 2>&1 | tail -5
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 8,16,32 --steps 30 --tag $tag $EXTRA > gpurun_out/r5i/$tag.jsonl 2> gpurun_out/r5i/$tag.err; }
EXTRA=--parity run pf
EXTRA=
run pf_min17 QK_SCAN_HOT_MIN=17
run pf_min9 QK_SCAN_HOT_MIN=9
run pf_min5 QK_SCAN_HOT_MIN=5
run pf_allexact QK_SCAN_RL_PROBE=32
run pf_noexact QK_SCAN_RL_PROBE=64
run hot_only QK_SCAN_RL_PROBE=16
cat gpurun_out/r5i/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'], 'hbm', r['hbm_frac_unique'], 'roof', r['frac_of_binding_roof'], r.get('ids_equal'), r.get('dist_bits_equal'))
"
tail -2 gpurun_out/r5i/pf.err
QK_SCAN_RL=1 QK_SCAN_WAVE_CLOCK=1 python scripts/nprobe_sweep.py --nprobes 32 --steps 2 --tag clock 2>&1 | grep -E "k_scan_rl hot|k_scan waves" | tail -2
