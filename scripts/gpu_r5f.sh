#!/bin/bash
# hot items with double-buffered B operands: with / without the MFMA-VALU interleave pipeline
mkdir -p gpurun_out/r5f
for v in probe probe_nosgb; do
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_$v.so
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 8,16,32 --steps 30 --tag ${v}_$tag $EXTRA > gpurun_out/r5f/${v}_$tag.jsonl 2> gpurun_out/r5f/${v}_$tag.err; }
EXTRA=--parity run full
EXTRA=
run hot_only QK_SCAN_RL_PROBE=16
run hot_only_tau0 QK_SCAN_RL_PROBE=16 QK_SCAN_TAU0=1
echo "== $v clock"
QK_SCAN_RL=1 QK_SCAN_RL_PROBE=16 QK_SCAN_WAVE_CLOCK=1 python scripts/nprobe_sweep.py --nprobes 32 --steps 2 --tag clock 2>&1 | grep -E "k_scan_rl hot" | tail -1
done
cat gpurun_out/r5f/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'], 'roof', r['frac_of_binding_roof'], r.get('ids_equal'), r.get('dist_bits_equal'))
"
