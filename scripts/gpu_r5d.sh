#!/bin/bash
# mixed work sequence after the staging rewrite / claim-ahead / interleaved keys
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5d
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 8,16,32 --steps 30 --tag $tag $EXTRA > gpurun_out/r5d/$tag.jsonl 2> gpurun_out/r5d/$tag.err; }
EXTRA=--parity run full
EXTRA=
run cold_only QK_SCAN_RL_PROBE=4
run hot_only QK_SCAN_RL_PROBE=16
run hot_nochain QK_SCAN_RL_PROBE=24
run full_tau0 QK_SCAN_TAU0=1
run hot_only_tau0 QK_SCAN_RL_PROBE=16 QK_SCAN_TAU0=1
run full_w20 QK_SCAN_HOT_W10=20
run full_u384 QK_SCAN_HOT_UNIT=384 QK_SCAN_HOT_W10=20
cat gpurun_out/r5d/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'], 'roof', r['frac_of_binding_roof'], r.get('ids_equal'), r.get('dist_bits_equal'))
"
