#!/bin/bash
# mixed sequence with prefilter: dynamic share of the per-wave sequence x item size x threshold
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5k
run() { tag=$1; shift; env "$@" QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 8,16,32 --steps 30 --tag $tag > gpurun_out/r5k/$tag.jsonl 2> gpurun_out/r5k/$tag.err; }
for dp in 25 50 70; do for un in 200 300 600; do
run min5_d${dp}_u${un} QK_SCAN_HOT_MIN=5 QK_SCAN_RL_DYN_PCT=$dp QK_SCAN_HOT_UNIT=$un
done; done
for mn in 1 2 3; do
run min${mn}_d50_u300 QK_SCAN_HOT_MIN=$mn QK_SCAN_RL_DYN_PCT=50 QK_SCAN_HOT_UNIT=300
done
cat gpurun_out/r5k/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['nprobe'], 'scan_ms', r['scan_ms'], 'hbm', r['hbm_frac_unique'], 'roof', r['frac_of_binding_roof'])
"
