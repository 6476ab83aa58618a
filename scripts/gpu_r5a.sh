#!/bin/bash
# mixed work sequence, first measurement: per-wave walk alone vs with hot items (probe build)
export QUAKE_HIP_LIB=$PWD/quake_amd/lib/libquake_hip_probe.so
mkdir -p gpurun_out/r5a
QK_SCAN_RL=1 QK_SCAN_HOT_MIN=0 python scripts/nprobe_sweep.py --nprobes 4,8,16,32 --tag rl_only > gpurun_out/r5a/rl_only.jsonl 2> gpurun_out/r5a/rl_only.err
QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 4,8,16,32 --tag mixed --parity > gpurun_out/r5a/mixed.jsonl 2> gpurun_out/r5a/mixed.err
QK_SCAN_RL=1 python scripts/nprobe_sweep.py --nprobes 16,32 --corpus hard --tag mixed_hard > gpurun_out/r5a/mixed_hard.jsonl 2> gpurun_out/r5a/mixed_hard.err
QK_SCAN_RL=1 QK_SCAN_HOT_MIN=0 python scripts/nprobe_sweep.py --nprobes 16,32 --corpus hard --tag rl_hard > gpurun_out/r5a/rl_hard.jsonl 2> gpurun_out/r5a/rl_hard.err
cat gpurun_out/r5a/*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['corpus'], r['nprobe'], r['kernel'], 'scan_ms', r['scan_ms'], 'hbm', r['hbm_frac_unique'], 'mfma', r['mfma_frac'], 'roof', r['frac_of_binding_roof'], 'step', r['step_ms'], r.get('ids_equal'), r.get('dist_bits_equal'))
"
tail -3 gpurun_out/r5a/*.err
