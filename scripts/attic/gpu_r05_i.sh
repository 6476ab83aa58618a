#!/bin/bash
# round 5, visit I: hot items in row-range-major order (the query blocks of a range claimed back to back) against the product order
mkdir -p gpurun_out
python scripts/nprobe_sweep.py --nprobes 16,32,64 --steps 50 2>/dev/null | tee gpurun_out/r05i_sweep_base.jsonl | cut -c1-400
QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_rrmajor.so python scripts/nprobe_sweep.py --nprobes 16,32,64 --steps 50 --parity 2>/dev/null | tee gpurun_out/r05i_sweep_rrmajor.jsonl | cut -c1-400
