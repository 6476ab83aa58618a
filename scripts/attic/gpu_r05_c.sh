#!/bin/bash
# round 5, visit C: fused prep (pipelined staging) headline; APS without stream synchronisation + schedule sweep
mkdir -p gpurun_out
python -m pytest tests/test_aps_gpu.py tests/test_scan_gpu.py tests/test_bench_parity_gpu.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 3 --no-extra --no-pmc --no-cpu > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r05c_bench.json')); print(d['value'], d['ms_per_step'], d['phases_ms'], d['roofline']['frac'], d['timed_groups']['min'], d['timed_groups']['max'])"
APS_NO_CPU=1 python scripts/aps_probe.py 10000000 4096 0.9 2>/dev/null | tee gpurun_out/r05c_aps_product.jsonl
for cfg in "2 32" "4 48" "8 48" "8 64" "16 64" "4 80" "12 80"; do
  set -- $cfg
  echo "FIRST=$1 CH=$2"
  QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_apsprobe.so QK_APS_FIRST=$1 QK_APS_CH=$2 APS_NO_CPU=1 python scripts/aps_probe.py 10000000 4096 0.9 0.99 2>/dev/null | tee -a gpurun_out/r05c_aps_sweep.jsonl
done
