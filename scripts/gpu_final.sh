#!/bin/bash
# end-of-round evidence: rocprofv3 passes of the bench command, bench JSON lines (both configurations, CPU leg included),
# recall-target probe, dynamic workload.  Outputs under gpurun_out/final/ (copy what is to be kept into profiles/).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash scripts/gpu_profile.sh final/prof > gpurun_out/final/profile.log 2>&1
python bench.py > gpurun_out/final/c2.json 2> gpurun_out/final/c2.log
python bench.py --dim 768 --metric ip --k 100 > gpurun_out/final/c3.json 2> gpurun_out/final/c3.log
python scripts/aps_probe.py > gpurun_out/final/aps.jsonl 2> gpurun_out/final/aps.log
python scripts/latency_probe.py > gpurun_out/final/latency.json 2> gpurun_out/final/latency.log
python scripts/rank_step_probe.py 8 > gpurun_out/final/rank8.json 2> gpurun_out/final/rank8.log
tail -c 600 gpurun_out/final/c2.json; tail -c 400 gpurun_out/final/c3.json; tail -3 gpurun_out/final/aps.jsonl | cut -c1-200; tail -1 gpurun_out/final/latency.json | cut -c1-300; tail -2 gpurun_out/final/rank8.json
