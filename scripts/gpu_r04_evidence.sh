#!/bin/bash
# round-4 evidence in one box visit: rocprofv3 passes (kernel trace + stats, then --pmc runs, each in its own process) of the
# headline, the nprobe 8 / 16 / 32 lines and the second corpus; the configs[2] bench line; probes.  Digests go to profiles/r04_*
# through scripts/make_pmc_json.py on the authoring side.
cd $GRAFT_REPO_ROOT
bash scripts/gpu_profile.sh r4p_headline --no-extra --no-pmc --steps 100 > /dev/null 2>&1
for np in 8 16 32; do bash scripts/gpu_profile.sh r4p_np$np --no-extra --no-pmc --steps 50 --nprobe $np > /dev/null 2>&1; done
bash scripts/gpu_profile.sh r4p_hard --no-extra --no-pmc --steps 50 --manifold 10 > /dev/null 2>&1
mkdir -p gpurun_out/r4p_misc
python bench.py --dim 768 --metric ip --k 100 --no-extra > gpurun_out/r4p_misc/bench_c2_768ip_k100.json 2> gpurun_out/r4p_misc/c2.err
python scripts/coarse_probe.py > gpurun_out/r4p_misc/coarse_probe.jsonl 2>/dev/null
PHASE_PROBE_TIMING=0 python scripts/phase_probe.py 64 128 256 > gpurun_out/r4p_misc/phase_probe_plain.jsonl 2>/dev/null
python scripts/phase_probe.py > gpurun_out/r4p_misc/phase_probe.jsonl 2>/dev/null
python scripts/latency_probe.py > gpurun_out/r4p_misc/latency_probe.json 2>/dev/null
python scripts/rank_step_probe.py 8 > gpurun_out/r4p_misc/rank_step_probe_n8.json 2>/dev/null
python scripts/nprobe_sweep.py --nprobes 2,4,8,12,16,32,64 --steps 50 --tag r04 --parity > gpurun_out/r4p_misc/nprobe_sweep_mixture.jsonl 2>/dev/null
python scripts/nprobe_sweep.py --nprobes 8,16,32,64 --corpus hard --steps 50 --tag r04 --parity > gpurun_out/r4p_misc/nprobe_sweep_hard.jsonl 2>/dev/null
python scripts/kmeans_probe.py 2>/dev/null | grep "^{" > gpurun_out/r4p_misc/kmeans_probe.jsonl
python scripts/skew_probe.py 2>/dev/null > gpurun_out/r4p_misc/skew_probe.jsonl
python scripts/skew_probe.py 10000000 4096 8 2>/dev/null >> gpurun_out/r4p_misc/skew_probe.jsonl
ls gpurun_out/r4p_*/summary.txt
