#!/bin/bash
# round-4 evidence in one box visit: rocprofv3 passes (kernel trace + stats, then --pmc runs, each in its own process) of the
# headline, the nprobe 8 / 16 / 32 lines and the second corpus; the configs[2] bench line; probes.  Digests go to profiles/r04_*
# through scripts/make_pmc_json.py on the authoring side.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p_misc
# each profile: four rocprofv3 passes, digested HERE into three small files (the raw traces are hundreds of MB and stay on the box)
digest() {  # tag, kernel substring, bench args...
  local tag=$1 kern=$2; shift 2
  bash scripts/gpu_profile.sh r4p_$tag "$@" > /dev/null 2>&1
  python scripts/make_pmc_json.py gpurun_out/r4p_$tag "$kern" gpurun_out/r4p_misc/r04_$tag "python bench.py --no-cpu $*" > /dev/null 2> gpurun_out/r4p_misc/digest_$tag.err
  rm -rf gpurun_out/r4p_$tag
}
digest headline "k_scan<" --no-extra --no-pmc --steps 100
for np in 8 16 32; do digest np$np "k_scan_rl" --no-extra --no-pmc --steps 50 --nprobe $np; done
digest hard "k_scan_rl" --no-extra --no-pmc --steps 50 --manifold 10
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
ls -la gpurun_out/r4p_misc
du -sh gpurun_out
