#!/bin/bash
# round-5 evidence in one box visit: rocprofv3 passes (kernel trace + stats, then --pmc runs, each in its own process) of the
# headline, configs[2], the nprobe 16 / 32 lines and the second corpus, digested on the box (the raw traces stay there); probes.
# Digests land in gpurun_out/r5p_misc/ and are copied to profiles/r05_* on the authoring side.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p_misc
M=gpurun_out/r5p_misc
digest() {  # tag, kernel substring, bench args...
  local tag=$1 kern=$2; shift 2
  bash scripts/gpu_profile.sh r5p_$tag "$@" > /dev/null 2>&1
  python scripts/make_pmc_json.py gpurun_out/r5p_$tag "$kern" $M/r05_$tag "python bench.py --no-cpu $*" > /dev/null 2> $M/digest_$tag.err
  rm -rf gpurun_out/r5p_$tag
}
digest headline "k_scan<" --no-extra --no-pmc --steps 100
digest c2_768ip_k100 "k_scan<" --no-extra --no-pmc --steps 50 --dim 768 --metric ip --k 100
for np in 16 32; do digest np$np "k_scan_rl" --no-extra --no-pmc --steps 50 --nprobe $np; done
digest hard "k_scan_rl" --no-extra --no-pmc --steps 50 --manifold 10
python scripts/coarse_probe.py > $M/r05_coarse_probe.jsonl 2>/dev/null
python scripts/phase_probe.py > $M/r05_phase_probe.jsonl 2>/dev/null
LAT_NO_CPU=1 python scripts/latency_probe.py > $M/r05_latency_probe.json 2>/dev/null
python scripts/nprobe_sweep.py --nprobes 2,4,8,16,32,64 --steps 50 --tag r05 --parity > $M/r05_nprobe_sweep_mixture.jsonl 2>/dev/null
python scripts/nprobe_sweep.py --nprobes 8,16,32 --corpus hard --steps 50 --tag r05 --parity > $M/r05_nprobe_sweep_hard.jsonl 2>/dev/null
python scripts/kmeans_probe.py 2>/dev/null | grep "^{" > $M/r05_kmeans_probe.jsonl
python scripts/aps_probe.py 10000000 4096 0.8 0.9 0.99 > $M/r05_aps_probe.jsonl 2>/dev/null
python scripts/skew_probe.py 2>/dev/null > $M/r05_skew_probe.jsonl
python scripts/group_probe.py > $M/r05_group_probe.jsonl 2>/dev/null
python bench.py --gpus 4 --single-process --steps 20 --warmup 3 --nvec-sharded 2500000 --nlist-sharded 1024 --batch-sharded 256 > $M/r05_bench_group4_one_gpu.json 2>/dev/null
python scripts/dynamic_workload.py 10000000 128 60 hot > $M/r05_dynamic_workload_hot_10M.json 2> $M/dyn10.err
cp gpurun_out/dynamic_workload/with_maintenance_results.json $M/r05_dynamic_workload_hot_10M_records.json 2>/dev/null
timeout 2400 python scripts/dynamic_workload.py 50000000 128 60 hot > $M/r05_dynamic_workload_hot_50M.json 2> $M/dyn50.err
ls -la $M
du -sh gpurun_out
