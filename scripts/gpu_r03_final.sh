#!/bin/bash
# round-3 evidence in one box visit: rocprofv3 passes (kernel stats, FETCH_SIZE, SQ counters) of the headline, of the mixed scan at
# nprobe 8 / 16 / 32 and of the low-intrinsic-dimension corpus; the bench line; the nprobe sweeps; the coarse probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final3
mkdir -p gpurun_out/final3/digest
digest() {  # <tag> <kernel substring> <command line>: keep the small files, drop the raw rocprofv3 output (64 MiB come back at most)
  python scripts/make_pmc_json.py gpurun_out/final3/$1 "$2" gpurun_out/final3/digest/r03_$1 "$3" > gpurun_out/final3/digest/$1.log 2>&1
  rm -rf gpurun_out/final3/$1
}
bash scripts/gpu_profile.sh final3/headline --no-extra --no-pmc --steps 100 > gpurun_out/final3/headline.log 2>&1
digest headline "k_scan<" "python bench.py --no-extra --no-pmc --steps 100"
for np in 8 16 32; do
bash scripts/gpu_profile.sh final3/np$np --no-extra --no-pmc --steps 50 --nprobe $np > gpurun_out/final3/np$np.log 2>&1
digest np$np "k_scan_rl" "python bench.py --no-extra --no-pmc --steps 50 --nprobe $np"
done
bash scripts/gpu_profile.sh final3/hard --no-extra --no-pmc --steps 50 --manifold 10 > gpurun_out/final3/hard.log 2>&1
digest hard "k_scan_rl" "python bench.py --no-extra --no-pmc --steps 50 --manifold 10"
python bench.py > gpurun_out/final3/bench.json 2> gpurun_out/final3/bench.log
python bench.py --dim 768 --metric ip --k 100 --no-extra > gpurun_out/final3/c2_768.json 2> gpurun_out/final3/c2_768.log
python scripts/nprobe_sweep.py --nprobes 1,2,4,8,12,16,32,64 --steps 50 --tag r03 --parity > gpurun_out/final3/sweep_mixture.jsonl 2>/dev/null
python scripts/nprobe_sweep.py --corpus hard --nprobes 4,8,12,16,32,64 --steps 50 --tag r03 --parity > gpurun_out/final3/sweep_hard.jsonl 2>/dev/null
python scripts/coarse_probe.py 1024,2048,4096,8192,16384,32768,65536 1,2,8,32,64 > gpurun_out/final3/coarse_probe.jsonl 2>/dev/null
python scripts/phase_probe.py > gpurun_out/final3/phase_probe.jsonl 2>/dev/null
python scripts/latency_probe.py > gpurun_out/final3/latency.json 2>/dev/null
python scripts/rank_step_probe.py 8 > gpurun_out/final3/rank8.json 2>/dev/null
bash scripts/gpu_km_trace.sh > gpurun_out/final3/kmeans_trace.txt 2>&1
tail -c 1500 gpurun_out/final3/bench.json; echo; grep -h "traffic_over_algorithmic\|kernel_avg_us_rocprof\|\"kernel\"" gpurun_out/final3/digest/*_pmc.json
du -sh gpurun_out/final3
