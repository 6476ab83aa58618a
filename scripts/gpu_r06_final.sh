#!/bin/bash
# round 6: the driver's sequence on one box -- whole -m gpu suite (timed), smoke(), default bench line -- then the rocprofv3 passes of the
# headline step (kernel trace + stats; FETCH_SIZE in its own run) and the group probe.  Outputs: gpurun_out/r6f/ (copied to profiles/).
cd $GRAFT_REPO_ROOT
M=gpurun_out/r6f
mkdir -p $M
(time python -m pytest tests -m gpu -x -q --durations=15) > $M/r06_pytest_gpu.log 2>&1; tail -4 $M/r06_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python bench.py > $M/r06_bench.json 2> $M/r06_bench.err); echo "bench rc=$?"; tail -3 $M/r06_bench.err
bash scripts/gpu_profile.sh r6f/headline --no-extra --no-pmc --no-host-api --steps 200 --warmup 20 > $M/profile_headline.log 2>&1
cp $M/headline/summary.txt $M/r06_headline_rocprof_summary.txt 2>/dev/null
f=$(find $M/headline/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $M/r06_headline_kernel_stats.csv
rm -rf $M/headline/trace $M/headline/pmc_sq $M/headline/pmc_write
head -20 $M/r06_headline_rocprof_summary.txt
python scripts/group_probe.py > $M/r06_group_probe.jsonl 2>/dev/null; cut -c1-200 $M/r06_group_probe.jsonl
du -sh $M
