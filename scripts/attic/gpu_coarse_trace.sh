#!/bin/bash
# per-kernel averages of the coarse step alone: bash scripts/gpu_coarse_trace.sh <nlist> <nprobe>
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/coarse_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/scripts/coarse_probe.py $1 $2 > $OUT/stdout.log 2> $OUT/stderr.log
cd $REPO
python scripts/summarize_prof.py $OUT | grep "k_pf\|k_prep\|k_dense\|k_select\|k_merge" | tee $OUT/summary_$1_$2.txt
cat $OUT/stdout.log
find $OUT -name "*kernel_trace.csv" -delete
