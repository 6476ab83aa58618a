#!/bin/bash
# per-kernel averages of the k-means update path (scripts/kmeans_probe.py)
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/km_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/scripts/kmeans_probe.py > $OUT/stdout.log 2> $OUT/stderr.log
cd $REPO
python scripts/summarize_prof.py $OUT | head -30 | tee $OUT/summary.txt
cat $OUT/stdout.log
find $OUT -name "*kernel_trace.csv" -delete
