#!/bin/bash
# rocprofv3 passes over the bench command.  Usage: bash scripts/gpu_profile.sh <tag> [bench args...]
REPO=$GRAFT_REPO_ROOT
TAG=${1:-prof}; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# pass 1: kernel trace + stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py --no-cpu "$@" > $OUT/trace_stdout.log 2> $OUT/trace_stderr.log
# pass 2/3: PMC counters, each in its own run (kernel-trace only)
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py --no-cpu "$@" > $OUT/pmc_fetch_stdout.log 2> $OUT/pmc_fetch_stderr.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py --no-cpu "$@" > $OUT/pmc_write_stdout.log 2> $OUT/pmc_write_stderr.log
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python $REPO/bench.py --no-cpu "$@" > $OUT/pmc_sq_stdout.log 2> $OUT/pmc_sq_stderr.log
cd $REPO
find $OUT -name "*.csv" | head -30
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
# keep the merged output small: drop the bulky per-dispatch traces except the counter ones
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
