#!/bin/bash
# SQ counters per kernel of the coarse step alone: bash scripts/gpu_coarse_pmc.sh <nlist> <nprobe>
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/coarse_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python $REPO/scripts/coarse_probe.py $1 $2 > $OUT/stdout.log 2> $OUT/stderr.log
cd $REPO
python scripts/summarize_prof.py $OUT | grep -A40 "SQ counters" | grep "k_pf\|k_prep\|k_dense\|k_select\|k_merge"
cat $OUT/stdout.log
find $OUT -name "*.csv" -size +5M -delete
