#!/bin/bash
# SQ counters of k_assign_pf on the bench shape (two passes of counters): bash scripts/gpu_assign_pmc.sh [shape ...]
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/assign_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPES=${@:-1048576x4096x128}
i=0
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  ( cd $REPO && timeout 600 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/p$i -- python scripts/assign_probe.py $SHAPES ) > $OUT/stdout$i.log 2> $OUT/stderr$i.log
  python - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/p$i/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "k_assign" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k, {c: round(v) for c, v in d.items()})
PY
  rm -rf $OUT/p$i
done
