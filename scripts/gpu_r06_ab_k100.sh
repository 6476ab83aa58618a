#!/bin/bash
# round 6: d = 128, k = 100, nprobe 2 / 8 / 32 on the 10M bench index: the product against side libraries (scripts/build_variant.sh) that
# differ in the size of the bound sample for 64 < k <= 128 (QK_SEED_M_WIDE) and in the slack of the LDS pools (QK_SLACK_CAP)
cd $GRAFT_REPO_ROOT
M=gpurun_out/r6ab; mkdir -p $M
for np in 8 2 32; do
  for lib in product seedM4 seedM8 slack128 slack32; do
    if [ $lib = product ]; then python scripts/step_ab.py $np 100 2>/dev/null | tail -1; else QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_$lib.so python scripts/step_ab.py $np 100 2>/dev/null | tail -1; fi
  done
done | tee $M/r06_ab_k100.jsonl | cut -c1-260
