#!/bin/bash
# round 6: the seeded stress scripts and the full-size soak on the final build (host-buffer calls now go through pinned staging, k > 64
# seeds from a larger sample, k-means scratch comes from a pool): every case against the oracle bit for bit
cd $GRAFT_REPO_ROOT
M=gpurun_out/r6s; mkdir -p $M
timeout 900 python scripts/stress_dense.py 400 600 2>/dev/null | tail -n 1 | cut -c1-400 | tee $M/r06_stress_dense.json
timeout 900 python scripts/stress_round4.py 200 600 2>/dev/null | tail -n 1 | cut -c1-400 | tee $M/r06_stress_round4.json
timeout 900 python scripts/stress_assign_pf.py 60 600 2>/dev/null | tail -n 1 | cut -c1-400 | tee $M/r06_stress_assign_pf.json
timeout 900 python scripts/stress_aps.py 600 6000 2>/dev/null | tail -n 1 | cut -c1-400 | tee $M/r06_stress_aps.json
timeout 1200 python scripts/soak_full_size.py 12 128 l2 10,32,100 1,8,32 2>/dev/null | tail -n 1 | cut -c1-400 | tee $M/r06_soak_full_size.json
