#!/bin/bash
# round 6, late: the 2-means of a maintenance call's splits on worker contexts (QUAKE_SPLIT_THREADS, default 8) -- parity files of both
# mirrors, then the 10M hot replay with 8 threads and with 1 (python mirror), and the compiled mirror with 8
R=$GRAFT_REPO_ROOT; M=$R/gpurun_out/r6t; mkdir -p $M
cd $R
timeout 900 python -m pytest tests/test_maintenance_gpu.py tests/test_index_gpu.py tests/test_bindings_gpu.py tests/test_random_index_streams_gpu.py tests/test_dynamic_workload_10m_gpu.py tests/test_workers_gpu.py -m gpu -x -q 2>&1 | tail -n 4 | tee $M/r06_split_threads_pytest.log
bash scripts/gpu_r06_maint.sh 10000000 thr8_10M DW_EXT=1 2>&1 | tail -n 2 | cut -c1-1200
QUAKE_SPLIT_THREADS=1 bash scripts/gpu_r06_maint.sh 10000000 thr1_10M DW_EXT=1 2>&1 | tail -n 1 | cut -c1-1200
bash scripts/gpu_r06_maint.sh 10000000 thr8_10M_compiled DW_EXT=1 DW_MIRROR=compiled 2>&1 | tail -n 1 | cut -c1-1200
