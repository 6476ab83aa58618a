#!/bin/bash
# the one-launch coarse form (qk_dense_fused.hip): its parity tests, the coarse probe over the sizes it serves and their neighbours,
# per-kernel averages at 2048 centroids
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fused
timeout 900 python -m pytest tests/test_dense_fused_gpu.py tests/test_dense_pf_gpu.py -m gpu -x -q 2>&1 | tail -3
python scripts/coarse_probe.py 1024,2048,4096 2,8,32,64 2>/dev/null | tee gpurun_out/fused/coarse_probe.jsonl
bash scripts/gpu_coarse_trace.sh 2048 8,32 2>&1 | grep -v "^{"
