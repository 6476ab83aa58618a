#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fused
timeout 600 python -m pytest tests/test_dense_fused_gpu.py -m gpu -x -q 2>&1 | tail -3
for sr in 256 512 1024; do
echo "SR=$sr MAX=8192"
QK_FUSED_MAX=8192 QK_FUSED_SR=$sr python scripts/coarse_probe.py 2048,4096,8192 2,8,32,64 2>/dev/null | tee -a gpurun_out/fused/tune2.jsonl
done
echo pf
QK_FUSED_MAX=1 python scripts/coarse_probe.py 2048,4096,8192 2,8,32,64 2>/dev/null | tee -a gpurun_out/fused/tune2.jsonl
export QK_FUSED_MAX=8192 QK_FUSED_SR=1024
bash scripts/gpu_coarse_trace.sh 4096 2,32 2>&1 | grep -v "^{" | sed 's/^/trace4096 sr1024: /'
