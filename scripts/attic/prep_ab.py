"""A/B of the query preparation folded into the nearest-centroid kernel (k_dense_argmin<.., FUSE>) against the separate prep launch, in ONE
process on one index: alternating blocks of steps with QK_NO_FUSED_PREP set / unset (probe build of qk_dense.hip:
QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_denseprobe.so)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context

dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n, d, nlist, Q, k = 10_000_000, 128, 4096, 1024, 10
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev)
idx = B.build_single(ctx, dev, x, nlist, "l2", 5, keep_host=False)
del x
qs = [B.gen_queries(Q, cent, seed=2 + b, device=dev) for b in range(4)]
out = (torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev))


def block(steps=200):
    for i in range(20):
        ctx.search(idx["parent"], idx["store"], qs[i % 4], 1, k, "l2", out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ctx.search(idx["parent"], idx["store"], qs[i % 4], 1, k, "l2", out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for i in range(300):
    ctx.search(idx["parent"], idx["store"], qs[i % 4], 1, k, "l2", out=out)
torch.cuda.synchronize()
res = {"fused": [], "separate": []}
for rep in range(6):
    os.environ.pop("QK_NO_FUSED_PREP", None)
    res["fused"].append(round(block(), 5))
    os.environ["QK_NO_FUSED_PREP"] = "1"
    res["separate"].append(round(block(), 5))
print(json.dumps({"ms_per_step": res, "fused_median": float(np.median(res["fused"])), "separate_median": float(np.median(res["separate"]))}))
