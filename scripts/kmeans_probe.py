"""k-means kernels on the bench shape and on a 65536-centroid case: ms of the assign and the update step of the last Lloyd iteration
(qk_kmeans_last_timing); wrap it in `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import json, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context
dev = torch.device("cuda", 0)
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for n, nlist in ((10_000_000, 4096), (4_000_000, 65536)):
    x, _ = B.gen_mixture(n, 128, min(nlist, 4096), seed=1, device=dev)
    c, a, _ = ctx.kmeans(x, nlist, "l2", niter=2, seed=1234)
    torch.cuda.synchronize()
    print(json.dumps({"n": n, "nlist": nlist, **ctx.kmeans_last_timing()}), flush=True)
    del x, c, a
