"""Per-list (rows, probing queries) of the bench batches, for the mixture and the low-intrinsic-dimension corpus at nprobe
1..64: the input of scripts/scan_model.py (what a work sequence of cold / hot items would cost).  Writes
gpurun_out/list_stats.npz."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context, Store

n, d, nlist, Q = 10_000_000, 128, 4096, 1024
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
out = {}
for name in ("mixture", "hard"):
    if name == "mixture":
        x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev)
        q = B.gen_queries(Q, cent_true, seed=2, device=dev)
    else:
        x, basis = B.gen_manifold(n, d, seed=1, device=dev)
        q, _ = B.gen_manifold(Q, d, seed=2, device=dev, basis=basis)
    centroids, assign, _ = ctx.kmeans(x, nlist, "l2", niter=5, seed=1234)
    sizes = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
    out[f"{name}_sizes"] = sizes
    for nprobe in (1, 2, 4, 8, 16, 32, 64):
        pids = ctx.coarse(parent, q, nprobe, "l2")[0].cpu().numpy().reshape(-1)
        out[f"{name}_cnt_{nprobe}"] = np.bincount(pids[pids >= 0], minlength=nlist)
    del x, assign, parent
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/list_stats.npz", **out)
print("ok", {k: (v.shape, int(v.sum())) for k, v in out.items()})
