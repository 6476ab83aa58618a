"""In-process interleaved sweep of QK_SCAN_WAVES_PER_CU on the bench index (10M x 128 mixture, k-means nlist=4096)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context, Store
dev = torch.device("cuda", 0)
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n, d, nlist, k, Q = 10_000_000, 128, 4096, 10, 1024
x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev)
centroids, assign, _ = ctx.kmeans(x, nlist, "l2", niter=5, seed=1234)
order = torch.argsort(assign, stable=True)
counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
offsets = np.zeros(nlist + 1, np.int64); offsets[1:] = np.cumsum(counts)
store = Store(ctx, d); store.build_csr(offsets, order.contiguous(), x[order].contiguous())
parent = Store(ctx, d); parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
q = B.gen_queries(Q, cent_true, seed=2, device=dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = {}
for rnd in range(4):
    for wpc in (4, 5, 6, 7, 8):
        os.environ["QK_SCAN_WAVES_PER_CU"] = str(wpc)
        ctx.set_timing(0)
        for _ in range(3): ctx.search(parent, store, q, P, k, "l2")
        ctx.set_timing(2)
        for _ in range(20): ctx.search(parent, store, q, P, k, "l2")
        t = ctx.read_timing()
        res.setdefault(wpc, []).append(round(t["scan_ms"] / t["calls"], 4))
print(json.dumps({"P": P, "scan_ms": res}))
