"""Unusual shapes through qk_search (huge batches, k = 448, d = 2048, nprobe > QK_MAX_K, one query over 2048 partitions):
sanity against GPU brute force.  python scripts/stress_shapes.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from quake_amd.capi import Context, Store
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
def run(n, d, nlist, Q, nprobe, k, metric="l2"):
    x, cent = B.gen_mixture(n, d, max(nlist // 2, 1), seed=1, device=dev, unit=(metric == "ip"))
    c, a, _ = ctx.kmeans(x, nlist, metric, niter=3, seed=1)
    order = torch.argsort(a, stable=True); counts = torch.bincount(a, minlength=nlist).cpu().numpy().astype(np.int64)
    off = np.zeros(nlist + 1, np.int64); off[1:] = np.cumsum(counts)
    s = Store(ctx, d); s.build_csr(off, order.contiguous(), x[order].contiguous())
    p = Store(ctx, d); p.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), c.contiguous())
    q = B.gen_queries(Q, cent, seed=2, device=dev, unit=(metric == "ip"))
    t0 = time.time(); ri, rd = ctx.search(p, s, q, nprobe, k, metric); torch.cuda.synchronize(); t = time.time() - t0
    m = min(Q, 256)
    gi, gd = B.brute_force_topk(q[:m], x, k, metric=metric)
    rec = B.recall_at_k(ri[:m], gi, k)
    ok = bool((ri >= 0).all().item()) if nprobe * (n // nlist) >= k else True
    print(f"n={n} d={d} nlist={nlist} Q={Q} nprobe={nprobe} k={k} {metric}: {t*1e3:.1f} ms recall={rec:.4f} all_filled={ok}", flush=True)
run(2_000_000, 128, 1024, 200_000, 4, 10)
run(2_000_000, 128, 1024, 50_000, 64, 10)
run(1_000_000, 16, 256, 4096, 8, 448)
run(200_000, 2048, 64, 512, 4, 10)
run(500_000, 96, 16, 20_000, 16, 100, "ip")
run(100_000, 128, 4096, 1000, 128, 10)
run(3_000_000, 128, 2048, 1, 2048, 10)
