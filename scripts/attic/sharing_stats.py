"""How often is a partition re-streamed?  For the bench index and nprobe 8 / 16 / 32: bytes-weighted mean of ceil(q / W) over
the probed lists, q = queries of the batch probing the list, W = queries served by one read (16: one MFMA query tile;
32: one row-per-lane pass; 64: query-sharing workgroup of 4 waves; 128: of 8 waves)."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context, Store

n, d, nlist, Q = 10_000_000, 128, 4096, 1024
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev)
centroids, assign, _ = ctx.kmeans(x, nlist, "l2", niter=5, seed=1234)
sizes = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
parent = Store(ctx, d); parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
q = B.gen_queries(Q, cent_true, seed=2, device=dev)
for nprobe in (1, 2, 4, 8, 16, 32):
    pids = ctx.coarse(parent, q, nprobe, "l2")[0].cpu().numpy().reshape(-1)
    cnt = np.bincount(pids[pids >= 0], minlength=nlist)
    live = cnt > 0
    uniq = float((sizes[live]).sum())
    out = {"nprobe": nprobe, "lists_probed": int(live.sum()), "unique_GB": round(uniq * d * 4 / 1e9, 3),
           "pair_rows_over_unique": round(float((sizes * cnt).sum()) / uniq, 2), "max_queries_per_list": int(cnt.max())}
    for W in (16, 32, 64, 128, 256):
        out[f"reads_W{W}"] = round(float((sizes[live] * np.ceil(cnt[live] / W)).sum()) / uniq, 3)
    print(json.dumps(out), flush=True)
