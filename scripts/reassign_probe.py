"""What the rejection rule of the maintenance policy costs on the device: the two nearest centroids of every row of the delete
candidates (maintenance_policies.cpp:79-101) = qk_coarse(k = 2) of ~2^18 rows per call against the parent.  Times that call, the
nearest-centroid search (k = 1) of the same rows and the k-means assignment kernel, for the list counts of the 10M / 50M replays.
    python scripts/reassign_probe.py"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd.capi import Context, Store

ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
d = 128
for nlist in (3920, 19920):
    g = torch.Generator(device=dev).manual_seed(1)
    cent = torch.randn(nlist, d, generator=g, device=dev)
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), cent.contiguous())
    for rows in (1 << 16, 1 << 18):
        x = cent[torch.randint(0, nlist, (rows,), generator=g, device=dev)] + 0.3 * torch.randn(rows, d, generator=g, device=dev)
        out = {}
        for name, fn in (("coarse_k2", lambda: ctx.coarse(parent, x, 2, "l2", values=False)),
                         ("coarse_k1", lambda: ctx.coarse(parent, x, 1, "l2", values=False)),
                         ("kmeans_assign", lambda: ctx.kmeans_assign(x, cent, "l2", values=False))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            out[name + "_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        print(json.dumps({"nlist": nlist, "rows": rows, **out, "pair_gflop": round(2e-9 * rows * nlist * d, 1)}), flush=True)
    parent.close()
