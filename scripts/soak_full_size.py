"""One-off soak at the bench size: one 10M x 128 index per corpus (mixture, low-intrinsic-dimension), many query batches x nprobe x k,
every answer compared with the oracle bit for bit (ids and float32 distances).
    python scripts/soak_full_size.py [n_batches] [d] [metric] [k,k,...] [nprobe,nprobe,...]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench as B
import oracle as O
from quake_amd.capi import Context

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
metric = sys.argv[3] if len(sys.argv) > 3 else "l2"
KS = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [10, 32]
NPS = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1, 2, 4, 8, 16, 32]
unit = metric == "ip"
n, nlist = 10_000_000, 4096
bad = 0
cases = 0
t_all = time.time()
for corpus in (("mixture", "manifold") if not unit else ("mixture",)):
    if corpus == "mixture":
        x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev, sigma=0.3, unit=unit)
    else:
        x, cent = B.gen_manifold(n, d, seed=1, device=dev, latent=10)
    idx = B.build_single(ctx, dev, x, nlist, metric, niter=5, keep_host=True)
    hv, hi, ho, hc = idx["host"]
    for b in range(nb):
        Q = [1024, 1000, 257, 64, 33, 2048][b % 6]
        if corpus == "mixture":
            q = B.gen_queries(Q, cent, seed=100 + b, device=dev, sigma=0.3, unit=unit)
        else:
            q = x[torch.randint(0, n, (Q,), device=dev, generator=torch.Generator(device=dev).manual_seed(200 + b))] + 0.01 * torch.randn(Q, d, device=dev, generator=torch.Generator(device=dev).manual_seed(300 + b))
        qh = q.cpu().numpy()
        for nprobe in NPS:
            for k in KS:
                gi, gd = ctx.search(idx["parent"], idx["store"], q, nprobe, k, metric)
                torch.cuda.synchronize()
                oi, od = O.search(qh, hc, hv, hi, ho, nprobe, k, metric, batched_scan=True)
                ok = (gi.cpu().numpy() == oi).all() and (gd.cpu().numpy().view(np.uint32) == od.view(np.uint32)).all()
                cases += 1
                if not ok:
                    bad += 1
                    print(json.dumps({"MISMATCH": True, "corpus": corpus, "batch": b, "Q": Q, "nprobe": nprobe, "k": k, "kernel": ctx.last_scan_kernel()}), flush=True)
    del x
    idx["store"].close(); idx["parent"].close(); del idx
    torch.cuda.empty_cache()
print(json.dumps({"cases": cases, "mismatches": bad, "corpora": ["mixture sigma 0.3"] + ([] if unit else ["latent dimension 10"]), "n": n, "d": d, "nlist": nlist,
                  "query_batches_per_corpus": nb, "metric": metric, "nprobes": NPS, "ks": KS, "wall_s": round(time.time() - t_all, 1)}))
sys.exit(1 if bad else 0)
