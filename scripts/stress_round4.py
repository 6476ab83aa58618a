"""Seeded stress of the pieces round 4 added, against the oracle, on shapes nobody picked by hand:
  (1) qk_kmeans_accumulate / _blocked: random n, m, d, skew, out-of-range assignments, host and device buffers
  (2) searches repeated on one context with form feedback ON (the context switches forms between calls): every call bit-equal
  (3) the packed exchange: pack -> (identity all-to-all) -> merge_packed against merge_topk
python scripts/stress_round4.py [n_cases] [seed0]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from helpers import make_ivf, make_queries
from quake_amd.capi import Context, Store

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # device tensors go through torch ops (cat, equal) below
bad = []
t0 = time.time()
for c in range(ncases):
    rng = np.random.default_rng(seed0 + c)
    # (1)
    n = int(rng.choice([1, 63, 1000, 4097, 50000, 300000]))
    m = int(rng.choice([1, 2, 17, 255, 256, 257, 4095, 4096, 4097, 70000]))
    d = int(rng.choice([1, 3, 4, 16, 100, 128, 132, 200]))
    x = (rng.standard_normal((n, d)) * 10.0 ** rng.integers(-2, 3, size=(n, 1))).astype(np.float32)
    a = rng.integers(0, m, size=n).astype(np.int64)
    if rng.random() < 0.5:
        a[rng.random(n) < 0.4] = rng.integers(0, m)        # one heavy cluster
    if rng.random() < 0.5:
        a[rng.integers(0, n, size=max(1, n // 50))] = rng.choice([-1, m, m + 7])
    for blocked in (False, True):
        os_, oc = O.kmeans_accumulate(x, a, m, blocked=blocked)
        if rng.random() < 0.5:
            gs, gc = ctx.kmeans_accumulate(x, a, m, blocked=blocked)
        else:
            ds, dc = ctx.kmeans_accumulate(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda(), m, blocked=blocked)
            torch.cuda.synchronize()
            gs, gc = ds.cpu().numpy(), dc.cpu().numpy()
        if not (np.array_equal(gc, oc) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))):
            bad.append(("accumulate", seed0 + c, n, m, d, blocked))
    # (2)
    dd = int(rng.choice([32, 64, 100, 128]))
    nlist = int(rng.choice([16, 48, 200]))
    nv = int(rng.choice([60000, 150000, 400000]))
    ivf = make_ivf(nv, dd, nlist, seed=seed0 + c)
    s = Store(ctx, dd); s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    p = Store(ctx, dd); p.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), ivf["centroids"])
    Q = int(rng.choice([256, 700, 1024]))
    nprobe = int(rng.choice([2, 4, 8, min(16, nlist)]))
    k = int(rng.choice([1, 10, 32]))
    if rng.random() < 0.5:
        q = (ivf["x"][rng.integers(0, 4, Q)] + 0.05 * rng.standard_normal((Q, dd))).astype(np.float32)   # concentrated batch
    else:
        q = make_queries(Q, dd, seed=seed0 + c + 5, like=ivf["x"])
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, "l2", batched_scan=True)
    qd = torch.from_numpy(q).cuda()
    forms = set()
    for rep in range(10):
        gi, gd = ctx.search(p, s, qd, nprobe, k, "l2")
        torch.cuda.synchronize()
        forms.add(ctx.last_scan_kernel())
        if not (np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gd.cpu().numpy().view(np.uint32), od.view(np.uint32))):
            bad.append(("search", seed0 + c, nv, dd, nlist, Q, nprobe, k, rep, ctx.last_scan_kernel()))
            break
    s.close(); p.close()
    # (3)
    G = int(rng.choice([1, 2, 3, 8])); per = int(rng.choice([1, 5, 64, 129])); kk = int(rng.choice([1, 7, 10, 100]))
    rid = rng.integers(0, 1 << 40, size=(G, per, kk)).astype(np.int64)
    rk = np.sort(rng.integers(0, 40, size=(G, per, kk)).astype(np.float32) * 0.5, axis=2)
    rid[:, :, kk - 1:] = np.where(rng.random((G, per, 1)) < 0.3, -1, rid[:, :, kk - 1:])
    # a rank's send buffer is [G*per][k]; receiving G such blocks for OUR per queries = stacking block 0 of each source
    recv = torch.cat([ctx.pack_topk(torch.from_numpy(rid[g]).cuda(), torch.from_numpy(rk[g]).cuda(), 1) for g in range(G)], 0)
    pi, pd = ctx.merge_topk_packed(recv, per, kk, "l2")
    mi, md = ctx.merge_topk(torch.from_numpy(rid).cuda(), torch.from_numpy(rk).cuda(), "l2")
    torch.cuda.synchronize()
    if not (torch.equal(pi, mi) and torch.equal(pd.view(torch.int32), md.view(torch.int32))):
        bad.append(("packed", seed0 + c, G, per, kk))
    print(json.dumps({"case": c, "forms": sorted(forms), "bad": len(bad)}), flush=True)
print(json.dumps({"cases": ncases, "seed0": seed0, "mismatches": len(bad), "detail": bad[:10], "wall_s": round(time.time() - t0, 1)}))
