"""Seeded stress of the prefiltered k-means assign (qk_assign_pf.hip: n >= 40960) against the fp32 MFMA kernel it stands in for
(k_assign answers calls under 40960 rows: the same rows in slices), on shapes and data nobody picked by hand: random n / m / d,
mixtures, structureless data, scales, duplicate centroids, rows that are centroids, heavy-tailed norms; both metrics; with and
without distances; every fourth case also as the nearest-list search of a parent store (coarse, nprobe 1) with permuted ids.
python scripts/stress_assign_pf.py [n_cases] [seed0]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quake_amd.capi import Context, Store

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
bad = []
t0 = time.time()
for c in range(ncases):
    g = torch.Generator(device=dev); g.manual_seed(seed0 + c)
    rng = np.random.default_rng(seed0 + c)
    n = int(rng.choice([65536, 65537, 70001, 131072, 200003, 262144, 300000]))
    m = int(rng.choice([64, 65, 100, 255, 256, 1000, 4095, 4096, 4111, 9000]))
    d = int(rng.choice([8, 16, 24, 32, 64, 72, 96, 120, 128]))
    kind = str(rng.choice(["mixture", "flat", "scaled", "dups", "rows", "tails"]))
    metric = str(rng.choice(["l2", "ip"]))
    cen = torch.randn(m, d, device=dev, generator=g)
    if kind == "flat":
        x = torch.randn(n, d, device=dev, generator=g); cen *= 0.2
    else:
        pick = torch.randint(0, m, (n,), device=dev, generator=g)
        x = cen[pick] + 0.3 * torch.randn(n, d, device=dev, generator=g)
    if kind == "scaled":
        s = float(10.0 ** rng.integers(-5, 12)); x = x * s; cen = cen * s
    if kind == "dups":
        cen[m // 2:] = cen[: m - m // 2].clone()
    if kind == "rows":
        cen = x[torch.randint(0, n, (m,), device=dev, generator=g)].clone()
    if kind == "tails":
        cen = cen * torch.exp(2.0 * torch.randn(m, 1, device=dev, generator=g)); x = x * torch.exp(torch.randn(n, 1, device=dev, generator=g))
    x = x.contiguous(); cen = cen.contiguous()
    ga, gv = ctx.kmeans_assign(x, cen, metric)
    na, _ = ctx.kmeans_assign(x, cen, metric, values=False)
    ra, rv = [], []
    for lo in range(0, n, 30000):
        a, v = ctx.kmeans_assign(x[lo:lo + 30000].contiguous(), cen, metric)
        ra.append(a); rv.append(v)
    ra, rv = torch.cat(ra), torch.cat(rv)
    torch.cuda.synchronize()
    ok = torch.equal(ga, ra) and torch.equal(na, ra) and torch.equal(gv.view(torch.int32), rv.view(torch.int32))
    if c % 4 == 0:  # the nearest-list search through the same kernels: a parent store whose ids are not the row numbers
        ids = torch.from_numpy(rng.permutation(4 * m)[:m].astype(np.int64))
        par = Store(ctx, d)
        par.build_csr(np.array([0, m], np.int64), ids.numpy(), cen.cpu().numpy())
        gp, gd = ctx.coarse(par, x, 1, metric)
        big = ctx.last_scan_kernel() == "k_assign_pf"
        sp, sd = [], []
        for lo in range(0, n, 30000):
            pp, dd = ctx.coarse(par, x[lo:lo + 30000].contiguous(), 1, metric)
            sp.append(pp); sd.append(dd)
        torch.cuda.synchronize()
        ok = ok and big and torch.equal(gp, torch.cat(sp)) and torch.equal(gd.view(torch.int32), torch.cat(sd).view(torch.int32))
        par.close()
    if not ok:
        bad.append((seed0 + c, n, m, d, kind, metric, int((ga != ra).sum()), int((na != ra).sum())))
    print(json.dumps({"case": c, "n": n, "m": m, "d": d, "kind": kind, "metric": metric, "ok": bool(ok)}), flush=True)
print(json.dumps({"cases": ncases, "seed0": seed0, "mismatches": len(bad), "detail": bad[:10], "wall_s": round(time.time() - t0, 1)}))
