"""Seeded sweep over shapes nobody picked by hand: dimension (1 ... 200, multiples of 16 and not), number of lists, skewed list
sizes with empty lists and lists shorter than k, batch sizes on both sides of every dispatch boundary (one-launch search up
to 32 queries; single-workgroup grouping up to 1024 pairs; row-per-lane / 16x16 / query-sharing scan forms), nprobe up to
and beyond the number of lists, k from 1 to 100, both metrics, SIFT-like integer data (dense ties).  Every case: qk_search
ids and float32 distance bits equal the oracle's canonical batched search; the same batch again as qk_coarse + qk_scan."""
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def _case(rng):
    d = int(rng.choice([1, 3, 16, 17, 31, 32, 48, 64, 100, 128, 129, 200]))
    nlist = int(rng.choice([1, 2, 5, 17, 64, 150, 300]))
    metric = str(rng.choice(["l2", "ip"]))
    integer = bool(rng.random() < 0.3)
    n = int(rng.choice([nlist, 200, 3000, 20000]))
    # skewed sizes: a few big lists, many small, some empty
    w = rng.random(nlist) ** 3
    w[rng.random(nlist) < 0.15] = 0.0
    if w.sum() == 0:
        w[0] = 1.0
    assign = rng.choice(nlist, size=n, p=w / w.sum())
    if integer:
        cent = rng.integers(0, 30, size=(nlist, d)).astype(np.float32)
        x = cent[assign] + rng.integers(-2, 3, size=(n, d)).astype(np.float32)
    else:
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        x = (cent[assign] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    if metric == "ip" and not integer:
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6)
    ids = rng.permutation(n).astype(np.int64) + int(rng.integers(0, 1000))
    order = np.argsort(assign, kind="stable")
    vecs, aids = np.ascontiguousarray(x[order]), np.ascontiguousarray(ids[order])
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(np.bincount(assign, minlength=nlist))
    Q = int(rng.choice([1, 2, 7, 16, 32, 33, 64, 100, 257, 1030]))
    nprobe = int(rng.choice([1, 2, 3, 8, 16, 40, 64, 100]))
    k = int(rng.choice([1, 5, 10, 32, 33, 100]))
    q = (x[rng.integers(0, n, size=Q)] + (0 if integer else 0.05) * rng.standard_normal((Q, d))).astype(np.float32)
    return dict(d=d, nlist=nlist, metric=metric, integer=integer, n=n, cent=cent, vecs=vecs, ids=aids, offsets=offsets, Q=Q,
                nprobe=nprobe, k=k, q=q)


# (a one-off run of 1000 seeds found one failure -- seed 107: d = 128, k = 32, nprobe 100: the row-per-lane form asked for more
#  LDS than a CU has -- and passes since; 107 stays in the default set)
@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("QK_RANDOM_SHAPES", "48")))) | {107}))
def test_random_shape_search_bit_exact(seed):
    from quake_amd.capi import Context, Store
    rng = np.random.default_rng(1000 + seed)
    c = _case(rng)
    ctx = Context(0)
    try:
        s = Store(ctx, c["d"])
        s.build_csr(c["offsets"], c["ids"], c["vecs"])
        parent = Store(ctx, c["d"])
        parent.build_csr(np.array([0, c["nlist"]], np.int64), np.arange(c["nlist"], dtype=np.int64), c["cent"])
        tag = {kk: c[kk] for kk in ("d", "nlist", "metric", "integer", "n", "Q", "nprobe", "k")}
        oi, od = O.search(c["q"], c["cent"], c["vecs"], c["ids"], c["offsets"], c["nprobe"], c["k"], c["metric"], batched_scan=True)
        for rep in range(2):  # twice: per-context state left by the first call (tickets, counters, cleared regions)
            gi, gd = ctx.search(parent, s, c["q"], c["nprobe"], c["k"], c["metric"])
            np.testing.assert_array_equal(gi, oi, err_msg=f"{tag} form={ctx.last_scan_kernel()} rep={rep}")
            np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32), err_msg=str(tag))
        # the two-call form: coarse, then scan of the returned lists
        pids, _ = ctx.coarse(parent, c["q"], c["nprobe"], c["metric"])
        op, _ = O.coarse(c["q"], c["cent"], None, c["nprobe"], c["metric"])
        np.testing.assert_array_equal(pids[:, :op.shape[1]], op, err_msg=str(tag))
        gi, gd = ctx.scan(s, c["q"], pids, c["k"], c["metric"])
        np.testing.assert_array_equal(gi, oi, err_msg=f"{tag} (coarse + scan) form={ctx.last_scan_kernel()}")
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
        if seed % 3 == 0:  # the same batch handed over in device memory (no staging, results written in place)
            import torch
            ti, td = ctx.search(parent, s, torch.from_numpy(c["q"]).cuda(), c["nprobe"], c["k"], c["metric"])
            ctx.synchronize()
            np.testing.assert_array_equal(ti.cpu().numpy(), oi, err_msg=f"{tag} (device buffers)")
            np.testing.assert_array_equal(td.cpu().numpy().view(np.uint32), od.view(np.uint32))
        # no parent: every list is scanned (query_coordinator.cpp:624-626)
        if c["Q"] * c["nlist"] <= 40000:
            fi, fd = ctx.search(None, s, c["q"], 1, c["k"], c["metric"])
            allp = np.broadcast_to(np.arange(c["nlist"], dtype=np.int64), (c["Q"], c["nlist"])).copy()
            ei, ed = O.batched_serial_scan(c["q"], c["vecs"], c["ids"], c["offsets"], allp, c["k"], c["metric"])
            np.testing.assert_array_equal(fi, ei, err_msg=f"{tag} (flat) form={ctx.last_scan_kernel()}")
            np.testing.assert_array_equal(fd.view(np.uint32), ed.view(np.uint32))
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_shape_wide_k_and_nprobe(seed):
    """the wide ends: k beyond the LDS pools (key emission + bisection select) and nprobe in the hundreds / thousands"""
    from quake_amd.capi import Context, Store
    rng = np.random.default_rng(5000 + seed)
    d = int(rng.choice([8, 24, 40]))
    nlist = int(rng.choice([600, 2500]))
    n = 30000
    metric = str(rng.choice(["l2", "ip"]))
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    assign = rng.integers(0, nlist, size=n)
    x = (cent[assign] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    ids = rng.permutation(n).astype(np.int64)
    order = np.argsort(assign, kind="stable")
    vecs, aids = np.ascontiguousarray(x[order]), np.ascontiguousarray(ids[order])
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(np.bincount(assign, minlength=nlist))
    Q = int(rng.choice([3, 40]))
    nprobe = int(rng.choice([100, 500, 2000]))
    k = int(rng.choice([100, 449, 1000, 5000]))
    q = (x[rng.integers(0, n, size=Q)] + 0.05 * rng.standard_normal((Q, d))).astype(np.float32)
    ctx = Context(0)
    try:
        s = Store(ctx, d)
        s.build_csr(offsets, aids, vecs)
        parent = Store(ctx, d)
        parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), cent)
        oi, od = O.search(q, cent, vecs, aids, offsets, nprobe, k, metric, batched_scan=True)
        gi, gd = ctx.search(parent, s, q, nprobe, k, metric)
        tag = dict(d=d, nlist=nlist, metric=metric, Q=Q, nprobe=nprobe, k=k)
        np.testing.assert_array_equal(gi, oi, err_msg=str(tag))
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32), err_msg=str(tag))
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_shape_kmeans_bit_exact(seed):
    """k-means driver (subsample or not, empty-cluster splits when m is close to n) and the refine of a store, random shapes"""
    from quake_amd.capi import Context
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([2, 17, 32, 100, 128, 160]))
    m = int(rng.choice([1, 2, 7, 33, 200]))
    n = int(rng.choice([max(m, 50), 2000, 30000]))
    metric = str(rng.choice(["l2", "ip"]))
    x = (rng.standard_normal((max(m // 4, 1), d)).astype(np.float32)[rng.integers(0, max(m // 4, 1), n)]
         + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    ctx = Context(0)
    try:
        tag = dict(d=d, m=m, n=n, metric=metric)
        gc, ga, gx = ctx.kmeans(x.copy(), m, metric, niter=3, seed=99 + seed)
        oc, oa, ox = O.kmeans(x, m, metric, niter=3, seed=99 + seed)
        np.testing.assert_array_equal(np.asarray(gc).view(np.uint32), oc.view(np.uint32), err_msg=str(tag))
        np.testing.assert_array_equal(np.asarray(ga), oa, err_msg=str(tag))
        np.testing.assert_array_equal(np.asarray(gx).view(np.uint32), ox.view(np.uint32), err_msg=str(tag))
        a, v = ctx.kmeans_assign(x, oc, metric)
        ra, rv = O.kmeans_assign(x, oc, metric)
        np.testing.assert_array_equal(np.asarray(a), ra, err_msg=str(tag))
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("QK_RANDOM_APS", "16")))))
def test_random_shape_aps_bit_exact(seed):
    """recall-target search on random shapes: ids, distance bits AND the number of partitions each query visited equal the
    oracle's walk"""
    from quake_amd.capi import Context, Store
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.choice([8, 33, 64, 128]))
    nlist = int(rng.choice([8, 40, 200]))
    n = int(rng.choice([2000, 30000]))
    metric = str(rng.choice(["l2", "ip"]))
    cent = rng.standard_normal((nlist, d)).astype(np.float32)
    w = rng.random(nlist) ** 2 + 1e-3
    assign = rng.choice(nlist, size=n, p=w / w.sum())
    x = (cent[assign] + 0.6 * rng.standard_normal((n, d))).astype(np.float32)
    if metric == "ip":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    ids = rng.permutation(n).astype(np.int64)
    order = np.argsort(assign, kind="stable")
    vecs, aids = np.ascontiguousarray(x[order]), np.ascontiguousarray(ids[order])
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(np.bincount(assign, minlength=nlist))
    Q = int(rng.choice([1, 9, 70, 300]))
    q = (x[rng.integers(0, n, size=Q)] + 0.1 * rng.standard_normal((Q, d))).astype(np.float32)
    k = int(rng.choice([1, 10, 60]))
    rt = float(rng.choice([0.5, 0.8, 0.9, 0.99]))
    frac = float(rng.choice([0.3, 0.6, 1.0]))
    thr = float(rng.choice([0.0, 0.001, 0.05]))
    pre = bool(rng.random() < 0.5)
    if int(nlist * frac) < 2:
        frac = 1.0
    ctx = Context(0)
    try:
        s = Store(ctx, d)
        s.build_csr(offsets, aids, vecs)
        parent = Store(ctx, d)
        parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), cent)
        tag = dict(d=d, nlist=nlist, n=n, metric=metric, Q=Q, k=k, rt=rt, frac=frac, thr=thr, pre=pre)
        gi, gd, gn = ctx.search_aps(parent, s, q, k, metric, rt, recompute_threshold=thr, use_precomputed=pre, initial_search_fraction=frac)
        oi, od, on = O.search_aps(q, cent, vecs, aids, offsets, k, metric, rt, recompute_threshold=thr, use_precomputed=pre,
                                  initial_search_fraction=frac, expanded=True, num_threads=8)
        np.testing.assert_array_equal(gn, on, err_msg=str(tag))
        np.testing.assert_array_equal(gi, oi, err_msg=str(tag))
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32), err_msg=str(tag))
    finally:
        ctx.close()
