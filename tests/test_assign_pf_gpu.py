"""k-means assign behind the bf16 prefilter (qk_assign_pf.hip: n >= 40960 rows, d % 8 == 0, d <= 128, m >= 64) against the oracle
(qo_kmeans_assign: clustering.cpp:51-66, :149-159 -- IndexFlat::search(k = 1) / batched_scan_list(k = 1)) AND against the fp32 MFMA
kernel it stands in for (k_assign answers calls of fewer than 40960 rows: the same rows in two halves).  Assignments and distance
bits must be equal: every key that decides goes through the exact chain, the prefilter only decides which keys are computed.
The shapes aim at what a filter can get wrong: ties (duplicate centroids, duplicate rows, distance 0 where the expanded form clamps),
rows at the same distance from many centroids, scales far from 1, ragged n / m / d, non-finite values."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    yield c
    c.close()


def _halves(ctx, x, c, metric):
    h = x.shape[0] // 2
    assert h < 40960 and x.shape[0] - h < 40960
    a0, v0 = ctx.kmeans_assign(x[:h], c, metric)
    a1, v1 = ctx.kmeans_assign(x[h:], c, metric)
    return np.concatenate([a0, a1]), np.concatenate([v0, v1])


def _check(ctx, x, c, metric, oracle=True):
    ga, gv = ctx.kmeans_assign(x, c, metric)
    na, _ = ctx.kmeans_assign(x, c, metric, values=False)   # (val = NULL: rows with one candidate take it without an exact key)
    np.testing.assert_array_equal(na, ga)
    if x.shape[0] < 81920:
        ha, hv = _halves(ctx, x, c, metric)
        np.testing.assert_array_equal(ga, ha)
        np.testing.assert_array_equal(gv.view(np.uint32), hv.view(np.uint32))
    if oracle:
        oa, ov = O.kmeans_assign(x, c, metric)
        np.testing.assert_array_equal(ga, oa)
        np.testing.assert_array_equal(gv.view(np.uint32), ov.view(np.uint32))
    return ga


def _mixture(rng, n, m, d, sigma=0.3):
    c = rng.standard_normal((m, d)).astype(np.float32)
    x = (c[rng.integers(0, m, n)] + sigma * rng.standard_normal((n, d))).astype(np.float32)
    return x, c


@pytest.mark.parametrize("n,m,d,metric", [
    (65536, 64, 128, "l2"),
    (40960, 300, 128, "l2"),
    (70001, 1000, 128, "l2"),
    (70001, 1000, 128, "ip"),
    (66000, 4096, 128, "l2"),
    (66000, 257, 64, "l2"),
    (66000, 300, 8, "l2"),
    (66000, 300, 24, "ip"),
    (67000, 777, 120, "l2"),   # (a last 16-column block that is half zeros)
    (67000, 100, 72, "ip"),
    (66000, 4100, 32, "l2"),   # (a last chunk of centroid tiles that is mostly padding)
])
def test_mixture(ctx, n, m, d, metric):
    rng = np.random.default_rng(n + m + d)
    x, c = _mixture(rng, n, m, d)
    _check(ctx, x, c, metric)


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_no_structure(ctx, metric):
    # every centroid about equally far: the filter keeps many candidates per row
    rng = np.random.default_rng(3)
    x = rng.standard_normal((66000, 128)).astype(np.float32)
    c = rng.standard_normal((2048, 128)).astype(np.float32) * 0.2
    _check(ctx, x, c, metric)


def test_rows_that_are_centroids_and_duplicate_centroids(ctx):
    # distance 0 (the expanded form gives a small value of either sign, clamped at +0) and exact ties: the smaller index wins
    rng = np.random.default_rng(4)
    x = rng.standard_normal((70000, 64)).astype(np.float32) * 3.0
    c = x[rng.integers(0, 70000, 512)].copy()
    c[100:200] = c[300:400]                      # duplicates: 100 + i and 300 + i
    c[511] = c[0]
    a = _check(ctx, x, c, "l2")
    assert not np.isin(a, np.arange(300, 400)).any() and not (a == 511).any()
    _check(ctx, x, c, "ip")


def test_all_rows_equal_and_all_centroids_equal(ctx):
    x = np.full((66000, 32), 0.37, np.float32)
    c = np.full((128, 32), 0.37, np.float32)
    a = _check(ctx, x, c, "l2")
    assert (a == 0).all()
    c2 = np.random.default_rng(5).standard_normal((128, 32)).astype(np.float32)
    _check(ctx, x, c2, "l2")
    _check(ctx, x, c2, "ip")


@pytest.mark.parametrize("scale", [1e-4, 1e3, 1e15])
def test_scales(ctx, scale):
    rng = np.random.default_rng(6)
    x, c = _mixture(rng, 66000, 500, 128)
    _check(ctx, (x * scale).astype(np.float32), (c * scale).astype(np.float32), "l2")
    _check(ctx, (x * scale).astype(np.float32), (c * scale).astype(np.float32), "ip")


def test_norms_that_differ_by_orders_of_magnitude(ctx):
    # the margin of the filter comes from the LARGEST centroid norm: one huge centroid makes everything a candidate for rows near it
    rng = np.random.default_rng(7)
    x, c = _mixture(rng, 66000, 256, 64)
    c[7] *= 1e4
    c[8] *= 1e-4
    x[::1000] *= 1e4
    _check(ctx, x, c, "l2")
    _check(ctx, x, c, "ip")


def test_non_finite_values_agree_with_the_fp32_kernel(ctx):
    rng = np.random.default_rng(8)
    x, c = _mixture(rng, 66000, 256, 64)
    x[5, 3] = np.nan
    x[6, 0] = np.inf
    x[7, 63] = -np.inf
    c[9, 1] = np.nan
    c[10, 2] = np.inf
    with np.errstate(all="ignore"):
        _check(ctx, x, c, "l2", oracle=False)
        _check(ctx, x, c, "ip", oracle=False)


def test_wide_workgroups(ctx):
    # 262144+ rows: 1024-row workgroups (eight row tiles per wave)
    rng = np.random.default_rng(9)
    x, c = _mixture(rng, 263000, 1000, 128)
    _check(ctx, x, c, "l2")
    x, c = _mixture(rng, 263000, 300, 64)
    _check(ctx, x, c, "ip")


def test_kmeans_driver_uses_it_and_stays_equal_to_the_oracle(ctx):
    rng = np.random.default_rng(10)
    x, _ = _mixture(rng, 80000, 100, 64)
    c, a, _ = ctx.kmeans(x.copy(), 100, "l2", niter=3, seed=11)
    oc, oa, _ = O.kmeans(x.copy(), 100, "l2", niter=3, seed=11)
    np.testing.assert_array_equal(a, oa)
    np.testing.assert_array_equal(c.view(np.uint32), oc.view(np.uint32))


# ---- the nearest-list search of many rows (qk_dense_device, k = 1: PartitionManager::add's parent search) through the same kernels ----
def _parent(ctx, cent, ids):
    from quake_amd.capi import Store
    p = Store(ctx, cent.shape[1])
    p.build_csr(np.array([0, cent.shape[0]], np.int64), ids, cent)
    return p


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_nearest_list_of_many_rows(ctx, metric):
    """coarse(nprobe = 1) over 70000 rows: ids that are NOT the row numbers (a parent after splits and deletes), duplicate centroids
    (equal keys: the smaller ID wins, not the smaller row), against the fp32 argmin (the same rows in halves) and the oracle."""
    rng = np.random.default_rng(21)
    x, c = _mixture(rng, 70000, 600, 64)
    c[100:150] = c[400:450]                               # duplicates
    ids = rng.permutation(5000)[:600].astype(np.int64)    # arbitrary ids: of a duplicate pair either row may hold the smaller one
    p = _parent(ctx, c, ids)
    gp, gd = ctx.coarse(p, x, 1, metric)
    assert ctx.last_scan_kernel() == "k_assign_pf"
    h = 35000
    p0, d0 = ctx.coarse(p, x[:h], 1, metric)
    assert ctx.last_scan_kernel() != "k_assign_pf"
    p1, d1 = ctx.coarse(p, x[h:], 1, metric)
    np.testing.assert_array_equal(gp, np.concatenate([p0, p1]))
    np.testing.assert_array_equal(gd.view(np.uint32), np.concatenate([d0, d1]).view(np.uint32))
    op, od = O.coarse(x, c, ids, 1, metric, num_threads=0)
    np.testing.assert_array_equal(gp, op)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    p.close()


def test_nearest_list_through_search(ctx):
    """search(nprobe = 1) of a 66000-query batch: the coarse step goes through the prefiltered kernels, the answer stays the oracle's"""
    from helpers import make_ivf, make_queries
    from quake_amd.capi import Store
    ivf = make_ivf(200000, 32, 128, seed=31)
    s = Store(ctx, 32)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    p = _parent(ctx, ivf["centroids"], np.arange(128, dtype=np.int64))
    q = make_queries(66000, 32, seed=32, like=ivf["x"])
    gi, gd = ctx.search(p, s, q, 1, 10, "l2")
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 1, 10, "l2", batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    s.close()
    p.close()


def test_rows_that_are_not_16_byte_aligned_take_the_fp32_kernel(ctx):
    """the prefiltered kernels read rows as float4: a base pointer off by one float must not reach them (and the answer stays right)"""
    import torch
    rng = np.random.default_rng(41)
    x, c = _mixture(rng, 50000, 200, 64)
    buf = torch.empty(50000 * 64 + 1, dtype=torch.float32, device="cuda")
    xt = buf[1:].view(50000, 64)
    xt.copy_(torch.from_numpy(x))
    ct = torch.from_numpy(c).cuda()
    assert xt.data_ptr() % 16 == 4
    torch.cuda.synchronize()
    a, v = ctx.kmeans_assign(xt, ct, "l2")
    ctx.synchronize()
    torch.cuda.synchronize()
    oa, ov = O.kmeans_assign(x, c, "l2")
    np.testing.assert_array_equal(a.cpu().numpy(), oa)
    np.testing.assert_array_equal(v.cpu().numpy().view(np.uint32), ov.view(np.uint32))
