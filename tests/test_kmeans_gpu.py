"""GPU k-means kernels vs the oracle (clustering.cpp:13-182 restated in oracle/quake_oracle.c).
assign: bit-exact indices and values; update: bit-exact fp32 sums (same ascending-row order); Lloyd driver:
identical centroids and assignments (same documented init / split rules).  FAISS parity itself is unpinned."""
import numpy as np
import pytest

import oracle as O
from helpers import make_ivf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("n,m,d", [(5000, 37, 32), (3000, 256, 128), (700, 5, 4), (2000, 64, 100)])
def test_assign_bit_exact(ctx, metric, n, m, d):
    rng = np.random.default_rng(n + m + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal((m, d)).astype(np.float32)
    ga, gv = ctx.kmeans_assign(x, c, metric)
    oa, ov = O.kmeans_assign(x, c, metric)
    np.testing.assert_array_equal(ga, oa)
    np.testing.assert_array_equal(gv.view(np.uint32), ov.view(np.uint32))


def test_assign_ties_take_lower_index(ctx):
    x = np.zeros((40, 8), np.float32)
    c = np.ones((10, 8), np.float32)  # all centroids equidistant
    ga, _ = ctx.kmeans_assign(x, c, "l2")
    assert (ga == 0).all()
    c[3] = 0
    c[7] = 0
    ga, _ = ctx.kmeans_assign(x, c, "l2")
    assert (ga == 3).all()


def test_accumulate_bit_exact(ctx):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((20000, 48)).astype(np.float32)
    a = rng.integers(0, 29, size=20000).astype(np.int64)
    a[a == 11] = 12  # an empty cluster
    gs, gc = ctx.kmeans_accumulate(x, a, 29)
    os_, oc = O.kmeans_accumulate(x, a, 29)
    np.testing.assert_array_equal(gc, oc)
    np.testing.assert_array_equal(gs.view(np.uint32), os_.view(np.uint32))
    assert gc[11] == 0


@pytest.mark.parametrize("blocked", [False, True])
@pytest.mark.parametrize("m", [1, 7, 255, 256, 300, 4095, 4096, 65535, 65536, 70000])
def test_accumulate_bucket_passes(ctx, m, blocked):
    """the stable bucketing of rows by assignment (k_rs_hist / k_rs_scan / k_rs_scatter): one pass of 8 bits up to 255 centroids,
    one of 12 bits up to 4095 (the segment bounds then come from the scan itself), two beyond; chunks with a ragged end, out-of-range
    assignments (ignored, as the oracle does), one centroid that owns a third of the rows -- sums are order-sensitive fp32, so a
    bucketing that is not stable fails here, under the reference's row-after-row order and under the blocked one"""
    rng = np.random.default_rng(m)
    n, d = 70001, 24
    x = (rng.standard_normal((n, d)) * 10.0 ** rng.integers(-3, 4, size=(n, 1))).astype(np.float32)
    a = rng.integers(0, m, size=n).astype(np.int64)
    a[rng.integers(0, n, size=n // 3)] = m // 2   # a long segment: many rounds of every chunk carry the same digit
    a[rng.integers(0, n, size=50)] = -1
    a[rng.integers(0, n, size=50)] = m + 3
    gs, gc = ctx.kmeans_accumulate(x, a, m, blocked=blocked)
    os_, oc = O.kmeans_accumulate(x, a, m, blocked=blocked)
    np.testing.assert_array_equal(gc, oc)
    np.testing.assert_array_equal(gs.view(np.uint32), os_.view(np.uint32))


@pytest.mark.parametrize("n,m,d,big", [(20000, 29, 48, 0), (60000, 5, 128, 30000), (9000, 3, 100, 5000), (40000, 17, 6, 20000),
                                       (12000, 4, 768, 9000), (3000, 2, 1100, 2500), (1024 * 9 + 5, 1, 32, 0), (1023, 1, 8, 0)])
def test_accumulate_blocked_order(ctx, n, m, d, big):
    """qk_kmeans_accumulate_blocked == the oracle's blocked order bit for bit: clusters below one block, of exactly a group, of
    more groups than the launch has group lanes (the last workgroup to arrive folds the group partials), rows wider than one
    float4 per thread (d = 1100) and not a multiple of four (d = 6, 100 ... no: 100 is), empty clusters; and the order really is
    another one than the row-after-row sum (it must differ from it somewhere on data with spread magnitudes)."""
    rng = np.random.default_rng(n + d)
    x = (rng.standard_normal((n, d)) * 10.0 ** rng.integers(-2, 3, size=(n, 1))).astype(np.float32)
    a = rng.integers(0, m, size=n).astype(np.int64)
    if big:
        a[rng.permutation(n)[:big]] = m - 1
    if m > 3:
        a[a == 1] = 2  # an empty cluster
    gs, gc = ctx.kmeans_accumulate(x, a, m, blocked=True)
    os_, oc = O.kmeans_accumulate(x, a, m, blocked=True)
    np.testing.assert_array_equal(gc, oc)
    np.testing.assert_array_equal(gs.view(np.uint32), os_.view(np.uint32))
    ser, _ = O.kmeans_accumulate(x, a, m)
    np.testing.assert_allclose(gs, ser, rtol=1e-3, atol=1e-2)
    if n >= 9000:
        assert (gs.view(np.uint32) != ser.view(np.uint32)).any()
    # device buffers in and out, twice on one context (the tickets of the multi-group fold must be back at zero)
    import torch
    xd, ad = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
    for _ in range(2):
        ds, dc = ctx.kmeans_accumulate(xd, ad, m, blocked=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(ds.cpu().numpy().view(np.uint32), os_.view(np.uint32))
        np.testing.assert_array_equal(dc.cpu().numpy(), oc)


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_kmeans_driver_matches_oracle(ctx, metric):
    ivf = make_ivf(30000, 32, 24, seed=3, metric=metric)
    x = ivf["x"]
    gc, ga, gx = ctx.kmeans(x, 24, metric, niter=5, seed=1234)
    oc, oa, ox = O.kmeans(x, 24, metric, niter=5, seed=1234)
    np.testing.assert_array_equal(gx.view(np.uint32), ox.view(np.uint32))
    np.testing.assert_array_equal(gc.view(np.uint32), oc.view(np.uint32))
    np.testing.assert_array_equal(ga, oa)
    # sanity: a real clustering (every list non-empty, sizes sum to n)
    assert np.bincount(ga, minlength=24).min() > 0


def test_kmeans_subsample_and_empty_split(ctx):
    # n > 256*m triggers the FAISS-style subsample; duplicated points force empty clusters -> split path
    rng = np.random.default_rng(5)
    base = rng.standard_normal((6, 16)).astype(np.float32)
    x = np.repeat(base, 500, axis=0)  # 3000 rows, only 6 distinct points, m = 8 > 6 -> empties
    x += 1e-3 * rng.standard_normal(x.shape).astype(np.float32)
    gc, ga, _ = ctx.kmeans(x, 8, "l2", niter=4, seed=7)
    oc, oa, _ = O.kmeans(x, 8, "l2", niter=4, seed=7)
    np.testing.assert_array_equal(gc.view(np.uint32), oc.view(np.uint32))
    np.testing.assert_array_equal(ga, oa)
    x2 = rng.standard_normal((3000, 8)).astype(np.float32)  # 3000 > 256*10 -> subsample
    gc, ga, _ = ctx.kmeans(x2, 10, "l2", niter=3, seed=9)
    oc, oa, _ = O.kmeans(x2, 10, "l2", niter=3, seed=9)
    np.testing.assert_array_equal(gc.view(np.uint32), oc.view(np.uint32))
    np.testing.assert_array_equal(ga, oa)


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("iters", [0, 1, 3])
def test_refine_partitions_matches_oracle(ctx, metric, iters):
    """kmeans_refine_partitions (clustering.cpp:99-182) on the device store: centroids and the refined partitions
    (ids in append order, vectors) equal the oracle's; with 0 iterations the sizes only move by re-assignment
    (test/cpp/partition_manager.cpp:121-167)."""
    from quake_amd.capi import Store
    ivf = make_ivf(6000, 24, 12, seed=21, metric=metric)
    s = Store(ctx, 24)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    sel = np.array([7, 2, 9, 4, 11], np.int64)  # a subset, in a non-sorted order
    # slightly off-centre centroids so that vectors really move
    rng = np.random.default_rng(5)
    cent = (ivf["centroids"][sel] + 0.05 * rng.standard_normal((5, 24))).astype(np.float32)
    pv = [ivf["part_vecs"][p] for p in sel]
    pi = [ivf["part_ids"][p] for p in sel]
    vecs, ids, offs = O.csr_from_partitions(pv, pi, 24)
    oc, ov, oi, oo = O.kmeans_refine_partitions(cent, vecs, ids, offs, metric, iters)
    gc = s.refine_lists(sel, cent, metric, iters)
    np.testing.assert_array_equal(gc.view(np.uint32), oc.view(np.uint32))
    total = 0
    for c, p in enumerate(sel):
        gv, gi = s.get_list(int(p))
        np.testing.assert_array_equal(gi, oi[oo[c]:oo[c + 1]])
        np.testing.assert_array_equal(gv, ov[oo[c]:oo[c + 1]])
        total += len(gi)
    assert total == len(ids) and s.ntotal() == 6000
    # untouched partitions are untouched
    gv, gi = s.get_list(0)
    np.testing.assert_array_equal(gi, ivf["part_ids"][0])
