"""The coarse step without a key matrix (qk_dense_pf.hip: approximate keys on bf16 MFMA bound which rows can be among a query's
k nearest, the exact k-ordered fmaf chain of those candidates is the answer) against the oracle's parent search
(query_coordinator.cpp:628-644 -> batched_scan_list, list_scanning.h:313-366): ids and float32 distance bits."""
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


def _parent(ctx, cent, ids=None):
    from quake_amd.capi import Store
    n, d = cent.shape
    p = Store(ctx, d)
    p.build_csr(np.array([0, n], np.int64), np.arange(n, dtype=np.int64) if ids is None else ids, cent)
    return p


def _check(ctx, parent, cent, q, k, metric, ids=None):
    gp, gd = ctx.coarse(parent, q, k, metric)
    op, od = O.coarse(q, cent, ids, k, metric)
    np.testing.assert_array_equal(gp, op)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("d", [128, 100, 32])
def test_coarse_prefiltered_matches_oracle(ctx, metric, d):
    rng = np.random.default_rng(7 + d)
    for n in (1024, 4096, 5000, 20000):
        cent = rng.standard_normal((n, d)).astype(np.float32)
        if metric == "ip":
            cent /= np.linalg.norm(cent, axis=1, keepdims=True)
        parent = _parent(ctx, cent)
        for nq in (64, 100, 1024):
            q = (cent[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
            if metric == "ip":
                q /= np.linalg.norm(q, axis=1, keepdims=True)
            for k in (2, 10, 32, 64) + ((100, 128, 150, 192) if n > 8192 else ()):  # (64 < k <= 192: the form answers from 8193 rows on)
                _check(ctx, parent, cent, q, k, metric)
                if k > 64:
                    assert ctx.last_scan_kernel() == "k_dense_pf", ctx.last_scan_kernel()
        parent.close()


def test_coarse_prefiltered_65536_centroids(ctx):
    rng = np.random.default_rng(3)
    cent = rng.standard_normal((65536, 128)).astype(np.float32)
    parent = _parent(ctx, cent)
    q = (cent[rng.integers(0, 65536, 512)] + 0.5 * rng.standard_normal((512, 128))).astype(np.float32)
    _check(ctx, parent, cent, q, 32, "l2")
    _check(ctx, parent, cent, q, 1, "l2")   # the nearest centroid alone takes this form from 32768 rows on
    _check(ctx, parent, cent, q, 100, "l2")
    for nq in (1, 7, 40):  # few queries against a long list take this form too (from 32768 rows on)
        for k in (1, 10, 100):
            _check(ctx, parent, cent, q[:nq], k, "l2")
            assert ctx.last_scan_kernel() == "k_dense_pf", ctx.last_scan_kernel()
    cent[2000:2300] = cent[1000]  # 300 identical rows: at k = 128 the cut falls inside the tie (ordered by id)
    parent2 = _parent(ctx, cent)
    q[:8] = cent[1000] + 1e-3
    _check(ctx, parent2, cent, q, 128, "l2")
    _check(ctx, parent2, cent, q, 192, "l2")
    parent2.close()
    parent.close()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_nearest_centroid_of_many_and_the_search_behind_it(ctx, metric):
    """k = 1 over >= 32768 rows: the prefiltered form instead of the fused fp32 argmin -- alone (d = 100: padded columns), with exact
    duplicates of the nearest row (ties go to the smaller id), and as the coarse step of a search with nprobe = 1, whose grouping
    then reads plain list numbers instead of the argmin's packed (key, id) words"""
    from quake_amd.capi import Store
    from helpers import make_ivf, make_queries
    rng = np.random.default_rng(5)
    n, d = 40000, 100
    cent = rng.standard_normal((n, d)).astype(np.float32)
    cent[20000:20050] = cent[100:150]       # duplicates: the same key under two ids
    if metric == "ip":
        cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    parent = _parent(ctx, cent)
    q = np.concatenate([cent[100:150] * np.float32(1.0), (cent[rng.integers(0, n, 250)] + 0.2 * rng.standard_normal((250, d))).astype(np.float32)])
    if metric == "ip":
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = np.ascontiguousarray(q, np.float32)
    _check(ctx, parent, cent, q, 1, metric)
    gp, _ = ctx.coarse(parent, q, 1, metric)
    assert (gp[:50, 0] == np.arange(100, 150)).all()
    parent.close()
    ivf = make_ivf(70000, 32, 33000, seed=9, metric=metric)
    s = Store(ctx, 32)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    par = _parent(ctx, ivf["centroids"])
    qq = make_queries(300, 32, seed=10, like=ivf["x"], metric=metric)
    gi, gd = ctx.search(par, s, qq, 1, 5, metric)
    oi, od = O.search(qq, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 1, 5, metric, batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    s.close()
    par.close()


def test_ties_duplicates_and_candidate_overflow(ctx):
    """exact duplicates straddling the k-th place (the (key, id) order decides), shuffled ids, and a list in which every row is
    a candidate of every query -- the candidate lists overflow and the finish walks all rows"""
    rng = np.random.default_rng(11)
    base = rng.integers(0, 6, size=(300, 64)).astype(np.float32)     # SIFT-like integers: exact arithmetic, many ties
    cent = base[rng.integers(0, 300, 6000)]
    ids = rng.permutation(6000).astype(np.int64) + 17
    parent = _parent(ctx, cent, ids)
    q = (base[rng.integers(0, 300, 200)] + rng.integers(-1, 2, size=(200, 64))).astype(np.float32)
    for k in (5, 40, 64):
        _check(ctx, parent, cent, q, k, "l2", ids)
    parent.close()
    same = np.tile(rng.standard_normal((1, 128)).astype(np.float32), (4000, 1))  # 4000 copies of one vector
    parent = _parent(ctx, same)
    q = rng.standard_normal((70, 128)).astype(np.float32)
    _check(ctx, parent, same, q, 10, "l2")
    _check(ctx, parent, same, q, 10, "ip")
    parent.close()


def test_large_norm_spread(ctx):
    """rows of very different norms (the bound scales with |x|^2 + |y|^2): tiny rows next to huge ones"""
    rng = np.random.default_rng(13)
    cent = rng.standard_normal((8192, 96)).astype(np.float32) * np.exp(rng.uniform(-6, 6, size=(8192, 1))).astype(np.float32)
    parent = _parent(ctx, cent)
    q = (cent[rng.integers(0, 8192, 256)] * (1 + 0.05 * rng.standard_normal((256, 96)))).astype(np.float32)
    for metric in ("l2", "ip"):
        _check(ctx, parent, cent, q, 16, metric)
    parent.close()


@pytest.mark.parametrize("d", [128, 100, 30])
def test_row_major_copy_follows_the_store(ctx, d):
    """the exact finish reads the candidates from a row-major copy of the list (d a multiple of 4; d = 30 keeps gathering from the
    tile-major arena), made on first use and dropped whenever the store's table changes: centroids added and removed between calls
    -- what split / delete of the maintenance policy do to the parent (partition_manager.cpp:236-258, 265-318) -- must show up"""
    rng = np.random.default_rng(11 + d)
    n = 3000
    cent = rng.standard_normal((n, d)).astype(np.float32)
    ids = np.arange(n, dtype=np.int64)
    parent = _parent(ctx, cent, ids)
    q = (cent[rng.integers(0, n, 200)] + 0.2 * rng.standard_normal((200, d))).astype(np.float32)
    _check(ctx, parent, cent, q, 8, "l2", ids)
    _check(ctx, parent, cent, q, 8, "l2", ids)                 # (second call: the copy exists)
    new = (q[:150] + 1e-3).astype(np.float32)                  # 150 new centroids right next to the queries
    new_ids = np.arange(150, dtype=np.int64) + 100000
    parent.add_entries(0, new_ids, new)
    cent2, ids2 = np.concatenate([cent, new]), np.concatenate([ids, new_ids])
    gp, _ = ctx.coarse(parent, q, 8, "l2")
    assert (gp[:150, 0] == new_ids).all()
    _check(ctx, parent, cent2, q, 8, "l2", ids2)
    parent.remove_ids(new_ids[:75])
    # (remove = swap with the last row, index_partition.cpp:79-102: compare as sets of (id, distance) per query through the oracle on
    #  the surviving rows in THEIR new order is not needed -- ids and distances are order-free under the total order)
    keep = ~np.isin(ids2, new_ids[:75])
    op, od = O.coarse(q, cent2[keep], ids2[keep], 8, "l2")
    gp, gd = ctx.coarse(parent, q, 8, "l2")
    np.testing.assert_array_equal(gp, op)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    parent.close()


def test_huge_batches_leave_the_prefiltered_form(ctx):
    """the prefiltered form keeps ~14 KB of state per query: qk_coarse hands it a batch beyond 65536 queries in pieces of 65536 (round
    6: the maintenance policy ranks 10^5 ... 10^6 rows per call; the key-matrix path took 26.5 ms where four pieces take 4.8); k = 1
    over many rows beyond 16384 queries -- the assignment batches of PartitionManager::add -- goes to the fused argmin; same answers,
    host and device buffers"""
    rng = np.random.default_rng(21)
    cent = rng.standard_normal((2048, 16)).astype(np.float32)
    parent = _parent(ctx, cent)
    q = (cent[rng.integers(0, 2048, 70000)] + 0.3 * rng.standard_normal((70000, 16))).astype(np.float32)
    _check(ctx, parent, cent, q, 4, "l2")
    import torch
    gp, gd = ctx.coarse(parent, torch.from_numpy(q).cuda(), 2, "ip")  # (device buffers: the pieces write at their offsets)
    torch.cuda.synchronize()
    op, od = O.coarse(q, cent, None, 2, "ip")
    np.testing.assert_array_equal(gp.cpu().numpy(), op)
    np.testing.assert_array_equal(gd.cpu().numpy().view(np.uint32), od.view(np.uint32))
    parent.close()
    cent = rng.standard_normal((33000, 16)).astype(np.float32)
    parent = _parent(ctx, cent)
    q = (cent[rng.integers(0, 33000, 20000)] + 0.3 * rng.standard_normal((20000, 16))).astype(np.float32)
    _check(ctx, parent, cent, q, 1, "l2")
    _check(ctx, parent, cent, q[:5000], 1, "l2")
    parent.close()


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("n", [8192, 4096])   # the prefiltered form and (at k = 40) the one-launch form below it
def test_bound_adversarial_at_the_kth_key(ctx, metric, n):
    """centroids whose keys lie within one bf16 rounding step of a query's k-th best -- clusters of near-copies
    scale * (base + 2^-12 noise) at scales 2^-20 .. 2^+20, with exact duplicates straddling the cut -- so that the approximate
    products say nothing about their order and only the one-sided bounds (group minima, candidate filter) keep every true
    member a candidate.  The coarse step must return the oracle's ids and distance bits for k inside, at and beyond the size
    of a cluster of near-copies."""
    rng = np.random.default_rng(n + (metric == "ip"))
    d = 96
    scales = [2.0 ** -20, 2.0 ** -10, 1.0, 2.0 ** 10, 2.0 ** 20]
    per = n // len(scales)
    bases = rng.standard_normal((len(scales), d)).astype(np.float32)
    rows = []
    for j, sc in enumerate(scales):
        m = per if j < len(scales) - 1 else n - per * (len(scales) - 1)
        c = (sc * (bases[j] + 2.0 ** -12 * rng.standard_normal((m, d)))).astype(np.float32)
        c[:30] = c[30:60]  # exact duplicates
        rows.append(c)
    cent = np.ascontiguousarray(np.concatenate(rows), np.float32)
    ids = rng.permutation(n).astype(np.int64) + 5
    parent = _parent(ctx, cent, ids)
    which = rng.integers(0, len(scales), size=320)
    q = np.stack([scales[w] * (bases[w] + 2.0 ** -9 * rng.standard_normal(d)) for w in which]).astype(np.float32)
    for k in (2, 8, 40, 64):
        _check(ctx, parent, cent, q, k, metric, ids)
    parent.close()
