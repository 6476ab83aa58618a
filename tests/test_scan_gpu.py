"""GPU parity of the scan / coarse / search entry points against the CPU oracle, through the C ABI.
Bar: bit-exact int64 ids AND bit-exact float32 distances vs the oracle's batched (expanded-L2) path, which
uses the same canonical arithmetic; <= 1e-4 on distances vs the oracle's serial (direct-L2) path."""
import numpy as np
import pytest

import oracle as O
from helpers import brute_force, make_ivf, make_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    yield c
    c.close()


def build_stores(ctx, ivf):
    from quake_amd.capi import Store
    s = Store(ctx, ivf["d"])
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    parent = Store(ctx, ivf["d"])
    nlist = ivf["nlist"]
    parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), ivf["centroids"])
    return parent, s


def test_store_roundtrip(ctx):
    ivf = make_ivf(1000, 20, 7, seed=1, empty=(3,))
    parent, s = build_stores(ctx, ivf)
    assert s.ntotal() == 1000 and s.nlist() == 7
    for p in range(7):
        v, i = s.get_list(p)
        np.testing.assert_array_equal(i, ivf["part_ids"][p])
        np.testing.assert_array_equal(v, ivf["part_vecs"][p])
    assert s.list_size(3) == 0
    v = s.get_vector(int(ivf["ids"][17]))
    np.testing.assert_array_equal(v, ivf["vecs"][17])


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("d", [32, 128, 100])
def test_scan_matches_oracle_bit_exact(ctx, metric, d):
    ivf = make_ivf(20000, d, 32, seed=2, metric=metric, empty=(5,))
    q = make_queries(200, d, seed=3, like=ivf["x"], metric=metric)
    parent, s = build_stores(ctx, ivf)
    rng = np.random.default_rng(4)
    for P, k in [(1, 1), (4, 10), (32, 100)]:
        pids = np.stack([rng.permutation(32)[:P] for _ in range(q.shape[0])]).astype(np.int64)
        pids[::7, -1] = -1  # skipped entries (query_coordinator.cpp:540)
        gi, gd = ctx.scan(s, q, pids, k, metric)
        oi, od = O.batched_serial_scan(q, ivf["vecs"], ivf["ids"], ivf["offsets"], pids, k, metric)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
        si, sd = O.serial_scan(q, ivf["vecs"], ivf["ids"], ivf["offsets"], pids, k, metric)
        np.testing.assert_allclose(gd, sd, atol=1e-4)


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_coarse_and_search_match_oracle(ctx, metric):
    ivf = make_ivf(30000, 64, 64, seed=5, metric=metric)
    q = make_queries(300, 64, seed=6, like=ivf["x"], metric=metric)
    parent, s = build_stores(ctx, ivf)
    for nprobe, k in [(1, 10), (8, 10), (64, 50), (100, 5)]:
        cp, cd = ctx.coarse(parent, q, nprobe, metric)
        op, od = O.coarse(q, ivf["centroids"], None, nprobe, metric)
        np.testing.assert_array_equal(cp, op)
        np.testing.assert_array_equal(cd.view(np.uint32), od.view(np.uint32))
        gi, gd = ctx.search(parent, s, q, nprobe, k, metric)
        oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, metric, batched_scan=True)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_integer_ties_bit_exact(ctx):
    # SIFT-like integer data: every fp32 partial sum exact, many exact distance ties -> exercises the (key,id) rule
    ivf = make_ivf(20000, 128, 16, seed=7, integer=True)
    q = make_queries(64, 128, seed=8, like=ivf["x"], integer=True)
    parent, s = build_stores(ctx, ivf)
    gi, gd = ctx.search(parent, s, q, 16, 50, "l2")
    bi, bd, gaps = brute_force(ivf["vecs"], ivf["ids"], q, 50, "l2")
    assert (gaps == 0).any()
    np.testing.assert_array_equal(gi, bi)
    np.testing.assert_array_equal(gd, bd)
    # on exact data the direct and the expanded form agree bit for bit, so the serial oracle matches too
    si, sd = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 16, 50, "l2", batched_scan=False)
    np.testing.assert_array_equal(gi, si)
    np.testing.assert_array_equal(gd, sd)


def test_flat_index_padding_and_errors(ctx):
    from quake_amd._lib import QuakeHipError
    ivf = make_ivf(7, 8, 2, seed=9)
    q = make_queries(5, 8, seed=10)
    parent, s = build_stores(ctx, ivf)
    for metric in ("l2", "ip"):
        gi, gd = ctx.search(None, s, q, 1, 10, metric)  # flat: every list scanned (query_coordinator.cpp:624-626)
        oi, od = O.search(q, None, ivf["vecs"], ivf["ids"], ivf["offsets"], 1, 10, metric, batched_scan=True)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
        assert (gi[:, 7:] == -1).all() and np.isinf(gd[:, 7:]).all()
    # zero partitions to scan (query_coordinator.cpp:459-497)
    zi, zd = ctx.scan(s, q, np.zeros((5, 0), np.int64), 4, "l2")
    assert (zi == -1).all() and np.isposinf(zd).all()
    with pytest.raises(QuakeHipError):  # "List does not exist"
        ctx.scan(s, q, np.full((5, 1), 99, np.int64), 4, "l2")


def test_large_flat_list_vs_torch_bruteforce(ctx):
    """The reference's own large-list check (test/cpp/list_scanning.cpp:432-562): randn 100 x 10000 x 128, k=10,
    ids equal to torch topk, |ddist| <= 0.01 -- here run on the GPU path."""
    import torch
    g = torch.Generator().manual_seed(1234)
    q = torch.randn(100, 128, generator=g)
    x = torch.randn(10000, 128, generator=g)
    from quake_amd.capi import Store
    s = Store(ctx, 128)
    s.build_csr(np.array([0, 10000], np.int64), np.arange(10000, dtype=np.int64), x.numpy())
    for metric in ("l2", "ip"):
        gi, gd = ctx.search(None, s, q.numpy(), 1, 10, metric)
        if metric == "l2":
            dist = torch.cdist(q.double(), x.double())
            tv, ti = torch.topk(dist, 10, dim=1, largest=False)
        else:
            dist = q.double() @ x.double().T
            tv, ti = torch.topk(dist, 10, dim=1, largest=True)
        np.testing.assert_array_equal(gi, ti.numpy())
        np.testing.assert_allclose(gd, tv.numpy(), atol=1e-4)


def test_large_k_and_limits(ctx):
    """k up to QK_MAX_K goes through the multi-chunk pools, up to 8192 (the reference's buffer capacity) through key emission + selection; beyond that the call
    fails loudly (no silent truncation)."""
    from quake_amd._lib import QK_MAX_K, QuakeHipError
    ivf = make_ivf(6000, 48, 6, seed=31)
    q = make_queries(40, 48, seed=32, like=ivf["x"])
    parent, s = build_stores(ctx, ivf)
    for k in (200, QK_MAX_K):
        gi, gd = ctx.search(parent, s, q, 6, k, "l2")
        oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 6, k, "l2", batched_scan=True)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    # beyond QK_MAX_K the keys are emitted and selected afterwards (k_scan MODE 4 + k_select_pairs_large): same answers
    for k, metric_ in ((QK_MAX_K + 1, "l2"), (1000, "l2"), (777, "ip"), (4096, "l2"), (8192, "ip")):
        iv = make_ivf(16000 if k > 4096 else 6000, 48, 6, seed=33, metric=metric_)
        vecs = iv["vecs"].copy()
        a0, a1 = int(iv["offsets"][1]), int(iv["offsets"][2])
        vecs[a0 + 50:a0 + 100] = vecs[a0:a0 + 50]  # duplicates inside a list: ties ordered by id
        from quake_amd.capi import Store
        sw = Store(ctx, 48)
        sw.build_csr(iv["offsets"], iv["ids"], vecs)
        qq = make_queries(9, 48, seed=34, like=iv["x"], metric=metric_)
        rng = np.random.default_rng(35)
        pids = np.stack([rng.permutation(6)[:4] for _ in range(9)]).astype(np.int64)
        pids[2, 1] = -1
        gi, gd = ctx.scan(sw, qq, pids, k, metric_)
        oi, od = O.batched_serial_scan(qq, vecs, iv["ids"], iv["offsets"], pids, k, metric_)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    gi, gd = ctx.search(parent, s, q, 6, 2000, "l2")  # through the whole search, every list probed
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 6, 2000, "l2", batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    with pytest.raises(QuakeHipError):
        ctx.search(parent, s, q, 6, 8193, "l2")


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_small_batches_one_launch_search(ctx, metric):
    """Q <= 32 with a flat parent goes through the one-launch search (qk_small.hip): coarse + selection + scan + merge in one
    kernel -- with the coarse keys computed once per workgroup (one or two queries) or split across the workgroups of a query
    behind an arrival counter (larger batches).  Same bits as the oracle for every (Q, nprobe, k), with an empty list, a list shorter than k, duplicated
    centroids (a tie that straddles the nprobe cut: ordered by partition id) and duplicated rows (ties ordered by id)."""
    ivf = make_ivf(30000, 96, 40, seed=41, metric=metric, empty=(7,))
    cent = ivf["centroids"].copy()
    cent[11] = cent[3]
    cent[12] = cent[3]
    cent[30] = cent[3]  # four identical centroids: whichever queries are near them see a 4-way tie in the coarse keys
    vecs = ivf["vecs"].copy()
    a0 = int(ivf["offsets"][3])
    vecs[a0 + 20:a0 + 40] = vecs[a0:a0 + 20]
    from quake_amd.capi import Store
    s = Store(ctx, 96)
    s.build_csr(ivf["offsets"], ivf["ids"], vecs)
    parent = Store(ctx, 96)
    parent.build_csr(np.array([0, 40], np.int64), np.arange(40, dtype=np.int64), cent)
    q = make_queries(32, 96, seed=42, like=ivf["x"], metric=metric)
    q[5] = vecs[a0 + 3]  # sits on a duplicated row of list 3 (and next to the duplicated centroids)
    for Q in (1, 3, 8, 16, 32):
        for nprobe, k in ((1, 1), (2, 10), (3, 10), (10, 10), (40, 32), (64, 5)):
            for rep in range(2):  # the second call finds the arrival counters and tickets the first one left behind
                gi, gd = ctx.search(parent, s, q[:Q], nprobe, k, metric)
                # (the envelope: at most 32k rows to scan per query and 320k per call -- beyond that the batch pipeline is faster)
                rows_q = min(nprobe, 40) * (30000 // 39)
                small = rows_q <= 32768 and rows_q * Q <= 327680
                assert (ctx.last_scan_kernel() == "k_search_small") == small, (Q, nprobe, ctx.last_scan_kernel())
                oi, od = O.search(q[:Q], cent, vecs, ivf["ids"], ivf["offsets"], nprobe, k, metric, batched_scan=True)
                np.testing.assert_array_equal(gi, oi, err_msg=f"Q={Q} nprobe={nprobe} k={k}")
                np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    # deferred timing on this path: one event pair per call, all of it "scan"
    ctx.set_timing(2)
    for _ in range(3):
        ctx.search(parent, s, q[:8], 10, 10, metric)
    t = ctx.read_timing()
    ctx.set_timing(0)
    assert t["calls"] == 3 and t["scan_ms"] > 0.0 and t["coarse_ms"] == 0.0
    # device buffers, and a tiny index where every list is shorter than k
    import torch
    qd = torch.from_numpy(q[:4]).cuda()
    gi, gd = ctx.search(parent, s, qd, 10, 10, metric)
    oi, od = O.search(q[:4], cent, vecs, ivf["ids"], ivf["offsets"], 10, 10, metric, batched_scan=True)
    np.testing.assert_array_equal(gi.cpu().numpy(), oi)
    np.testing.assert_array_equal(gd.cpu().numpy().view(np.uint32), od.view(np.uint32))
    tiny = make_ivf(30, 96, 8, seed=43, metric=metric)
    pt, st = build_stores(ctx, tiny)
    gi, gd = ctx.search(pt, st, q[:2], 3, 20, metric)
    oi, od = O.search(q[:2], tiny["centroids"], tiny["vecs"], tiny["ids"], tiny["offsets"], 3, 20, metric, batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    assert (gi == -1).any()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_mid_sized_batches_one_launch_coarse(ctx, metric):
    """33..256 queries against <= 4096 centroids, 2 <= nprobe <= 64: the coarse step is ONE launch (k_coarse_small,
    qk_small.hip) -- same partitions, same order, same distance bits as the oracle's coarse step (= the reference's parent
    search, query_coordinator.cpp:628-644), duplicated centroids on the nprobe cut included; shapes outside the envelope
    (nprobe 100, 300 queries) take the two-kernel form and must agree as well"""
    rng = np.random.default_rng(17)
    for nlist, d in [(5, 24), (64, 128), (1000, 128), (4096, 40), (300, 200)]:
        ivf = make_ivf(max(4 * nlist, 2000), d, nlist, seed=40 + nlist, metric=metric)
        cent = ivf["centroids"].copy()
        if nlist >= 64:  # exact duplicates: ties between partitions, resolved by partition id
            cent[nlist // 2:nlist // 2 + 20] = cent[:20]
            ivf = dict(ivf, centroids=cent)
        parent, s = build_stores(ctx, ivf)
        for Q in (33, 64, 200, 256, 300):
            q = make_queries(Q, d, seed=int(rng.integers(1 << 30)), like=ivf["x"], metric=metric)
            if nlist >= 64:
                q[: Q // 2] = cent[rng.integers(0, 20, Q // 2)] + (0.01 * rng.standard_normal((Q // 2, d))).astype(np.float32)
            for nprobe in (2, 10, 64, 100):
                cp, cd = ctx.coarse(parent, q, nprobe, metric)
                op, od = O.coarse(q, cent, None, nprobe, metric)
                np.testing.assert_array_equal(cp, op)
                np.testing.assert_array_equal(cd.view(np.uint32), od.view(np.uint32))
            gi, gd = ctx.search(parent, s, q, 10, 10, metric)
            oi, od = O.search(q, cent, ivf["vecs"], ivf["ids"], ivf["offsets"], 10, 10, metric, batched_scan=True)
            np.testing.assert_array_equal(gi, oi)
            np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_long_rows_sliced_selection(ctx, metric):
    """>= 8192 rows in the scanned list (coarse step over many centroids, flat index) with k <= 64: the selection runs one wave
    per 4096-key slice + a merge of the slices' candidates (k_select_rows slices, k_merge_slices) -- same ids, order and
    distance bits as the oracle; duplicated rows across slices included; k = 100 takes the one-wave-per-row form"""
    rng = np.random.default_rng(23)
    for nlist, d in [(8192, 16), (20000, 24), (70000, 8)]:
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        cent[nlist - 50:] = cent[:50]  # exact duplicates in the first and the last slice: ties cut by id
        if metric == "ip":
            cent /= np.linalg.norm(cent, axis=1, keepdims=True)
        from quake_amd.capi import Store
        parent = Store(ctx, d)
        parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), cent)
        for Q in (40, 300):
            q = (cent[rng.integers(0, nlist, Q)] + 0.05 * rng.standard_normal((Q, d))).astype(np.float32)
            q[: Q // 2] = cent[rng.integers(0, 50, Q // 2)]
            for nprobe in (2, 10, 64, 100):
                cp, cd = ctx.coarse(parent, q, nprobe, metric)
                op, od = O.coarse(q, cent, None, nprobe, metric)
                np.testing.assert_array_equal(cp, op)
                np.testing.assert_array_equal(cd.view(np.uint32), od.view(np.uint32))
    # flat index (one list of 30000 rows, ids not in row order): same path through qk_search
    ivf = make_ivf(30000, 32, 1, seed=77, metric=metric)
    parent, s = build_stores(ctx, ivf)
    q = make_queries(64, 32, seed=78, like=ivf["x"], metric=metric)
    gi, gd = ctx.search(None, s, q, 1, 10, metric)
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 1, 10, metric, batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_wide_k_with_timing_on_a_fresh_context():
    """k > QK_MAX_K with a qk_timing requested as the FIRST call of a context (what QuakeIndex.search always does): the wide-k
    pipeline records its own phase events, so the timings are real and the call does not fail on never-recorded events."""
    from quake_amd.capi import Context
    c = Context(0)
    c.set_timing(1)
    ivf = make_ivf(6000, 48, 6, seed=36)
    q = make_queries(12, 48, seed=37, like=ivf["x"])
    parent, s = build_stores(c, ivf)
    gi, gd, t = c.search(parent, s, q, 4, 1000, "l2", timing=True)
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 4, 1000, "l2", batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    for f in ("coarse_ms", "group_ms", "scan_ms", "merge_ms", "total_ms"):
        assert 0.0 <= t[f] < 1000.0, (f, t[f])
    assert t["total_ms"] >= t["scan_ms"] > 0.0
    # and the pair count of a plain search is the number of (query, partition) pairs that reached a non-empty list
    gi, gd, t = c.search(parent, s, q, 4, 10, "l2", timing=True)
    assert t["partitions_scanned"] == 12 * 4
    c.close()


@pytest.mark.parametrize("d,metric", [(768, "ip"), (768, "l2"), (1024, "l2"), (512, "ip")])
def test_wide_rows_shared_query_tile(ctx, d, metric):
    """d >= 512: several waves of a workgroup share one LDS query tile and split each segment (k_scan nw = 2/4), pools
    wider than one wave use the bisection select; exact duplicates force the tie fallback of both compactions."""
    ivf = make_ivf(12000, d, 12, seed=41, metric=metric, empty=(2,))
    # duplicate vectors under different ids: equal keys that must come back ordered by id
    vecs = ivf["vecs"].copy()
    off = ivf["offsets"]
    for p in (0, 4, 7):
        a, b = int(off[p]), int(off[p + 1])
        if b - a > 40:
            vecs[a + 20:a + 40] = vecs[a:a + 20]
    from quake_amd.capi import Store
    s = Store(ctx, d)
    s.build_csr(off, ivf["ids"], vecs)
    q = make_queries(70, d, seed=42, like=ivf["x"], metric=metric)
    q[:20] = vecs[int(off[0]):int(off[0]) + 20]  # queries that hit the duplicated rows exactly
    rng = np.random.default_rng(43)
    for P, k in [(1, 10), (5, 100), (12, 100), (12, 37), (3, 300)]:
        pids = np.stack([rng.permutation(12)[:P] for _ in range(q.shape[0])]).astype(np.int64)
        pids[:20, 0] = 0
        gi, gd = ctx.scan(s, q, pids, k, metric)
        oi, od = O.batched_serial_scan(q, vecs, ivf["ids"], off, pids, k, metric)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_large_nprobe_selection(ctx):
    """nprobe / candidate counts beyond QK_MAX_K (up to QK_MAX_NPROBE) go through k_select_rows_large: bisection on the key
    bits, id cut among ties, bitonic sort.  Exact (key, id) order vs the oracle, duplicated centroids included."""
    from quake_amd._lib import QK_MAX_K, QK_MAX_NPROBE, QuakeHipError
    from quake_amd.capi import Store
    rng = np.random.default_rng(51)
    nl, d = 3000, 32
    c = rng.standard_normal((nl, d)).astype(np.float32)
    c[100:160] = c[40:100]   # exact duplicates: equal keys, order decided by id
    c[2000:2400] = c[1000]   # 400-way tie
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nl], np.int64), np.arange(nl, dtype=np.int64), c)
    for Q in (3, 70):
        x = rng.standard_normal((Q, d)).astype(np.float32)
        x[0] = c[1000]
        for metric in ("l2", "ip"):
            for kk in (QK_MAX_K + 1, 1000, 2999, 3000):
                gp, gd = ctx.coarse(parent, x, kk, metric)
                op, od = O.coarse(x, c, None, kk, metric)
                np.testing.assert_array_equal(gp, op)
                np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    # more candidates than rows: padded with -1 like every other entry point
    gp, gd = ctx.coarse(parent, x[:2], 3500, "l2")
    assert gp.shape[1] == 3000  # kk = min(nprobe, ntotal)
    # the whole search with nprobe > QK_MAX_K, and the flat index with k > QK_MAX_K
    ivf = make_ivf(30000, 16, 600, seed=52)
    pq, s = build_stores(ctx, ivf)
    q = make_queries(40, 16, seed=53, like=ivf["x"])
    gi, gd = ctx.search(pq, s, q, 500, 10, "l2")
    oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 500, 10, "l2", batched_scan=True)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    flat = Store(ctx, 16)
    flat.build_csr(np.array([0, 5000], np.int64), ivf["ids"][:5000], ivf["vecs"][:5000])
    gi, gd = ctx.search(None, flat, q, 1, 1000, "l2")
    oi, od = O.batched_serial_scan(q, ivf["vecs"][:5000], ivf["ids"][:5000], np.array([0, 5000], np.int64),
                                   np.zeros((40, 1), np.int64), 1000, "l2")
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    with pytest.raises(QuakeHipError):
        ctx.coarse(parent, x, QK_MAX_NPROBE + 1, "l2") if nl > QK_MAX_NPROBE else ctx.search(None, flat, q, 1, QK_MAX_NPROBE + 1, "l2")


def test_deferred_timing_modes(ctx):
    """qk_ctx_set_timing 2 (events around every phase) and 3 (one pair around the scan kernel): summed by
    qk_ctx_read_timing without a synchronisation inside the calls; results are unaffected."""
    import torch
    ivf = make_ivf(20000, 32, 32, seed=21)
    q = torch.from_numpy(make_queries(64, 32, seed=22, like=ivf["x"])).cuda()
    parent, s = build_stores(ctx, ivf)
    ref_i, ref_d = ctx.search(parent, s, q, 4, 10, "l2")
    for mode in (2, 3):
        ctx.set_timing(mode)
        for _ in range(3):
            gi, gd = ctx.search(parent, s, q, 4, 10, "l2")
        t = ctx.read_timing()
        ctx.set_timing(0)
        assert t["calls"] == 3 and t["scan_ms"] > 0.0
        if mode == 3:
            assert t["group_ms"] == 0.0 and t["merge_ms"] == 0.0 and t["coarse_ms"] == 0.0
        else:
            assert t["group_ms"] > 0.0 and t["coarse_ms"] > 0.0
        assert torch.equal(gi, ref_i) and torch.equal(gd, ref_d)
    assert ctx.read_timing()["calls"] == 0


def test_two_contexts_search_one_store_concurrently():
    """two contexts (two HIP streams) searching the same stores back to back without synchronising in between -- the
    batches_in_flight measurement of bench.py -- give the answers of a single context"""
    import torch
    from quake_amd.capi import Context, Store
    c1, c2 = Context(0), Context(0)
    ivf = make_ivf(200000, 64, 128, seed=91)
    s = Store(c1, 64)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    parent = Store(c1, 64)
    parent.build_csr(np.array([0, 128], np.int64), np.arange(128, dtype=np.int64), ivf["centroids"])
    qs = [torch.from_numpy(make_queries(512, 64, seed=92 + b, like=ivf["x"])).cuda() for b in range(4)]
    ref = []
    for nprobe, k in ((1, 10), (8, 10)):
        for q in qs:
            ri, rd = c1.search(parent, s, q, nprobe, k, "l2")
            c1.synchronize()
            ref.append((ri.clone(), rd.clone()))
    outs = [(torch.empty((512, 10), dtype=torch.int64, device="cuda"), torch.empty((512, 10), dtype=torch.float32, device="cuda"))
            for _ in range(8)]
    for rep in range(5):
        j = 0
        for nprobe, k in ((1, 10), (8, 10)):
            for b, q in enumerate(qs):
                (c1, c2)[j % 2].search(parent, s, q, nprobe, k, "l2", out=outs[j])
                j += 1
        c1.synchronize()
        c2.synchronize()
        for j in range(8):
            assert torch.equal(outs[j][0], ref[j][0]) and torch.equal(outs[j][1], ref[j][1]), (rep, j)
    s.close()
    parent.close()
    c2.close()
    c1.close()


@pytest.mark.parametrize("mem", ["host", "device"])
def test_search_tracked_hands_out_the_probed_lists(ctx, mem):
    """qk_search_tracked = qk_search + the list numbers every query scanned (what QuakeIndex::search passes to
    MaintenancePolicy::record_query_hits, maintenance_policies.cpp:179-182): ids and distance bits of qk_search and of the oracle,
    probed lists those of qk_coarse -- for one query and a batch, nprobe 1 (the packed nearest-centroid path of qk_search), nprobe
    beyond the number of lists, host and device buffers."""
    import torch
    ivf = make_ivf(30000, 64, 48, seed=31, empty=(7,))
    parent, s = build_stores(ctx, ivf)
    for Q, nprobe, k in [(1, 5, 10), (9, 1, 3), (700, 1, 10), (300, 6, 10), (40, 100, 20)]:
        q = make_queries(Q, 64, seed=32 + Q, like=ivf["x"])
        qq = torch.from_numpy(q).cuda() if mem == "device" else q
        gi, gd, gp, tm = ctx.search_tracked(parent, s, qq, nprobe, k, "l2", timing=True)
        si, sd = ctx.search(parent, s, qq, nprobe, k, "l2")
        cp, _ = ctx.coarse(parent, qq, nprobe, "l2")
        if mem == "device":
            torch.cuda.synchronize()
            gi, gd, gp, si, sd, cp = (t.cpu().numpy() for t in (gi, gd, gp, si, sd, cp))
        assert gp.shape == (Q, min(nprobe, 48))
        np.testing.assert_array_equal(gp, cp)
        np.testing.assert_array_equal(gi, si)
        np.testing.assert_array_equal(gd.view(np.uint32), sd.view(np.uint32))
        oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, "l2", batched_scan=True)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
        assert tm["partitions_scanned"] > 0
    s.close()
    parent.close()


def test_get_vectors_in_one_call(ctx):
    """qk_store_get_vectors == qk_store_get_vector id by id (PartitionManager::get, partition_manager.cpp:264-283): a few ids per list
    (the list is scanned), hundreds from one list (a position table is built for it), ids the store does not hold, an id asked twice."""
    ivf = make_ivf(5000, 20, 9, seed=41, empty=(2,))
    parent, s = build_stores(ctx, ivf)
    rng = np.random.default_rng(42)
    few = rng.choice(ivf["ids"], 12, replace=False)
    many = ivf["part_ids"][4][:300]
    ask = np.concatenate([few, many, [10**9, few[0]], np.arange(9)])  # (0 .. 8: whatever of them the corpus holds)
    vecs, found = s.get_vectors(ask)
    for i, vid in enumerate(ask):
        one = s.get_vector(int(vid))
        assert bool(found[i]) == (one is not None), (i, vid)
        if one is not None:
            np.testing.assert_array_equal(vecs[i], one)
    assert not found[len(few) + len(many)] and found[:len(few) + len(many)].all()
    c, f = parent.get_vectors(np.arange(9))  # the parent: one list of 9 centroids
    assert f.all()
    np.testing.assert_array_equal(c, ivf["centroids"])
    assert s.get_vectors(np.zeros(0, np.int64))[0].shape == (0, 20)
    s.close()
    parent.close()
