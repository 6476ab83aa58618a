"""Device group (qk_group_*, include/quake_hip.h): IndexBuildParams::num_workers as GPUs -- ONE process, G members, list p in
member p % G.  On a one-GPU box the members share device 0 (exactly the rehearsal the 4-rank gloo tests do with processes); every
step -- replicated centroids, coarse split by queries with peer-written list numbers, per-member scan, peer-written packed top-k,
lead merge -- is the code an 8-GPU node runs, with local instead of peer addresses.

Bar: ids and float32 distance bits equal the one-store search (and the oracle's batched path) on the same lists.
Reference: QueryCoordinator::worker_scan == serial scan (test/cpp/query_coordinator.cpp:201-254: ids equal, distances <= 1e-4)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from helpers import make_ivf, make_queries  # noqa: E402


def _bits(a):
    a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _np(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


@pytest.fixture(scope="module")
def capi():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from quake_amd import capi
    return capi


def _devices(G):
    n = torch.cuda.device_count()
    return [j % n for j in range(G)]


def _build(capi, ivf, G, metric="l2", mem="host"):
    d = ivf["d"]
    ctx = capi.Context(0)
    single = capi.Store(ctx, d)
    parent = capi.Store(ctx, d)
    nl = ivf["nlist"]
    parent.build_csr(np.array([0, nl], np.int64), np.arange(nl, dtype=np.int64), ivf["centroids"])
    grp = capi.Group(_devices(G), d)
    if mem == "host":
        single.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
        grp.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    else:
        iv, vv = torch.from_numpy(ivf["ids"]).cuda(), torch.from_numpy(ivf["vecs"]).cuda()
        single.build_csr(ivf["offsets"], iv, vv)
        grp.build_csr(ivf["offsets"], iv, vv)
    return ctx, parent, single, grp


@pytest.mark.parametrize("G", [1, 2, 4, 7])
@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_group_search_equals_single_store_and_oracle(capi, G, metric):
    ivf = make_ivf(60000, 64, 96, seed=3, metric=metric, empty=(5, 17))
    ctx, parent, single, grp = _build(capi, ivf, G, metric)
    assert grp.size() == G and grp.ntotal() == single.ntotal() and grp.nlist() == single.nlist()
    assert (grp.list_ids() == single.list_ids()).all()
    for j in range(G):
        held = grp.member_list_ids(j)
        assert (held % G == j).all()
    for Q, nprobe, k in [(1, 8, 10), (33, 1, 10), (700, 12, 10), (1500, 4, 100), (64 * G, 3, 7)]:
        q = make_queries(Q, 64, seed=100 + Q, like=ivf["x"], metric=metric)
        si, sd = ctx.search(parent, single, q, nprobe, k, metric)
        gi, gd, tm = grp.search(parent, q, nprobe, k, metric, timing=True)
        assert (gi == si).all(), (G, Q, nprobe, k)
        assert (_bits(gd) == _bits(sd)).all()
        oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, metric, batched_scan=True)
        assert (gi == oi).all()
        assert (_bits(gd) == _bits(od)).all()
        assert tm["total_ms"] > 0
        # device buffers in, device buffers out
        qd = torch.from_numpy(q).cuda()
        di, dd = grp.search(parent, qd, nprobe, k, metric)
        grp.synchronize()
        assert (_np(di) == si).all() and (_bits(dd) == _bits(sd)).all()


@pytest.mark.parametrize("G,k", [(2, 961), (3, 1500), (4, 4000)])
def test_group_large_k_equals_single_store(capi, G, k):
    """k beyond the LDS pools of the cross-member merge (k > 960): the sorted-run merge (k_merge_ranks_large, qk_merge.hip) -- the
    reference has no k limit with workers (query_coordinator.cpp:243-469), and one store serves k up to 8192.  Also k larger than
    everything the probed lists hold (padding behind the merged run)."""
    ivf = make_ivf(30000, 32, 48, seed=9, empty=(3,))
    ctx, parent, single, grp = _build(capi, ivf, G)
    for Q, nprobe in [(5, 48), (40, 7), (3, 1)]:
        q = make_queries(Q, 32, seed=200 + Q, like=ivf["x"])
        si, sd = ctx.search(parent, single, q, nprobe, k, "l2")
        gi, gd = grp.search(parent, q, nprobe, k, "l2")
        assert (gi == si).all(), (G, k, Q, nprobe)
        assert (_bits(gd) == _bits(sd)).all()
        if nprobe == 1:
            assert (gi[:, -1] == -1).all()  # one list of ~625 rows: fewer than k entries, the rest is padding
    grp.close()
    single.close()
    parent.close()
    ctx.close()


def test_group_submit_threads_change_nothing(capi):
    """one enqueue thread per member (qk_group_set_submit_threads, the default) against the caller's thread doing every member: the
    same streams and events, so the same bits -- over a few hundred calls of changing shape (the pieces of consecutive calls
    interleave on the members' streams; an ordering slip between threads would show as a stale or torn answer)."""
    ivf = make_ivf(40000, 32, 64, seed=12)
    ctx, parent, single, grp = _build(capi, ivf, 5)
    rng = np.random.default_rng(3)
    shapes = [(1, 4, 5), (70, 1, 10), (640, 6, 10), (333, 3, 40), (2000, 2, 7)]
    want = {}
    for sh in shapes:
        q = make_queries(sh[0], 32, seed=300 + sh[0], like=ivf["x"])
        want[sh] = (q, *ctx.search(parent, single, q, sh[1], sh[2], "l2"))
    for rep in range(240):
        if rep % 40 == 0:
            grp.set_submit_threads(rep % 80 == 0)
        sh = shapes[int(rng.integers(0, len(shapes)))]
        q, si, sd = want[sh]
        if rep % 3 == 0:  # device buffers: nothing synchronises between calls
            gi, gd = grp.search(parent, torch.from_numpy(q).cuda(), sh[1], sh[2], "l2")
            if rep % 12 == 0:
                grp.synchronize()
                assert (_np(gi) == si).all() and (_bits(gd) == _bits(sd)).all(), (rep, sh)
        else:
            gi, gd = grp.search(parent, q, sh[1], sh[2], "l2")
            assert (gi == si).all() and (_bits(gd) == _bits(sd)).all(), (rep, sh)
    grp.synchronize()
    grp.close()
    single.close()
    parent.close()
    ctx.close()


def test_group_counters_sum_over_members(capi):
    ivf = make_ivf(40000, 32, 64, seed=5)
    ctx, parent, single, grp = _build(capi, ivf, 4)
    q = make_queries(512, 32, seed=9, like=ivf["x"])
    ctx.set_timing(1)
    _, _, ts = ctx.search(parent, single, q, 6, 10, "l2", timing=True)
    ctx.set_timing(0)
    _, _, tg = grp.search(parent, q, 6, 10, "l2", timing=True)
    assert tg["partitions_scanned"] == ts["partitions_scanned"]
    assert tg["scan_bytes"] == ts["scan_bytes"]  # every probed list is read once, by the member that holds it


def test_group_scan_seam_and_padding(capi):
    ivf = make_ivf(20000, 48, 40, seed=7, empty=(3,))
    ctx, parent, single, grp = _build(capi, ivf, 3)
    q = make_queries(200, 48, seed=8, like=ivf["x"])
    rng = np.random.default_rng(0)
    pids = rng.integers(0, 40, size=(200, 5)).astype(np.int64)
    pids[::7, 2] = -1
    pids[5] = -1  # nothing to scan for this query: padding (query_coordinator.cpp:459-497)
    si, sd = ctx.scan(single, q, pids, 10, "l2")
    gi, gd = grp.scan(q, pids, 10, "l2")
    assert (gi == si).all() and (_bits(gd) == _bits(sd)).all()
    assert (gi[5] == -1).all() and np.isinf(gd[5]).all()
    # the same set for every query (1-D list, query_coordinator.cpp:506-508) and zero partitions
    gi2, gd2 = grp.scan(q, np.array([1, 2, 4, 9], np.int64), 3, "l2")
    si2, sd2 = ctx.scan(single, q, np.array([1, 2, 4, 9], np.int64), 3, "l2")
    assert (gi2 == si2).all() and (_bits(gd2) == _bits(sd2)).all()
    gi3, gd3 = grp.scan(q, np.zeros((200, 0), np.int64), 4, "l2")
    assert (gi3 == -1).all() and np.isinf(gd3).all()
    # a list nobody holds: the reference's "List does not exist"
    with pytest.raises(RuntimeError, match="List does not exist"):
        grp.scan(q, np.full((200, 1), 4000, np.int64), 4, "l2")


def test_group_build_from_device_arrays(capi):
    ivf = make_ivf(30000, 32, 50, seed=11)
    ctx, parent, single, grp = _build(capi, ivf, 4, mem="device")
    q = make_queries(300, 32, seed=12, like=ivf["x"])
    si, sd = ctx.search(parent, single, q, 5, 10, "l2")
    gi, gd = grp.search(parent, q, 5, 10, "l2")
    assert (gi == si).all() and (_bits(gd) == _bits(sd)).all()
    for p in (0, 7, 49):
        gv, gid = grp.get_list(p)
        sv, sid = single.get_list(p)
        assert (gid == sid).all() and (gv == sv).all()


@pytest.mark.parametrize("mem", ["host", "device"])
def test_group_mutations_follow_the_single_store(capi, mem):
    """add_batch / remove_ids / add_list / add_entries / remove_list routed to the owners: the same lists, row for row, as one
    store fed the same calls (append order, swap-with-last removal: index_partition.cpp:52-59,79-102)."""
    d, nl, G = 32, 30, 4
    ivf = make_ivf(12000, d, nl, seed=21)
    ctx, parent, single, grp = _build(capi, ivf, G)
    rng = np.random.default_rng(2)

    def same():
        assert grp.ntotal() == single.ntotal() and grp.nlist() == single.nlist()
        assert (grp.list_ids() == single.list_ids()).all()
        for p in single.list_ids():
            gv, gi = grp.get_list(int(p))
            sv, si = single.get_list(int(p))
            assert (gi == si).all() and (gv == sv).all(), p

    next_id = 1_000_000
    for step in range(4):
        n = 3000
        x = (ivf["centroids"][rng.integers(0, nl, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
        ids = np.arange(next_id, next_id + n, dtype=np.int64)
        next_id += n
        assign = ctx.coarse(parent, x, 1, "l2", values=False)[0].reshape(-1)
        if mem == "device":
            xd, idd, ad = torch.from_numpy(x).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(np.ascontiguousarray(assign)).cuda()
            single.add_batch(idd, xd, ad)
            grp.add_batch(idd, xd, ad)
        else:
            single.add_batch(ids, x, assign)
            grp.add_batch(ids, x, assign)
        same()
        allids = np.concatenate([single.get_list(int(p))[1] for p in single.list_ids()])
        kill = rng.choice(allids, 2500, replace=False)
        assert grp.remove_ids(kill) == single.remove_ids(kill) == 2500
        same()
    # new lists (a split hands out the next partition numbers, partition_manager.cpp:492-493), one dropped
    for p in (nl, nl + 1, nl + 2):
        v = rng.standard_normal((257, d)).astype(np.float32)
        i = np.arange(next_id, next_id + 257, dtype=np.int64)
        next_id += 257
        for st in (single, grp):
            st.add_list(p)
            st.add_entries(p, i, v)
    single.remove_list(4)
    grp.remove_list(4)
    same()
    assert grp.owner(nl + 1) == (nl + 1) % G
    v = grp.get_vector(int(single.get_list(nl)[1][5]))
    assert v is not None and (v == single.get_list(nl)[0][5]).all()
    assert grp.get_vector(999_999_999) is None
    with pytest.raises(RuntimeError, match="List does not exist"):
        grp.add_batch(np.array([5], np.int64), np.zeros((1, d), np.float32), np.array([4], np.int64))
    q = make_queries(400, d, seed=30, like=ivf["x"])
    pids = np.broadcast_to(single.list_ids()[None, :8], (400, 8)).copy()
    si, sd = ctx.scan(single, q, pids, 10, "l2")
    gi, gd = grp.scan(q, pids, 10, "l2")
    assert (gi == si).all() and (_bits(gd) == _bits(sd)).all()


def test_group_refine_lists_across_members(capi):
    """kmeans_refine_partitions (clustering.cpp:99-182) over lists that live on different members == the same refinement of one
    store: centroids returned, list contents and row order."""
    d, nl, G = 24, 20, 4
    ivf = make_ivf(9000, d, nl, seed=31)
    ctx, parent, single, grp = _build(capi, ivf, G)
    lists = np.array([3, 8, 9, 14, 17], np.int64)  # owners 3, 0, 1, 2, 1
    cent = ivf["centroids"][lists] + 0.05
    for iters in (0, 2):
        cs = single.refine_lists(lists, cent, "l2", iters)
        cg = grp.refine_lists(lists, cent, "l2", iters)
        assert (_bits(cs) == _bits(cg)).all()
        for p in lists:
            gv, gi = grp.get_list(int(p))
            sv, si = single.get_list(int(p))
            assert (gi == si).all() and (gv == sv).all()
    # lists of ONE member take the member's own path
    one = np.array([1, 5, 13], np.int64)
    cs = single.refine_lists(one, ivf["centroids"][one], "l2", 1)
    cg = grp.refine_lists(one, ivf["centroids"][one], "l2", 1)
    assert (_bits(cs) == _bits(cg)).all()
    assert grp.ntotal() == single.ntotal()


def test_group_follows_a_changing_parent(capi):
    """the replicas of the centroids are refreshed when the parent store changes (a split adds centroids, a refine moves them)"""
    d, nl, G = 32, 24, 3
    ivf = make_ivf(15000, d, nl, seed=41)
    ctx, parent, single, grp = _build(capi, ivf, G)
    q = make_queries(256, d, seed=42, like=ivf["x"])
    gi0, _ = grp.search(parent, q, 4, 10, "l2")
    # move every centroid (list p gets the centroid of list p + 1): remove + add under the same ids (QuakeIndex::modify,
    # quake_index.cpp:147-150)
    moved = np.ascontiguousarray(np.roll(ivf["centroids"], -1, axis=0))
    parent.remove_ids(np.arange(nl, dtype=np.int64))
    parent.add_entries(0, np.arange(nl, dtype=np.int64), moved)
    si, sd = ctx.search(parent, single, q, 4, 10, "l2")
    gi, gd = grp.search(parent, q, 4, 10, "l2")
    assert (gi == si).all() and (_bits(gd) == _bits(sd)).all()
    assert not (gi == gi0).all()


def test_group_aps_equals_single_store(capi):
    """qk_group_search_aps == qk_search_aps: ids, distance bits and partitions visited per query"""
    ivf = make_ivf(80000, 32, 256, seed=51)
    ctx, parent, single, grp = _build(capi, ivf, 4)
    q = make_queries(300, 32, seed=52, like=ivf["x"])
    for rt, frac in ((0.9, 0.05), (0.99, 0.2)):
        si, sd, sn = ctx.search_aps(parent, single, q, 10, "l2", rt, initial_search_fraction=frac)
        gi, gd, gn, tm = grp.search_aps(parent, q, 10, "l2", rt, initial_search_fraction=frac, timing=True)
        assert (gi == si).all() and (_bits(gd) == _bits(sd)).all() and (gn == sn).all()
        assert tm["n_items"] >= 1
        qd = torch.from_numpy(q).cuda()
        di, dd, dn = grp.search_aps(parent, qd, 10, "l2", rt, initial_search_fraction=frac)
        grp.synchronize()
        assert (_np(di) == si).all() and (_bits(dd) == _bits(sd)).all() and (_np(dn) == sn).all()
    # a plain search afterwards still returns distances (the lead's merge-key mode was put back)
    si, sd = ctx.search(parent, single, q, 4, 10, "l2")
    gi, gd = grp.search(parent, q, 4, 10, "l2")
    assert (gi == si).all() and (_bits(gd) == _bits(sd)).all()


def test_group_errors(capi):
    with pytest.raises(RuntimeError):
        capi.Group([], 16)
    with pytest.raises(RuntimeError):
        capi.Group([torch.cuda.device_count()], 16)
    g = capi.Group([0, 0], 16)
    ctx = capi.Context(0)
    p = capi.Store(ctx, 8)
    p.build_csr(np.array([0, 2], np.int64), np.arange(2, dtype=np.int64), np.zeros((2, 8), np.float32))
    with pytest.raises(RuntimeError, match="dimension"):
        g.search(p, np.zeros((4, 16), np.float32), 1, 1, "l2")
    # (k beyond the LDS pools goes through the wide-k path and the sorted-run merge since round 6 -- test_group_large_k_equals_single_store;
    #  an EMPTY group still has nothing to scan)
    with pytest.raises(RuntimeError, match="no lists"):
        g.scan(np.zeros((4, 16), np.float32), np.zeros((4, 1), np.int64) - 1, 2000, "l2")
