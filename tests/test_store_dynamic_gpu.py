"""Dynamic device partition store: append / swap-with-last remove / add-drop list / batched add, checked
against a host model with the reference's semantics (index_partition.cpp:52-102, dynamic_inverted_list.cpp:137-173),
and search parity with the oracle after a stream of mutations."""
import numpy as np
import pytest

import oracle as O
from helpers import make_ivf, make_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    yield c
    c.close()


class HostModel:
    """list number -> (ids list, vectors list) with append and the reference's remove sweep."""

    def __init__(self, ivf):
        self.parts = {p: (list(ivf["part_ids"][p]), [v for v in ivf["part_vecs"][p]]) for p in range(ivf["nlist"])}

    def add(self, p, ids, vecs):
        self.parts[p][0].extend(int(i) for i in ids)
        self.parts[p][1].extend(v for v in vecs)

    def remove(self, kill):
        kill = set(int(i) for i in kill)
        for p, (ids, vecs) in self.parts.items():
            i = 0
            while i < len(ids):  # DynamicInvertedLists::remove_vectors: swap-with-last, re-examine position i
                if ids[i] in kill:
                    ids[i], vecs[i] = ids[-1], vecs[-1]
                    ids.pop()
                    vecs.pop()
                else:
                    i += 1

    def csr(self, d):
        keys = sorted(self.parts)
        pv = [np.array(self.parts[p][1], np.float32).reshape(-1, d) for p in keys]
        pi = [np.array(self.parts[p][0], np.int64) for p in keys]
        return keys, O.csr_from_partitions(pv, pi, d)


def check_equal(store, model, d):
    for p, (ids, vecs) in model.parts.items():
        gv, gi = store.get_list(p)
        np.testing.assert_array_equal(gi, np.array(ids, np.int64))
        np.testing.assert_array_equal(gv, np.array(vecs, np.float32).reshape(-1, d))


def test_mutation_stream_matches_host_model_and_oracle(ctx):
    from quake_amd.capi import Store
    d, nlist = 40, 9
    ivf = make_ivf(5000, d, nlist, seed=41, empty=(4,))
    s = Store(ctx, d)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    m = HostModel(ivf)
    rng = np.random.default_rng(42)
    next_id = 100000
    for step in range(12):
        kind = step % 3
        if kind == 0:  # per-list append, large enough to force extent relocation
            p = int(rng.integers(0, nlist))
            n = int(rng.integers(1, 900))
            v = rng.standard_normal((n, d)).astype(np.float32)
            ids = np.arange(next_id, next_id + n, dtype=np.int64)
            next_id += n
            s.add_entries(p, ids, v)
            m.add(p, ids, v)
        elif kind == 1:  # batched add with arbitrary assignment
            n = int(rng.integers(1, 700))
            v = rng.standard_normal((n, d)).astype(np.float32)
            ids = np.arange(next_id, next_id + n, dtype=np.int64)
            next_id += n
            a = rng.integers(0, nlist, size=n).astype(np.int64)
            s.add_batch(ids, v, a)
            for i in range(n):
                m.add(int(a[i]), ids[i:i + 1], v[i:i + 1])
        else:  # remove a random subset of resident ids (+ ids that do not exist)
            allids = np.concatenate([np.array(x[0], np.int64) for x in m.parts.values()])
            kill = rng.choice(allids, size=min(400, len(allids) // 3), replace=False)
            kill = np.concatenate([kill, np.array([10 ** 9, 10 ** 9 + 1])])
            removed = s.remove_ids(kill)
            m.remove(kill)
            assert removed == len(kill) - 2
        assert s.ntotal() == sum(len(x[0]) for x in m.parts.values())
    check_equal(s, m, d)
    # get_vector through the id index
    some = m.parts[2][0][5]
    np.testing.assert_array_equal(s.get_vector(some), m.parts[2][1][5])
    assert s.get_vector(10 ** 9) is None
    # drop / re-add a list
    s.remove_list(1)
    del m.parts[1]
    s.add_list(nlist)
    m.parts[nlist] = ([], [])
    v = rng.standard_normal((50, d)).astype(np.float32)
    s.add_entries(nlist, np.arange(next_id, next_id + 50), v)
    m.add(nlist, np.arange(next_id, next_id + 50), v)
    check_equal(s, m, d)
    # search over the mutated store == oracle over the host model (partition numbers are the model's keys)
    keys, (vecs, ids, offs) = m.csr(d)
    dense_offs = np.zeros(max(keys) + 2, np.int64)
    sizes = {p: len(m.parts[p][0]) for p in keys}
    for p in range(max(keys) + 1):
        dense_offs[p + 1] = dense_offs[p] + sizes.get(p, 0)
    order = np.concatenate([np.arange(offs[keys.index(p)], offs[keys.index(p) + 1]) for p in range(max(keys) + 1) if p in sizes])
    q = make_queries(30, d, seed=43)
    pids = np.stack([rng.permutation(np.array(keys))[:4] for _ in range(30)]).astype(np.int64)
    gi, gd = ctx.scan(s, q, pids, 10, "l2")
    oi, od = O.batched_serial_scan(q, vecs[order], ids[order], dense_offs, pids, 10, "l2")
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_skewed_inserts_compact_the_arena(ctx):
    """Every doubling of a hot partition abandons its old extent; once a quarter of the arena is abandoned the store compacts
    instead of growing (qk_store.hip compact_arena).  Contents, order and search results must be unaffected, and the arena
    must stay within a small multiple of the live data."""
    from quake_amd.capi import Store
    d, nlist = 24, 12
    ivf = make_ivf(6000, d, nlist, seed=61)
    s = Store(ctx, d)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    model = HostModel(ivf)
    rng = np.random.default_rng(62)
    next_id = 100000
    for step in range(60):  # hot lists 2 and 7 get almost everything, in growing batches
        n = int(rng.integers(200, 1500))
        p = 2 if step % 3 else 7
        vecs = rng.standard_normal((n, d)).astype(np.float32)
        ids = np.arange(next_id, next_id + n, dtype=np.int64)
        next_id += n
        if step % 4 == 0:  # batched form, a few rows for other lists too
            assign = np.full(n, p, np.int64)
            assign[::50] = rng.integers(0, nlist, assign[::50].shape[0])
            s.add_batch(ids, vecs, assign)
            for q_ in range(nlist):
                m = assign == q_
                if m.any():
                    model.add(q_, ids[m], vecs[m])
        else:
            s.add_entries(p, ids, vecs)
            model.add(p, ids, vecs)
        if step % 7 == 6:
            kill = rng.choice(np.array(model.parts[p][0]), 300, replace=False)
            s.remove_ids(kill)
            model.remove(kill)
    check_equal(s, model, d)
    live_bytes = s.ntotal() * d * 4
    # bump-only growth would be >> 10x here.  What bounds the arena: it doubles when it must grow, a list's extent doubles, and up to
    # a quarter of it may be abandoned before a compaction pays (2 x 2 x 4/3); where in its doubling sequence the arena stands at the
    # end depends on the compaction's form (in place: the arena keeps its size; replaced: sized to what is live)
    assert s.device_bytes() < 6.0 * live_bytes, (s.device_bytes(), live_bytes)
    assert s.counters()["arena_compactions"] >= 1, s.counters()
    keys, (vecs, ids_, offsets) = model.csr(d)
    q = make_queries(30, d, seed=63, like=ivf["x"])
    pids = np.tile(np.arange(nlist, dtype=np.int64), (30, 1))
    gi, gd = ctx.scan(s, q, pids, 10, "l2")
    oi, od = O.batched_serial_scan(q, vecs, ids_, offsets, pids, 10, "l2")
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("bounce_rows", [16, 48, 1024, None])
def test_compaction_in_place_keeps_every_list(ctx, bounce_rows, monkeypatch):
    """compact_arena (qk_store.hip) slides the live extents down inside the arena through a bounce buffer, in batches; an extent
    larger than the buffer goes in pieces.  With a buffer of one tile (16 rows), three tiles, 1024 rows and the product's 1 GiB:
    lists dropped in the middle of the arena, hot lists that outgrow their extents again and again -- contents and order of every
    list, and a scan of all of them, as the host model and the oracle have them."""
    from quake_amd.capi import Store
    if bounce_rows is None:
        monkeypatch.delenv("QK_COMPACT_BOUNCE_ROWS", raising=False)
    else:
        monkeypatch.setenv("QK_COMPACT_BOUNCE_ROWS", str(bounce_rows))
    d, nlist = 20, 40
    ivf = make_ivf(30000, d, nlist, seed=71)
    s = Store(ctx, d)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    model = HostModel(ivf)
    rng = np.random.default_rng(72)
    next_id = 1000000
    for p in (3, 4, 11, 12, 13, 25, 31):  # abandoned extents all over the arena
        s.remove_list(p)
        del model.parts[p]
    live = sorted(model.parts)
    for step in range(80):
        p = int(live[int(rng.integers(0, 4))]) if step % 5 else int(rng.choice(live))  # four hot lists, now and then any list
        n = int(rng.integers(100, 2500))
        vecs = rng.standard_normal((n, d)).astype(np.float32)
        ids = np.arange(next_id, next_id + n, dtype=np.int64)
        next_id += n
        s.add_entries(p, ids, vecs)
        model.add(p, ids, vecs)
        if step % 9 == 8:
            kill = rng.choice(np.array(model.parts[p][0]), 200, replace=False)
            s.remove_ids(kill)
            model.remove(kill)
        if step % 20 == 19:
            check_equal(s, model, d)
    cnt = s.counters()
    assert cnt["arena_compactions"] >= 1, cnt
    check_equal(s, model, d)
    for _ in range(3):
        _search_equals_oracle(ctx, s, model, d, rng, "l2", 10)
    s.close()


def _search_equals_oracle(ctx, s, m, d, rng, metric, k):
    keys, (vecs, ids, offs) = m.csr(d)
    live = [p for p in keys]
    sizes = {p: len(m.parts[p][0]) for p in keys}
    top = max(keys) + 1
    dense_offs = np.zeros(top + 1, np.int64)
    for p in range(top):
        dense_offs[p + 1] = dense_offs[p] + sizes.get(p, 0)
    order = np.concatenate([np.arange(offs[keys.index(p)], offs[keys.index(p) + 1]) for p in range(top) if p in sizes] or
                           [np.zeros(0, np.int64)]).astype(np.int64)
    nq = int(rng.choice([1, 5, 40, 300]))
    q = rng.standard_normal((nq, d)).astype(np.float32)
    P = min(len(live), int(rng.choice([1, 3, 8])))
    pids = np.stack([rng.permutation(np.array(live))[:P] for _ in range(nq)]).astype(np.int64)
    gi, gd = ctx.scan(s, q, pids, k, metric)
    oi, od = O.batched_serial_scan(q, vecs[order], ids[order], dense_offs, pids, k, metric)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_mutation_streams(ctx, seed):
    """40 random operations per stream -- per-list and batched appends, removals (resident and unknown ids), list drop /
    creation, in-place refinement of random list subsets with 0-2 Lloyd iterations -- with the store compared to the host
    model (contents AND row order) after every operation and a scan compared to the oracle every few operations."""
    from quake_amd.capi import Store
    rng = np.random.default_rng(900 + seed)
    d = int(rng.choice([8, 40, 100, 128]))
    nlist = int(rng.choice([3, 9, 30]))
    metric = str(rng.choice(["l2", "ip"]))
    ivf = make_ivf(int(rng.choice([50, 3000])), d, nlist, seed=901 + seed, metric=metric)
    s = Store(ctx, d)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    m = HostModel(ivf)
    next_id, next_list = 10 ** 6, nlist
    for step in range(40):
        live = sorted(m.parts)
        kind = int(rng.integers(0, 6))
        if kind == 0 and live:
            p = int(rng.choice(live))
            n = int(rng.integers(1, 600))
            v = rng.standard_normal((n, d)).astype(np.float32)
            ids = np.arange(next_id, next_id + n, dtype=np.int64)
            next_id += n
            s.add_entries(p, ids, v)
            m.add(p, ids, v)
        elif kind == 1 and live:
            n = int(rng.integers(1, 500))
            v = rng.standard_normal((n, d)).astype(np.float32)
            ids = np.arange(next_id, next_id + n, dtype=np.int64)
            next_id += n
            a = rng.choice(live, size=n).astype(np.int64)
            s.add_batch(ids, v, a)
            for i in range(n):
                m.add(int(a[i]), ids[i:i + 1], v[i:i + 1])
        elif kind == 2:
            allids = np.concatenate([np.array(x[0], np.int64) for x in m.parts.values()] + [np.zeros(0, np.int64)])
            if len(allids):
                kill = rng.choice(allids, size=max(1, min(300, len(allids) // 3)), replace=False)
                kill = np.concatenate([kill, np.array([10 ** 9 + step])])
                assert s.remove_ids(kill) == len(kill) - 1
                m.remove(kill)
        elif kind == 3 and len(live) > 2:
            p = int(rng.choice(live))
            s.remove_list(p)
            del m.parts[p]
        elif kind == 4:
            s.add_list(next_list)
            m.parts[next_list] = ([], [])
            next_list += 1
        elif kind == 5 and len(live) >= 2:
            sub = [int(p) for p in rng.permutation(live)[:int(rng.integers(2, min(len(live), 6) + 1))]]
            if sum(len(m.parts[p][0]) for p in sub) == 0:
                continue
            cent = rng.standard_normal((len(sub), d)).astype(np.float32)
            if metric == "ip":
                cent /= np.linalg.norm(cent, axis=1, keepdims=True)
            iters = int(rng.integers(0, 3))
            pv = [np.array(m.parts[p][1], np.float32).reshape(-1, d) for p in sub]
            pi = [np.array(m.parts[p][0], np.int64) for p in sub]
            vecs, ids, offs = O.csr_from_partitions(pv, pi, d)
            rc, rv, ri, ro = O.kmeans_refine_partitions(cent, vecs, ids, offs, metric, iters)
            if np.isnan(rc).any():
                continue  # an emptied cluster: the store refuses (QK_ERR_INVALID), covered in test_kmeans_gpu
            gc = s.refine_lists(np.array(sub, np.int64), cent, metric, iters)
            np.testing.assert_array_equal(np.asarray(gc).view(np.uint32), rc.view(np.uint32))
            for j, p in enumerate(sub):
                m.parts[p] = (list(ri[ro[j]:ro[j + 1]]), [v for v in rv[ro[j]:ro[j + 1]]])
        assert s.ntotal() == sum(len(x[0]) for x in m.parts.values())
        assert sorted(int(p) for p in s.list_ids()) == sorted(m.parts)
        check_equal(s, m, d)
        if step % 5 == 4 and m.parts and s.ntotal() > 0:
            _search_equals_oracle(ctx, s, m, d, rng, metric, int(rng.choice([1, 10, 40])))


def test_remove_of_an_id_held_many_times(ctx):
    """The store does not enforce unique ids: a list that holds ONE id many times loses all of those rows when that id is
    removed -- more row moves than ids asked for.  The staging buffer of the move list is sized for n ids; a longer move list must
    go through it in chunks, not past its end (round-3 advisor finding: silent arena corruption from ~32 such moves on)."""
    from quake_amd.capi import Store
    d = 24
    rng = np.random.default_rng(5)
    ivf = make_ivf(3000, d, 3, seed=77)
    s = Store(ctx, d)
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    m = HostModel(ivf)
    dup_id = 7_000_000
    # interleave 400 copies of one id with 400 fresh rows in list 1, so that the sweep swaps rows hundreds of times
    n = 800
    v = rng.standard_normal((n, d)).astype(np.float32)
    ids = np.where(np.arange(n) % 2 == 0, dup_id, 8_000_000 + np.arange(n)).astype(np.int64)
    s.add_entries(1, ids, v)
    m.add(1, ids, v)
    before = s.ntotal()
    removed = s.remove_ids(np.array([dup_id], np.int64))
    m.remove([dup_id])
    assert removed == 400 and s.ntotal() == before - 400
    check_equal(s, m, d)
    # the neighbours of the staging buffer's user: the other lists are untouched and the store still searches correctly
    keys, (vecs, aids, offs) = m.csr(d)
    q = make_queries(16, d, seed=3, like=ivf["x"])
    pids = np.tile(np.array(keys, np.int64), (16, 1))
    gi, gd = ctx.scan(s, q, pids, 10, "l2")
    oi, od = O.batched_serial_scan(q, vecs, aids, offs, np.tile(np.arange(len(keys), dtype=np.int64), (16, 1)), 10, "l2")
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
    s.close()


def test_large_removals_take_the_threaded_paths(ctx):
    """Millions of rows: the id index is built by several threads (QkIdMap::build_from_segments) and a removal's touched lists are
    swept by several threads (qk_store_remove_ids) -- same lists, same row order as the one-thread semantics (numpy model of the
    swap-with-last sweep per list), ids held twice included; get_vector and a second removal go through the built index."""
    from quake_amd.capi import Store
    rng = np.random.default_rng(77)
    d, nlist, n = 4, 900, 4_300_000
    sizes = rng.integers(3000, 6500, nlist)
    sizes = (sizes * (n / sizes.sum())).astype(np.int64)
    n = int(sizes.sum())
    offs = np.zeros(nlist + 1, np.int64)
    offs[1:] = np.cumsum(sizes)
    ids = rng.permutation(n).astype(np.int64)
    ids[offs[7]] = ids[offs[3]]  # one id held by two lists (the first one answers get_vector; a removal takes both)
    vecs = rng.standard_normal((n, d)).astype(np.float32)
    s = Store(ctx, d)
    s.build_csr(offs, ids, vecs)
    kill = rng.choice(n, 600_000, replace=False).astype(np.int64)
    kill[0] = ids[offs[3]]
    removed = s.remove_ids(kill)

    def model(list_ids, killset):  # the sweep of one list: swap-with-last, re-examine
        cur = list(list_ids)
        i = 0
        while i < len(cur):
            if cur[i] in killset:
                cur[i] = cur[-1]
                cur.pop()
            else:
                i += 1
        return cur

    killset = set(int(v) for v in kill)
    present = np.isin(ids, kill)
    assert removed == int(present.sum())
    assert s.ntotal() == n - removed
    for p in list(rng.choice(nlist, 25, replace=False)) + [3, 7]:
        want = model(ids[offs[p]:offs[p + 1]].tolist(), killset)
        gv, gi = s.get_list(int(p))
        assert gi.tolist() == want
        pos = {int(v): t for t, v in enumerate(ids[offs[p]:offs[p + 1]])}
        np.testing.assert_array_equal(gv, vecs[offs[p] + np.array([pos[v] for v in want], np.int64)] if want else np.zeros((0, d), np.float32))
    # the index built for the removal answers lookups; removed ids are gone, a second removal finds nothing
    alive = ids[~present]
    for v in alive[rng.choice(len(alive), 50, replace=False)]:
        row = int(np.nonzero(ids == v)[0][0])
        np.testing.assert_array_equal(s.get_vector(int(v)), vecs[row])
    assert s.get_vector(int(kill[5])) is None
    assert s.remove_ids(kill[:1000]) == 0
    s.close()
