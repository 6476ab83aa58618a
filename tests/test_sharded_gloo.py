"""N>1 orchestration (quake_amd/sharded.py) under torch.distributed with the gloo backend, world_size 2, on CPU.
The per-rank arithmetic is injected (oracle-backed engine): what is tested here is the sharding by list number, the
single all-gather exchange and that merging per-rank top-k under the (key,id) order reproduces the unsharded result."""
import os
import tempfile
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """test double: local search through the CPU oracle; keys are squared L2 / dot like the GPU engine's."""

    def __init__(self, centroids, vecs, ids, offsets, metric):
        self.c, self.v, self.i, self.o, self.metric = centroids, vecs, ids, offsets, metric

    def coarse(self, q, nprobe):
        import oracle as O
        return O.coarse(q, self.c, None, nprobe, self.metric)[0]

    def scan(self, q, pids, k, out=None):
        import oracle as O
        oi, od = O.batched_serial_scan(q, self.v, self.i, self.o, pids, k, self.metric)
        key = od.copy()
        if self.metric == "l2":
            # recover the squared merge key exactly: recompute it with the oracle's canonical arithmetic
            idmap = {int(i): r for r, i in enumerate(self.i)}
            xn = O.row_norms(q)
            for a in range(oi.shape[0]):
                for b in range(oi.shape[1]):
                    if oi[a, b] >= 0:
                        r = idmap[int(oi[a, b])]
                        key[a, b] = O.lib().qo_l2sqr_expanded(float(xn[a]), float(O.row_norms(self.v[r:r + 1])[0]),
                                                              O.ip(q[a], self.v[r]))
        return oi, key

    def assign(self, x):
        import oracle as O
        return O.coarse(x, self.c, None, 1, self.metric)[0].reshape(-1)

    def add_local(self, ids, x, assign):
        # append to the owned lists in input order (IndexPartition::append), keeping the CSR arrays of this double in step
        nlist = self.o.shape[0] - 1
        for p in range(nlist):
            m = assign == p
            if not m.any():
                continue
            at = int(self.o[p + 1])
            self.v = np.concatenate([self.v[:at], x[m], self.v[at:]])
            self.i = np.concatenate([self.i[:at], ids[m], self.i[at:]])
            self.o[p + 1:] += int(m.sum())

    def remove_local(self, ids):
        keep = ~np.isin(self.i, ids)
        removed = int((~keep).sum())
        nlist = self.o.shape[0] - 1
        sizes = np.array([keep[self.o[p]:self.o[p + 1]].sum() for p in range(nlist)])
        self.v, self.i = self.v[keep], self.i[keep]
        self.o = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        return removed

    def merge(self, ids, keys):
        ids = ids.numpy() if torch.is_tensor(ids) else ids
        keys = keys.numpy() if torch.is_tensor(keys) else keys
        G, Q, k = ids.shape
        out_i = np.full((Q, k), -1, np.int64)
        out_d = np.full((Q, k), np.inf if self.metric == "l2" else -np.inf, np.float32)
        for q in range(Q):
            ci, ck = ids[:, q, :].reshape(-1), keys[:, q, :].reshape(-1)
            m = ci >= 0
            ci, ck = ci[m], ck[m]
            order = np.lexsort((ci, ck if self.metric == "l2" else -ck))[:k]
            out_i[q, :len(order)] = ci[order]
            out_d[q, :len(order)] = np.sqrt(ck[order]) if self.metric == "l2" else ck[order]
        return out_i, out_d


def _worker(rank, world, port, metric, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from helpers import make_ivf, make_queries
        from quake_amd.sharded import ShardedIndex, owner_of_list, shard_offsets
        ivf = make_ivf(6000, 24, 16, seed=3, metric=metric, empty=(2,))
        q = make_queries(18 if world == 2 else 3 * world, 24, seed=4, like=ivf["x"], metric=metric)  # (a multiple of the ranks)
        lo, rows = shard_offsets(ivf["offsets"], rank, world)
        # every list is owned by exactly one rank
        own = [owner_of_list(p, world) for p in range(16)]
        assert set(own) == set(range(world))
        eng = OracleEngine(ivf["centroids"], ivf["vecs"][rows], ivf["ids"][rows], lo, metric)
        idx = ShardedIndex(eng, dist, world, rank)
        for nprobe, k in [(1, 5), (4, 10), (16, 40)]:
            gi, gd = idx.search(q, nprobe, k)
            fi, fd = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, metric, batched_scan=True)
            assert (gi == fi).all(), (rank, nprobe, k)
            assert (gd.view(np.uint32) == fd.view(np.uint32)).all(), (rank, nprobe, k)
        # owner layout: all-to-all of the per-rank top-k, rank r keeps the answer of its slice of the batch
        idx_o = ShardedIndex(eng, dist, world, rank, result="owner")
        per = q.shape[0] // world
        for nprobe, k in [(1, 5), (16, 40)]:
            gi, gd = idx_o.search(q, nprobe, k)
            fi, fd = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, metric, batched_scan=True)
            sl = slice(rank * per, (rank + 1) * per)
            assert gi.shape == (per, k)
            assert (gi == fi[sl]).all(), (rank, nprobe, k)
            assert (gd.view(np.uint32) == fd[sl].view(np.uint32)).all(), (rank, nprobe, k)
        # dynamic updates: every rank gets the batch, each applies what it owns; the union behaves like one index
        rng = np.random.default_rng(9)
        nx = (ivf["x"][rng.integers(0, 6000, 40)] + 0.01 * rng.standard_normal((40, 24))).astype(np.float32)
        nid = np.arange(100000, 100040, dtype=np.int64)
        n_add = idx.add(nx, nid)
        rm = np.concatenate([ivf["ids"][:30], nid[:5]])
        n_rm = idx.remove(rm)
        tot = torch.tensor([n_add, n_rm])
        dist.all_reduce(tot)
        assert tot.tolist() == [40, 35], tot
        # reference: the same mutations on the unsharded CSR
        full = OracleEngine(ivf["centroids"], ivf["vecs"].copy(), ivf["ids"].copy(), ivf["offsets"].copy(), metric)
        full.add_local(nid, nx, full.assign(nx))
        full.remove_local(rm)
        for nprobe, k in [(4, 10), (16, 20)]:
            gi, gd = idx.search(q, nprobe, k)
            fi, fd = O.search(q, ivf["centroids"], full.v, full.i, full.o, nprobe, k, metric, batched_scan=True)
            # (append order inside a list differs between the sharded stores and the single store only across lists,
            #  never inside one list, so even the tie order is the same)
            assert (gi == fi).all(), (rank, nprobe, k)
        open(os.path.join(ret, "rank%d.ok" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,metric", [(2, "l2"), (2, "ip"), (8, "l2")])
def test_sharded_search_equals_unsharded(world, metric):
    """world 8 = the rank count of BASELINE.json configs[3] / [4] (16 lists: two per rank; 24 queries: three per rank): the same
    worker, every collective with 8 participants"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (each rank leaves a file: a multiprocessing.Manager is a FORK of this process, HIP runtime and all, and its server
    #  died now and then in long sessions)
    ret = tempfile.mkdtemp(prefix="qk_ranks_")
    mp.spawn(_worker, args=(world, port, metric, ret), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(ret, "rank%d.ok" % r)) for r in range(world))


class OracleKmeans:
    """test double for the per-rank k-means arithmetic (the product engine is quake_amd.capi.Context)"""

    def normalize_rows(self, x):
        import oracle as O
        return O.normalize_rows(x)

    def rand_perm(self, n, m, seed):
        import oracle as O
        return O.rand_perm(n, m, seed)

    def kmeans_assign(self, x, c, metric):
        import oracle as O
        return O.kmeans_assign(x, c, metric)

    def kmeans_accumulate(self, x, a, m, blocked=False):
        import oracle as O
        return O.kmeans_accumulate(x, a, m, blocked=blocked)

    def kmeans_update(self, sums, counts, c):
        import oracle as O
        return O.kmeans_update(sums, counts, c)


def _kmeans_shards(metric, n=3000, d=12, seed=21):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((8, d)).astype(np.float32)
    return [(cent[rng.integers(0, 8, n)] + 0.4 * rng.standard_normal((n, d))).astype(np.float32) for _ in range(2)]


def _kmeans_worker(rank, world, port, metric, m, ordered, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from helpers import sharded_kmeans_reference
        from quake_amd.sharded import sharded_kmeans
        shards = _kmeans_shards(metric)
        c, a = sharded_kmeans(OracleKmeans(), dist, shards[rank], m, metric, niter=4, seed=77, rank=rank, world=world, ordered=ordered)
        rc, ra = sharded_kmeans_reference(O, shards, m, metric, niter=4, seed=77)
        assert (c.view(np.uint32) == rc.view(np.uint32)).all(), rank  # same centroids on every rank, bit for bit
        assert (a == ra[rank]).all(), rank
        open(os.path.join(ret, "rank%d.ok" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric,m,ordered", [("l2", 16, True), ("ip", 16, True), ("l2", 8, False), ("l2", 2000, True)])
def test_sharded_kmeans_world2(metric, m, ordered):
    """cross-shard Lloyd (local assign + partial sums, reduction of [m,d] sums and [m] counts per iteration): both ranks end
    with the centroids of the single-process restatement.  m = 2000 > n/256 exercises the no-subsample branch with
    empty clusters (the split rule); ordered=False is the plain all-reduce (two ranks: a + b == b + a)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (each rank leaves a file: a multiprocessing.Manager is a FORK of this process, HIP runtime and all, and its server
    #  died now and then in long sessions)
    ret = tempfile.mkdtemp(prefix="qk_ranks_")
    mp.spawn(_kmeans_worker, args=(2, port, metric, m, ordered, ret), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(ret, "rank%d.ok" % r)) for r in range(2))


def test_sharded_kmeans_world1_is_plain_kmeans():
    import oracle as O
    from quake_amd.sharded import sharded_kmeans
    x = _kmeans_shards("l2")[0]
    c, a = sharded_kmeans(OracleKmeans(), None, x, 16, "l2", niter=3, seed=5)
    rc, ra, _ = O.kmeans(x, 16, "l2", niter=3, seed=5)
    assert (c.view(np.uint32) == rc.view(np.uint32)).all() and (a == ra).all()


def test_shard_offsets_block_and_mod():
    from quake_amd.sharded import shard_offsets
    off = np.array([0, 3, 3, 10, 12], np.int64)
    lo0, r0 = shard_offsets(off, 0, 2)
    lo1, r1 = shard_offsets(off, 1, 2)
    assert lo0.tolist() == [0, 3, 3, 10, 10] and lo1.tolist() == [0, 0, 0, 0, 2]
    assert sorted(np.concatenate([r0, r1]).tolist()) == list(range(12))
    lb0, _ = shard_offsets(off, 0, 2, lists_per_rank=2)
    assert lb0.tolist() == [0, 3, 3, 3, 3]
