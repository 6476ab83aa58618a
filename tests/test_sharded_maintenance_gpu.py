"""Sharded maintenance on the product path (GpuPartitions over libquake_hip.so): world = 1 in process must do exactly what
index.QuakeIndex does, and two ranks (two PROCESSES sharing GPU 0 over gloo, collectives staged through the host) must end
with the same index as one process applying the same operations -- compared through searches (bit-exact ids and
distances) and through the partition contents."""
import os
import tempfile
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _corpus(metric, d=32, nlist=16, n=40000, seed=91):
    from helpers import make_ivf
    ivf = make_ivf(n, d, nlist, seed=seed, metric=metric)
    keep = {5: 12, 10: 12}  # two tiny partitions: delete candidates
    pv = [v[:keep.get(p, len(v))] for p, v in enumerate(ivf["part_vecs"])]
    pi = [i[:keep.get(p, len(i))] for p, i in enumerate(ivf["part_ids"])]
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum([len(i) for i in pi])
    ivf.update(part_vecs=pv, part_ids=pi, vecs=np.concatenate(pv), ids=np.concatenate(pi), offsets=offsets)
    return ivf


def _policy_params(window, iters):
    import quake_amd as quake
    p = quake.MaintenancePolicyParams()
    p.window_size = window
    p.refinement_radius = 3
    p.refinement_iterations = iters
    p.min_partition_size = 32
    p.delete_threshold_ns = 0.1
    p.split_threshold_ns = 0.1
    return p


def _cost(d):
    from quake_amd.maintenance import ListScanLatencyEstimator, MaintenanceCostEstimator
    lat = ListScanLatencyEstimator(d, [1, 2, 4, 16, 64, 256, 1024, 4096, 16384, 65536], [1, 4, 16, 64, 256], 1,
                                   profile_fn=lambda n, k: 100.0 + 1.0 * n)
    return MaintenanceCostEstimator(d, 0.9, 10, latency_estimator=lat)


def _plain(ivf, metric):
    from quake_amd.index import QuakeIndex
    return QuakeIndex.from_partitions(ivf["centroids"], ivf["offsets"], ivf["ids"], ivf["vecs"], metric)


def _search_plain(ix, qd, nprobe, k):
    import quake_amd as quake
    sp = quake.SearchParams()
    sp.nprobe, sp.k, sp.batched_scan = nprobe, k, True
    r = ix.search(qd, sp)
    return r.ids, r.distances


def _apply(s, plain, ops):
    """the same maintenance script on a ShardedPartitions-like object (collective) or on a plain QuakeIndex"""
    for op in ops:
        if op[0] == "split":
            cl = s._split_partitions(op[1])
            s._delete_partitions(op[1], reassign=False)
            s._add_partitions(cl)
        elif op[0] == "delete":
            s._delete_partitions(op[1], reassign=True)
        elif op[0] == "refine":
            s.refine_partitions(torch.tensor(op[1], dtype=torch.int64) if plain else op[1], op[2])


def _same_lists(sh, ix, rank, world, ordered=True):
    """owned partitions of the sharded index hold the rows the plain index holds (in the same order, unless told
    otherwise); centroids are replicated bit for bit"""
    assert sh.partitions._list_ids() == ix._list_ids()
    pids = ix._list_ids()
    a = sh.local.centroids(pids)
    b = ix.parent.get(torch.tensor(pids, dtype=torch.int64)).numpy()
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    for p in pids:
        v, i = sh.local.get_list(p)  # (the product `local` keeps row blocks on the device: CUDA tensor + host ids)
        assert torch.is_tensor(v) and v.is_cuda
        v = v.cpu().numpy()
        rv, ri = ix._store.get_list(p)
        if p % world != rank:
            assert len(i) == 0, p
            continue
        if ordered:
            assert (i == ri).all(), p
        o, ro = np.argsort(i), np.argsort(ri)
        assert (i[o] == ri[ro]).all(), p
        assert (v[o].view(np.uint32) == rv[ro].view(np.uint32)).all(), p
    assert sh.partitions._partition_sizes(pids) == ix._partition_sizes(pids)


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_world1_equals_quake_index(metric):
    """ShardedPartitions with one rank == QuakeIndex's own split / delete / refine (3 Lloyd iterations included: same
    assignment kernel, same accumulation order, same mean update) and the same policy run."""
    from quake_amd.sharded_maintenance import ShardedQuakeIndex
    ivf = _corpus(metric)
    nlist, d = ivf["nlist"], ivf["d"]
    ix = _plain(ivf, metric)
    sh = ShardedQuakeIndex(_plain(ivf, metric), None, 1, 0)
    ops = [("split", [3, 8]), ("delete", [5, 10]), ("refine", [0, 1, 2, nlist, nlist + 1], 3), ("refine", [4, 6, nlist + 2], 0)]
    _apply(sh.partitions, False, ops)
    _apply(ix, True, ops)
    _same_lists(sh, ix, 0, 1)
    # row ORDER too (world 1): refine leaves the rows of a list in the reference's append order
    for p in [0, 1, 2, nlist, 4]:
        assert (sh.local.get_list(p)[1] == ix._store.get_list(p)[1]).all()
    # the policy
    q = torch.from_numpy(ivf["x"][:256]).cuda()
    for s in (sh, ix):
        s.initialize_maintenance_policy(_policy_params(256, 2), cost_estimator=_cost(d))
        s.track_hits = True
    sh.search(q, 2, 10)
    _search_plain(ix, q, 2, 10)
    ta, tb = sh.maintenance(), ix.maintenance()
    assert (ta.n_splits, ta.n_deletes) == (tb.n_splits, tb.n_deletes) and ta.n_splits >= 1
    _same_lists(sh, ix, 0, 1)
    gi, gd = sh.search(q, 4, 10)
    ri, rd = _search_plain(ix, q, 4, 10)
    assert torch.equal(gi, ri) and torch.equal(gd.view(torch.int32), rd.view(torch.int32))


def _world2_worker(rank, world, port, metric, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import make_queries
        from quake_amd.capi import Context, Store
        from quake_amd.sharded_maintenance import ShardedQuakeIndex
        torch.cuda.set_device(0)
        ivf = _corpus(metric)
        nlist, d = ivf["nlist"], ivf["d"]
        q = make_queries(128, d, seed=92, like=ivf["x"], metric=metric)
        qd = torch.from_numpy(q).cuda()
        ix = _plain(ivf, metric)  # the one-process index, same operations
        sh = ShardedQuakeIndex.from_global(dist, world, rank, ivf["centroids"], ivf["offsets"], ivf["ids"], ivf["vecs"], metric)

        def same_search(nprobe, k):
            gi, gd = sh.search(qd, nprobe, k)
            ri, rd = _search_plain(ix, qd, nprobe, k)
            torch.cuda.synchronize()
            assert torch.equal(gi, ri), (rank, nprobe, k)
            assert torch.equal(gd.view(torch.int32), rd.view(torch.int32)), (rank, nprobe, k)

        same_search(3, 10)
        # split / delete / one-pass refine: exact
        ops = [("split", [3, 8]), ("delete", [5, 10]), ("refine", [0, 1, 2, nlist, nlist + 1, nlist + 3], 1)]
        _apply(sh.partitions, False, ops)
        _apply(ix, True, ops)
        _same_lists(sh, ix, rank, world)
        same_search(3, 10)
        same_search(sh.nlist(), 50)
        # dynamic updates between maintenance rounds
        rng = np.random.default_rng(93)
        nx = (ivf["x"][rng.integers(0, len(ivf["x"]), 500)] + 0.01 * rng.standard_normal((500, d))).astype(np.float32)
        if metric == "ip":
            nx /= np.linalg.norm(nx, axis=1, keepdims=True)
        nid = np.arange(10_000_000, 10_000_500, dtype=np.int64)
        stored = sh.add(nx, nid)
        ix.add(torch.from_numpy(nx), torch.from_numpy(nid))
        removed = sh.remove(np.concatenate([nid[:100], ivf["ids"][:200]]))
        ix.remove(torch.from_numpy(np.concatenate([nid[:100], ivf["ids"][:200]])))
        tot = sh.partitions.comm.all_sum(np.array([stored, removed], np.int64))
        assert tot.tolist() == [500, 300] and sh.ntotal() == ix.ntotal()
        _same_lists(sh, ix, rank, world)
        # the policy on the hit window of real searches (refinement_iterations 1: exact)
        for s in (sh, ix):
            s.initialize_maintenance_policy(_policy_params(256, 1), cost_estimator=_cost(d))
            s.track_hits = True
        hot = torch.from_numpy(ivf["x"][:256]).cuda()
        sh.search(hot, 2, 10)
        _search_plain(ix, hot, 2, 10)
        ta, tb = sh.maintenance(), ix.maintenance()
        assert (ta.n_splits, ta.n_deletes) == (tb.n_splits, tb.n_deletes) and ta.n_splits >= 1, (ta.n_splits, tb.n_splits)
        for s in (sh, ix):
            s.track_hits = False
        _same_lists(sh, ix, rank, world)
        same_search(4, 10)
        # three Lloyd iterations across the shards (partials reduced in rank order: centroids are the sharded algorithm's
        # own, equal on both ranks): nothing lost, exhaustive probing is still exact k-NN of the same resident set
        pids = sh.partitions._list_ids()[:8]
        sh.refine_partitions(pids, 3)
        ix.refine_partitions(torch.tensor(pids, dtype=torch.int64), 3)
        assert sh.ntotal() == ix.ntotal()
        c = sh.local.centroids(pids)
        g = sh.partitions.comm.all_gather(c)
        assert (g[0].view(np.uint32) == g[1].view(np.uint32)).all()
        np.testing.assert_allclose(c, ix.parent.get(torch.tensor(pids, dtype=torch.int64)).numpy(), rtol=1e-4, atol=1e-5)
        same_search(sh.nlist(), 20)
        # default cost model: rank 0 profiles its device, the grid is broadcast
        sh.initialize_maintenance_policy(_policy_params(64, 1))
        grid = np.array(sh.partitions._policy().cost_estimator_.get_latency_estimator().scan_latency_model_)
        gg = sh.partitions.comm.all_gather(grid)
        assert (gg[0] == gg[1]).all() and (gg[0] > 0).all()
        open(os.path.join(ret, "rank%d.ok" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_world2_processes(metric):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (each rank leaves a file: a multiprocessing.Manager is a FORK of this process, HIP runtime and all, and its server
    #  died now and then in long sessions)
    ret = tempfile.mkdtemp(prefix="qk_ranks_")
    mp.spawn(_world2_worker, args=(2, port, metric, ret), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(ret, "rank%d.ok" % r)) for r in range(2))


def _build_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quake_amd.capi import Context, Store
        from quake_amd.sharded_maintenance import ShardedQuakeIndex
        torch.cuda.set_device(0)
        rng = np.random.default_rng(101)
        d, n_r, nlist = 24, 20000, 32
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
        shards = [(cent[rng.integers(0, nlist, n_r)] + 0.4 * rng.standard_normal((n_r, d))).astype(np.float32) for _ in range(world)]
        ids = [np.arange(r * n_r, (r + 1) * n_r, dtype=np.int64) for r in range(world)]
        sh = ShardedQuakeIndex.build(dist, world, rank, shards[rank], ids[rank], nlist, "l2", niter=3, seed=5)
        assert sh.ntotal() == world * n_r and sh.nlist() == nlist
        for p in sh.partitions._list_ids():
            assert p % world == rank or sh.local.list_size(p) == 0
        # exhaustive probing == exact k-NN over the union of the shards (one flat list on the same GPU)
        ctx = Context(0)
        flat = Store(ctx, d)
        allx, alli = np.concatenate(shards), np.concatenate(ids)
        flat.build_csr(np.array([0, len(alli)], np.int64), alli, allx)
        q = torch.from_numpy((allx[rng.integers(0, len(allx), 64)] + 0.05 * rng.standard_normal((64, d))).astype(np.float32)).cuda()
        gi, gd = sh.search(q, nlist, 10)
        ri, rd = ctx.search(None, flat, q, 1, 10, "l2")
        torch.cuda.synchronize()
        assert torch.equal(gi, ri) and torch.equal(gd.view(torch.int32), rd.view(torch.int32))
        # a realistic probe count finds nearly all of it
        gi2, _ = sh.search(q, 4, 10)
        rec = np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(gi2.cpu().tolist(), ri.cpu().tolist())])
        assert rec > 0.9, rec
        open(os.path.join(ret, "rank%d.ok" % rank), "w").close()
    finally:
        dist.destroy_process_group()


def test_sharded_build_world2():
    """QuakeIndex::build across two ranks: cross-shard k-means, every row routed once to the owner of its list"""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (each rank leaves a file: a multiprocessing.Manager is a FORK of this process, HIP runtime and all, and its server
    #  died now and then in long sessions)
    ret = tempfile.mkdtemp(prefix="qk_ranks_")
    mp.spawn(_build_worker, args=(2, port, ret), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(ret, "rank%d.ok" % r)) for r in range(2))


def _random_worker(rank, world, port, seed, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quake_amd.sharded_maintenance import ShardedQuakeIndex
        torch.cuda.set_device(0)
        rng = np.random.default_rng(4000 + seed)  # the same stream of decisions on both ranks
        metric = str(rng.choice(["l2", "ip"]))
        ivf = _corpus(metric, d=int(rng.choice([16, 64])), nlist=int(rng.choice([6, 16])), n=int(rng.choice([6000, 30000])), seed=4100 + seed)
        d = ivf["d"]
        ix = _plain(ivf, metric)
        sh = ShardedQuakeIndex.from_global(dist, world, rank, ivf["centroids"], ivf["offsets"], ivf["ids"], ivf["vecs"], metric)
        next_id = 50_000_000
        for step in range(14):
            live = ix._list_ids()
            sizes = dict(zip(live, ix._partition_sizes(live)))
            kind = int(rng.integers(0, 6))
            if kind == 0:  # split one or two partitions that are large enough
                big = [p for p in live if sizes[p] >= 16]
                if not big:
                    continue
                pick = [int(p) for p in rng.permutation(big)[:int(rng.integers(1, 3))]]
                op = ("split", pick)
                _apply(sh.partitions, False, [op])
                _apply(ix, True, [op])
            elif kind == 1 and len(live) > 3:  # delete (with reassignment) one partition
                op = ("delete", [int(rng.choice(live))])
                _apply(sh.partitions, False, [op])
                _apply(ix, True, [op])
            elif kind == 2 and len(live) >= 2:  # one assignment pass over a random subset
                sub = [int(p) for p in rng.permutation(live)[:int(rng.integers(2, min(len(live), 6) + 1))]]
                if sum(sizes[p] for p in sub) == 0:
                    continue
                op = ("refine", sub, int(rng.integers(0, 2)))
                _apply(sh.partitions, False, [op])
                _apply(ix, True, [op])
            elif kind == 3:
                n = int(rng.integers(1, 400))
                nx = rng.standard_normal((n, d)).astype(np.float32)
                nid = np.arange(next_id, next_id + n, dtype=np.int64)
                next_id += n
                sh.add(nx, nid)
                ix.add(torch.from_numpy(nx), torch.from_numpy(nid))
            elif kind == 4 and ix.ntotal() > 50:
                allids = ix.get_ids().numpy()
                kill = rng.choice(allids, size=min(len(allids) // 5, 300), replace=False)
                sh.remove(kill)
                ix.remove(torch.from_numpy(kill))
            assert sh.ntotal() == ix.ntotal() and sh.nlist() == ix.nlist(), (seed, step, kind)
            _same_lists(sh, ix, rank, world)
            nq = 64
            q = torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float32)).cuda()
            nprobe = int(rng.choice([1, 3, sh.nlist()]))
            k = int(rng.choice([1, 10, 40]))
            gi, gd = sh.search(q, min(nprobe, sh.nlist()), k)
            ri, rd = _search_plain(ix, q, min(nprobe, sh.nlist()), k)
            torch.cuda.synchronize()
            assert torch.equal(gi, ri), (seed, step, kind, nprobe, k)
            assert torch.equal(gd.view(torch.int32), rd.view(torch.int32)), (seed, step, kind, nprobe, k)
        open(os.path.join(ret, "rank%d.ok" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("QK_RANDOM_SHARDED", "3")))))  # (a one-off run of 30 passed)
def test_world2_random_streams(seed):
    """random maintenance / update streams on two ranks against one process applying the same operations"""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (each rank leaves a file: a multiprocessing.Manager is a FORK of this process, HIP runtime and all, and its server
    #  died now and then in long sessions)
    ret = tempfile.mkdtemp(prefix="qk_ranks_")
    mp.spawn(_random_worker, args=(2, port, seed, ret), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(ret, "rank%d.ok" % r)) for r in range(2))


def _replay_worker(rank, world, port, wdir, ret, maintain=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import quake_amd as quake
        from quake_amd.sharded_maintenance import ShardedQuakeIndex
        from quake_amd.workload import replay_workload
        torch.cuda.set_device(0)
        base = torch.load(os.path.join(wdir, "base_vectors.pt"), weights_only=True).to(torch.float32)
        first = torch.load(os.path.join(wdir, "initial_indices.pt"), weights_only=True).to(torch.int64)
        bp = quake.IndexBuildParams()
        bp.metric, bp.nlist, bp.niter = "l2", 8, 3
        sp = quake.SearchParams()
        sp.k, sp.nprobe, sp.batched_scan = 5, 3, True
        plain = quake.QuakeIndex()
        plain.build(base[first], first, bp)
        pids = [int(p) for p in plain._store.list_ids()]
        lists = [plain._store.get_list(p) for p in pids]
        offs = np.concatenate([[0], np.cumsum([len(i) for _, i in lists])]).astype(np.int64)
        sh = ShardedQuakeIndex.from_global(dist, world, rank, plain.parent.get(torch.tensor(pids)).numpy(), offs,
                                           np.concatenate([i for _, i in lists]), np.concatenate([v for v, _ in lists]), "l2")
        mp_ = None
        if maintain:  # maintenance after every operation on both (split / delete / refine as collectives on the sharded one); the cost
            mp_ = _policy_params(128, 1)  # model is a fixed grid, so both indexes take the same decisions
            for ix in (plain, sh):
                ix.initialize_maintenance_policy(mp_, cost_estimator=_cost(16))
                ix.track_hits = True
        ra = replay_workload(wdir, os.path.join(wdir, f"plain_r{rank}"), "plain", nlist=8, search_params=sp, index=plain,
                             maintenance_params=mp_, keep_policy=True)
        rb = replay_workload(wdir, os.path.join(wdir, f"sharded_r{rank}"), "sharded", nlist=8, search_params=sp, index=sh,
                             maintenance_params=mp_, keep_policy=True)
        assert len(ra) == len(rb) > 0
        for a, b in zip(ra, rb):  # same resident set, same answers (recall is computed from the ids found)
            assert (a["operation_type"], a["n_total"], a["n_list"]) == (b["operation_type"], b["n_total"], b["n_list"]), (a, b)
            assert a["recall"] == b["recall"], (a, b)
            assert b["n_total"] == b["n_resident"]
            if maintain:
                assert (a["n_splits"], a["n_deletes"]) == (b["n_splits"], b["n_deletes"]), (a, b)
        if maintain:
            assert sum(r["n_splits"] + r["n_deletes"] for r in rb) > 0, "the policy never acted: the replay would prove nothing"
        open(os.path.join(ret, "rank%d.ok" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,maintain", [(2, False), (8, True)])
def test_world2_workload_replay(tmp_path, world, maintain):
    """the dynamic-workload harness over a sharded index: the ranks replay a runbook (inserts, deletes, query batches) and get the
    recalls and resident counts of one process replaying it on a plain index.  (8, True) = BASELINE.json configs[4] in CI size: 8
    ranks (over gloo, sharing this box's GPU), maintenance() after every operation -- split / delete / refine and the policy run as
    8-way collectives and take the decisions of the unsharded index."""
    import torch.multiprocessing as mp
    from quake_amd.workload import WorkloadSpec, generate_workload
    g = torch.Generator().manual_seed(77)
    cent = torch.randn(8, 16, generator=g) * 3
    base = cent[torch.randint(0, 8, (20000,), generator=g)] + torch.randn(20000, 16, generator=g)
    wdir = str(tmp_path / "w")
    spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.2, query_ratio=0.5, update_batch_size=500, query_batch_size=64 if maintain else 65,
                        number_of_operations=20, initial_size=8000, cluster_size=2500, seed=5)
    generate_workload(wdir, base, spec)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (each rank leaves a file: a multiprocessing.Manager is a FORK of this process, HIP runtime and all, and its server
    #  died now and then in long sessions)
    ret = tempfile.mkdtemp(prefix="qk_ranks_")
    mp.spawn(_replay_worker, args=(world, port, wdir, ret, maintain), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(ret, "rank%d.ok" % r)) for r in range(world))
