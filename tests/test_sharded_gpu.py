"""GpuEngine + ShardedIndex: the product engine behind the sharded orchestration -- search, the sharded add / remove with
device tensors and the cross-shard k-means -- against the plain C-ABI calls on an identical store.  world = 1 in process,
and world = 2 as two PROCESSES sharing GPU 0 over gloo (RCCL refuses two ranks on one device; the collectives are staged
through the host there, everything else is the product path with device tensors)."""
import os
import tempfile
import socket
import sys

import numpy as np
import pytest
import torch

from helpers import make_ivf, make_queries

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_engine_world1_search_add_remove():
    from quake_amd.capi import Context, Store
    from quake_amd.sharded import GpuEngine, ShardedIndex
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ivf = make_ivf(20000, 32, 24, seed=71)
    q = make_queries(64, 32, seed=72, like=ivf["x"])
    stores = []
    for _ in range(2):
        s = Store(ctx, 32)
        s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
        stores.append(s)
    parent = Store(ctx, 32)
    parent.build_csr(np.array([0, 24], np.int64), np.arange(24, dtype=np.int64), ivf["centroids"])
    idx = ShardedIndex(GpuEngine(ctx, parent, stores[0], "l2"), None, 1, 0)
    qd = torch.from_numpy(q).cuda()
    gi, gd = idx.search(qd, 4, 10)
    ri, rd = ctx.search(parent, stores[1], qd, 4, 10, "l2")  # (the engine's squared-distance mode does not leak into the context)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(gi.cpu().numpy(), ri.cpu().numpy())
    np.testing.assert_array_equal(gd.cpu().numpy(), rd.cpu().numpy())
    # sharded add / remove with device tensors == the same mutations applied directly
    rng = np.random.default_rng(73)
    nx = torch.from_numpy((ivf["x"][rng.integers(0, 20000, 300)] + 0.01 * rng.standard_normal((300, 32))).astype(np.float32)).cuda()
    nid = torch.arange(500000, 500300, dtype=torch.int64).cuda()
    assert idx.add(nx, nid) == 300
    assign = ctx.coarse(parent, nx, 1, "l2")[0].reshape(-1)
    stores[1].add_batch(nid, nx, assign.contiguous())
    rm = np.concatenate([ivf["ids"][:100], np.arange(500000, 500050)])
    assert idx.remove(torch.from_numpy(rm).cuda()) == 150
    assert stores[1].remove_ids(rm) == 150
    assert stores[0].ntotal() == stores[1].ntotal() == 20000 + 300 - 150
    gi, gd = idx.search(qd, 24, 10)
    ri, rd = ctx.search(parent, stores[1], qd, 24, 10, "l2")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(gi.cpu().numpy(), ri.cpu().numpy())
    np.testing.assert_array_equal(gd.cpu().numpy(), rd.cpu().numpy())
    ctx.close()


def _world2_worker(rank, world, port, metric, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from helpers import make_ivf, make_queries, sharded_kmeans_reference
        from quake_amd.capi import Context, Store
        from quake_amd.sharded import GpuEngine, ShardedIndex, shard_offsets, sharded_kmeans
        torch.cuda.set_device(0)
        ctx = Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        d, nlist = 64, 48
        ivf = make_ivf(60000, d, nlist, seed=81, metric=metric, empty=(5,))
        q = make_queries(128, d, seed=82, like=ivf["x"], metric=metric)
        qd = torch.from_numpy(q).cuda()
        # unsharded reference on the same GPU through the plain C ABI
        full = Store(ctx, d)
        full.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
        parent = Store(ctx, d)
        parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), ivf["centroids"])
        # this rank's half of the lists (list p on rank p % 2)
        lo, rows = shard_offsets(ivf["offsets"], rank, world)
        mine = Store(ctx, d)
        mine.build_csr(lo, ivf["ids"][rows], ivf["vecs"][rows])
        eng = GpuEngine(ctx, parent, mine, metric)
        per = q.shape[0] // world
        for nprobe, k in [(1, 10), (8, 10), (48, 100)]:
            ri, rd = ctx.search(parent, full, qd, nprobe, k, metric)
            gi, gd = ShardedIndex(eng, dist, world, rank, result="all").search(qd, nprobe, k)
            torch.cuda.synchronize()
            assert torch.equal(gi, ri), (rank, nprobe, k, "all")
            assert torch.equal(gd.view(torch.int32), rd.view(torch.int32)), (rank, nprobe, k, "all")
            oi, od = ShardedIndex(eng, dist, world, rank, result="owner").search(qd, nprobe, k)
            torch.cuda.synchronize()
            sl = slice(rank * per, (rank + 1) * per)
            assert torch.equal(oi, ri[sl]) and torch.equal(od.view(torch.int32), rd[sl].view(torch.int32)), (rank, nprobe, k, "owner")
        # cross-shard k-means on the product engine: both ranks end with the centroids of the single-process restatement
        rng = np.random.default_rng(83)
        cent = rng.standard_normal((32, d)).astype(np.float32)
        shards = [(cent[rng.integers(0, 32, 20000)] + 0.4 * rng.standard_normal((20000, d))).astype(np.float32) for _ in range(world)]
        c, a = sharded_kmeans(ctx, dist, torch.from_numpy(shards[rank]).cuda(), 64, metric, niter=3, seed=9, rank=rank, world=world)
        rc, ra = sharded_kmeans_reference(O, shards, 64, metric, niter=3, seed=9)
        assert (c.cpu().numpy().view(np.uint32) == rc.view(np.uint32)).all(), rank
        assert (a.cpu().numpy() == ra[rank]).all(), rank
        open(os.path.join(ret, "rank%d.ok" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_gpu_engine_world2_processes(metric):
    """two ranks, two processes, one GPU: GpuEngine + the real collectives (gloo) -- sharded search equals the unsharded
    qk_search bit for bit in both result layouts; sharded k-means equals its single-process restatement."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (each rank leaves a file: a multiprocessing.Manager is a FORK of this process, HIP runtime and all, and its server
    #  died now and then in long sessions)
    ret = tempfile.mkdtemp(prefix="qk_ranks_")
    mp.spawn(_world2_worker, args=(2, port, metric, ret), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(ret, "rank%d.ok" % r)) for r in range(2))


def test_sharded_kmeans_world1_equals_qk_kmeans():
    from quake_amd.capi import Context
    from quake_amd.sharded import sharded_kmeans
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(5)
    for metric, n, m in [("l2", 50000, 64), ("ip", 30000, 200)]:  # 50000 > 256*64: the subsample branch; 30000 < 256*200: all rows
        x = torch.randn(n, 48, generator=g, device="cuda")
        rc, ra, _ = ctx.kmeans(x, m, metric, niter=3, seed=11)
        c, a = sharded_kmeans(ctx, None, x.clone(), m, metric, niter=3, seed=11)
        torch.cuda.synchronize()
        assert torch.equal(c.view(torch.int32), rc.view(torch.int32)) and torch.equal(a, ra), metric
    ctx.close()


@pytest.mark.parametrize("G,per,k,metric", [(1, 7, 3, "l2"), (3, 5, 7, "l2"), (8, 64, 10, "l2"), (4, 33, 100, "ip"), (2, 1, 1, "ip")])
def test_packed_exchange_layout_and_merge(G, per, k, metric):
    """qk_pack_topk / qk_merge_topk_packed (the one-all-to-all exchange): the packed layout is the documented one (block j = ids
    then keys of queries [j*per, (j+1)*per), also with an odd number of entries: 16-byte padded blocks), the host restatement
    sharded.pack_topk_host writes the same bytes, and merging G packed blocks equals qk_merge_topk on [G][per][k] bit for bit
    (ties between ranks, -1 padded entries)."""
    from quake_amd.capi import Context
    from quake_amd.sharded import pack_topk_host, topk_block_bytes, unpack_topk_host
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(100 * G + per + k)
    # what one rank sends: its [G*per][k] local results
    ids = rng.integers(0, 1 << 40, size=(G * per, k)).astype(np.int64)
    keys = rng.integers(0, 50, size=(G * per, k)).astype(np.float32) * 0.25  # many ties
    ids[rng.random(ids.shape) < 0.1] = -1
    di, dk = torch.from_numpy(ids).cuda(), torch.from_numpy(keys).cuda()
    packed = ctx.pack_topk(di, dk, G)
    assert tuple(packed.shape) == (G, topk_block_bytes(per, k)) == (G, ctx.topk_block_bytes(per, k))
    host = pack_topk_host(torch.from_numpy(ids), torch.from_numpy(keys), G)
    n = per * k * 12
    assert torch.equal(packed.cpu()[:, :n], host[:, :n])
    ui, uk = unpack_topk_host(packed.cpu(), per, k)
    np.testing.assert_array_equal(ui.numpy().reshape(G * per, k), ids)
    np.testing.assert_array_equal(uk.numpy().reshape(G * per, k), keys)
    # what one rank receives: G blocks (one per source rank) for ITS per queries
    rid = rng.integers(0, 1 << 40, size=(G, per, k)).astype(np.int64)
    rk = np.sort(rng.integers(0, 30, size=(G, per, k)).astype(np.float32) * 0.5, axis=2)
    if metric == "ip":
        rk = -rk
    rid[:, :, k - 1:] = np.where(rng.random((G, per, 1)) < 0.3, -1, rid[:, :, k - 1:])
    recv = torch.empty((G, topk_block_bytes(per, k)), dtype=torch.uint8)
    for g in range(G):
        recv[g] = pack_topk_host(torch.from_numpy(rid[g]), torch.from_numpy(rk[g]), 1)[0]
    pi, pd = ctx.merge_topk_packed(recv.cuda(), per, k, metric)
    mi, md = ctx.merge_topk(torch.from_numpy(rid).cuda(), torch.from_numpy(rk).cuda(), metric)
    torch.cuda.synchronize()
    assert torch.equal(pi, mi) and torch.equal(pd.view(torch.int32), md.view(torch.int32))
    ctx.close()


@pytest.mark.parametrize("G,per,k,metric", [(2, 3, 961, "l2"), (5, 4, 2000, "ip"), (8, 2, 8192, "l2")])
def test_merge_of_sorted_runs_for_large_k(G, per, k, metric):
    """qk_merge_topk beyond k = 960 (k_merge_ranks_large): every rank's entries arrive sorted under (key, id) with the -1 padding
    behind them -- what qk_search returns --, the merged row is the first k of the union under the same order; ties between ranks,
    ranks with fewer than k entries, rows with fewer than k entries in total."""
    from quake_amd.capi import Context
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(7 * G + k)
    rid = np.full((G, per, k), -1, np.int64)
    rk = np.full((G, per, k), -np.inf if metric == "ip" else np.inf, np.float32)
    for q in range(per):
        ids = rng.permutation(G * k * 2)[:G * k].astype(np.int64)
        for g in range(G):
            nv = int(rng.integers(0, k + 1)) if q else k // (g + 2)  # row 0: few entries everywhere -> padding in the merged row
            kk = (rng.integers(0, 40, nv) * 0.25).astype(np.float32)
            ii = ids[g * k:g * k + nv]
            order = np.lexsort((ii, -kk if metric == "ip" else kk))
            rid[g, q, :nv], rk[g, q, :nv] = ii[order], kk[order]
    mi, md = ctx.merge_topk(torch.from_numpy(rid).cuda(), torch.from_numpy(rk).cuda(), metric)
    torch.cuda.synchronize()
    mi, md = mi.cpu().numpy(), md.cpu().numpy()
    for q in range(per):
        ii, kk = rid[:, q].reshape(-1), rk[:, q].reshape(-1)
        ok = ii >= 0
        ii, kk = ii[ok], kk[ok]
        order = np.lexsort((ii, -kk if metric == "ip" else kk))[:k]
        want_i = np.full(k, -1, np.int64)
        want_d = np.full(k, -np.inf if metric == "ip" else np.inf, np.float32)
        want_i[:order.size] = ii[order]
        want_d[:order.size] = kk[order] if metric == "ip" else np.sqrt(kk[order])
        np.testing.assert_array_equal(mi[q], want_i)
        np.testing.assert_array_equal(md[q].view(np.uint32), want_d.view(np.uint32))
    ctx.close()
