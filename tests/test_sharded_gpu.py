"""GpuEngine + ShardedIndex on one GPU (world = 1): the product engine behind the sharded orchestration -- search, and the
sharded add / remove with device tensors -- against the plain C-ABI calls on an identical store."""
import numpy as np
import pytest
import torch

from helpers import make_ivf, make_queries

pytestmark = pytest.mark.gpu


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
    ctx.set_squared_l2(False)
    ri, rd = ctx.search(parent, stores[1], qd, 4, 10, "l2")
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
    ctx.set_squared_l2(True)
    gi, gd = idx.search(qd, 24, 10)
    ctx.set_squared_l2(False)
    ri, rd = ctx.search(parent, stores[1], qd, 24, 10, "l2")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(gi.cpu().numpy(), ri.cpu().numpy())
    np.testing.assert_array_equal(gd.cpu().numpy(), rd.cpu().numpy())
    ctx.close()
