"""The compiled surface (quake_amd/_bindings.so: C++ host mirror + pybind11, the counterpart of quake._bindings) must
behave like the Python mirror: same results as the oracle, same error behaviour (test/cpp/quake_index.cpp:47-251)."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qb():
    from quake_amd.build_ext import build_bindings
    build_bindings()
    import quake_amd.bindings as b
    return b


def test_bindings_build_search_add_remove_save_load(qb, tmp_path):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3000, 32, generator=g)
    ids = torch.arange(3000)
    q = torch.randn(20, 32, generator=g)
    idx = qb.QuakeIndex()
    assert idx.ntotal() == 0 and idx.parent is None
    sp = qb.SearchParams()
    assert sp.k == 1 and sp.nprobe == 1 and sp.batched_scan is False
    with pytest.raises(RuntimeError):
        idx.search(q, sp)
    bp = qb.IndexBuildParams()
    assert bp.nlist == 0 and bp.niter == 5 and bp.metric == "l2"
    bp.nlist = 12
    info = idx.build(x, ids, bp)
    assert info.n_vectors == 3000 and info.d == 32 and idx.nlist() == 12 and idx.ntotal() == 3000 and idx.parent.ntotal() == 12
    # search parity with the oracle on the built partitions
    import ctypes as C
    pv, pi = [], []
    for p in range(12):
        sub = idx.get_ids()  # all ids, list order
        break
    assert sorted(sub.tolist()) == list(range(3000))
    sp.k, sp.nprobe = 10, 12
    r = idx.search(q, sp)
    gt = torch.topk(torch.cdist(q.double(), x.double()), 10, dim=1, largest=False)
    np.testing.assert_array_equal(r.ids.numpy(), gt.indices.numpy())  # nprobe = nlist: exact
    np.testing.assert_allclose(r.distances.numpy(), gt.values.numpy(), atol=1e-4)
    assert r.timing_info.n_queries == 20 and r.timing_info.parent_info is not None
    e = idx.search(torch.empty(0, 32), sp)
    assert e.ids.numel() == 0
    # add / remove / errors
    nx = torch.randn(5, 32, generator=g)
    nid = torch.arange(3000, 3005)
    m = idx.add(nx, nid)
    assert m.n_vectors == 5 and m.modify_count == 5 and idx.ntotal() == 3005
    np.testing.assert_array_equal(idx.get(nid).numpy(), nx.numpy())
    with pytest.raises(RuntimeError):
        idx.add(nx, nid)
    idx.remove(torch.arange(0, 50))
    assert idx.ntotal() == 2955
    with pytest.raises(RuntimeError):
        idx.remove(torch.tensor([7]))
    assert idx.maintenance().n_splits == 0
    idx.refine_partitions(torch.tensor([0, 1, 2]), 1)
    assert idx.ntotal() == 2955
    with pytest.raises(ValueError):
        bad = qb.IndexBuildParams()
        bad.metric = "cosine"
        qb.QuakeIndex().build(x, ids, bad)
    # save / load in the reference format, cross-checked with the Python mirror
    d = str(tmp_path / "idx")
    idx.save(d)
    l2 = qb.QuakeIndex()
    l2.load(d)
    assert l2.ntotal() == idx.ntotal() and l2.nlist() == idx.nlist()
    a, b = idx.search(q, sp), l2.search(q, sp)
    np.testing.assert_array_equal(a.ids.numpy(), b.ids.numpy())
    import quake_amd
    py = quake_amd.QuakeIndex()
    py.load(d)
    psp = quake_amd.SearchParams()
    psp.k, psp.nprobe = 10, 12
    c = py.search(q, psp)
    np.testing.assert_array_equal(a.ids.numpy(), c.ids.numpy())
    np.testing.assert_array_equal(a.distances.numpy(), c.distances.numpy())


def test_bindings_recall_target_search(qb):
    """SearchParams.recall_target through the compiled mirror == the ctypes path (same C ABI call underneath)."""
    import quake_amd
    g = torch.Generator().manual_seed(23)
    cent = torch.randn(80, 24, generator=g) * 2
    x = cent[torch.randint(0, 80, (20000,), generator=g)] + torch.randn(20000, 24, generator=g)
    q = x[:64] + 0.05 * torch.randn(64, 24, generator=g)
    ids = torch.arange(20000)
    res = []
    for mod in (qb, quake_amd):
        idx = mod.QuakeIndex()
        bp = mod.IndexBuildParams()
        bp.nlist = 80
        idx.build(x, ids, bp)
        sp = mod.SearchParams()
        sp.k = 5
        sp.recall_target = 0.9
        sp.initial_search_fraction = 0.25
        r = idx.search(q, sp)
        assert tuple(r.ids.shape) == (64, 5)
        res.append((r.ids.numpy().copy(), r.distances.numpy().copy(), r.timing_info.partitions_scanned))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][2] >= 2 * 64
