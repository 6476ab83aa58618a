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
    # the PQ fields of the bound build result (wrap.cpp:332-335) and its one-line summary (wrap.cpp:338-350): a caller printing it
    assert info.code_size == -1 and info.n_codebooks == -1
    import json
    assert json.loads(repr(info))["n_vectors"] == 3000 and '"n_codebooks": -1' in repr(info)
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
    assert json.loads(repr(r.timing_info))["n_queries"] == 20 and "parent_scan_time_ns" in repr(r.timing_info)
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


def test_bindings_internal_seams(qb):
    """The collaborators the reference's own tests reach into (test/cpp/quake_index.cpp:50-54, query_coordinator.cpp:42-254):
    partition_manager / query_coordinator members, the three scan entry points (one device pipeline: identical results, like the
    reference's worker-vs-serial equality test :201-254), padding when k exceeds what the scanned partitions hold (:309-371) and
    with zero partitions (:459-497), and the list_scanning seam (batched_scan_list on a raw list, list_scanning.cpp:432-562)."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(4000, 16, generator=g)
    ids = torch.arange(4000) + 100
    q = torch.randn(12, 16, generator=g)
    idx = qb.QuakeIndex()
    assert idx.partition_manager is None and idx.query_coordinator is None  # ConstructorTest
    bp = qb.IndexBuildParams()
    bp.nlist = 8
    idx.build(x, ids, bp)
    pm, qc = idx.partition_manager, idx.query_coordinator
    assert pm.ntotal() == 4000 and pm.nlist() == 8 and pm.d() == 16 and pm.validate()
    pids = pm.get_partition_ids()
    sizes = pm.get_partition_sizes(pids)
    assert pids.tolist() == list(range(8)) and int(sizes.sum()) == 4000
    assert sorted(pm.get_ids().tolist()) == ids.tolist()
    sp = qb.SearchParams()
    sp.k, sp.nprobe = 5, 3
    a = idx.search(q, sp)
    b = qc.search(q, sp)
    np.testing.assert_array_equal(a.ids.numpy(), b.ids.numpy())
    # scan_partitions with an explicit [Q, P] list == the three scan variants; -1 entries are skipped
    probe = torch.tensor([[0, 3, -1]] * 12)
    r0 = qc.scan_partitions(q, probe, sp)
    for fn in (qc.serial_scan, qc.batched_serial_scan, qc.worker_scan):
        r = fn(q, probe, sp)
        np.testing.assert_array_equal(r.ids.numpy(), r0.ids.numpy())
        np.testing.assert_array_equal(r.distances.numpy(), r0.distances.numpy())
    # 1-D partition list = the same set for every query; k larger than the partitions hold -> -1 / inf padding
    small = int(sizes[5])
    sp2 = qb.SearchParams()
    sp2.k = small + 3
    r = qc.scan_partitions(q, torch.tensor([5]), sp2)
    assert (r.ids[:, :small] >= 0).all() and (r.ids[:, small:] == -1).all() and torch.isinf(r.distances[:, small:]).all()
    r = qc.scan_partitions(q, torch.empty((12, 0), dtype=torch.int64), sp)  # zero partitions
    assert (r.ids == -1).all() and torch.isinf(r.distances).all()
    assert qc.search(torch.empty(0, 16), sp).ids.numel() == 0
    # list_scanning seam: a raw list against torch brute force (ids equal, distances within 1e-4)
    lv = torch.randn(1000, 16, generator=g)
    lid = torch.arange(1000) * 3
    for metric in ("l2", "ip"):
        gi, gd = qb.batched_scan_list(q, lv, lid, 10, metric)
        if metric == "l2":
            t = torch.topk(torch.cdist(q.double(), lv.double()), 10, dim=1, largest=False)
        else:
            t = torch.topk(q.double() @ lv.double().T, 10, dim=1, largest=True)
        np.testing.assert_array_equal(gi.numpy(), lid[t.indices].numpy())
        np.testing.assert_allclose(gd.numpy(), t.values.numpy(), atol=1e-4)
    gi, gd = qb.batched_scan_list(q, lv[:4], torch.empty(0, dtype=torch.int64), 10, "l2")  # no ids: row numbers; fewer than k
    assert sorted(gi[0, :4].tolist()) == [0, 1, 2, 3] and (gi[:, 4:] == -1).all()


def test_bindings_device_tensors_follow_torch_streams(qb):
    """CUDA tensors in, CUDA tensors out, on whatever stream torch is on: the mirror binds the library to torch's current stream for
    the call (no device-wide synchronisation), so queries produced by a kernel enqueued just before the call -- on the default
    stream or on a side stream -- are seen, and the answers equal those for the same queries as CPU tensors (search: the reference's
    entry point query_coordinator.cpp:612-657; scan_partitions :659-673 keeps device queries and list numbers on the device)."""
    g = torch.Generator().manual_seed(51)
    x = torch.randn(20000, 32, generator=g)
    ids = torch.arange(20000) + 7
    idx = qb.QuakeIndex()
    bp = qb.IndexBuildParams()
    bp.nlist = 40
    idx.build(x, ids, bp)
    qc = idx.query_coordinator
    sp = qb.SearchParams()
    sp.k, sp.nprobe = 10, 6
    base = torch.randn(300, 32, generator=g)
    side = torch.cuda.Stream()
    for it in range(4):
        qh = base * (1.0 + 0.25 * it) + 0.01 * it
        want = idx.search(qh, sp)
        assert not want.ids.is_cuda
        bd = base.cuda()
        torch.cuda.synchronize()
        if it % 2 == 0:
            qd = bd * (1.0 + 0.25 * it) + 0.01 * it          # produced on the current stream right before the call
            got = idx.search(qd, sp)
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                qd = bd * (1.0 + 0.25 * it) + 0.01 * it      # ... or on a side stream, which the call then runs on
                got = idx.search(qd, sp)
                gi, gdist = got.ids.cpu(), got.distances.cpu()
            side.synchronize()
        assert got.ids.is_cuda and got.distances.is_cuda
        np.testing.assert_array_equal(got.ids.cpu().numpy(), want.ids.numpy())
        np.testing.assert_array_equal(got.distances.cpu().numpy().view(np.uint32), want.distances.numpy().view(np.uint32))
        probe = torch.tensor([[1, 7, 30, -1]] * 300)
        rh = qc.scan_partitions(qh, probe, sp)
        rd = qc.scan_partitions(qd, probe.cuda(), sp)
        assert rd.ids.is_cuda and not rh.ids.is_cuda
        np.testing.assert_array_equal(rd.ids.cpu().numpy(), rh.ids.numpy())
        np.testing.assert_array_equal(rd.distances.cpu().numpy().view(np.uint32), rh.distances.numpy().view(np.uint32))
        r1 = qc.scan_partitions(qd, torch.tensor([3, 9]), sp)  # host list numbers with device queries: moved over
        r2 = qc.scan_partitions(qh, torch.tensor([3, 9]), sp)
        np.testing.assert_array_equal(r1.ids.cpu().numpy(), r2.ids.numpy())


def test_bindings_compiled_maintenance(qb):
    """maintenance() in the compiled mirror runs the policy (hit window -> split hot / delete cold partitions -> refine): after it,
    every vector is still resident exactly once and exhaustive search stays exact."""
    g = torch.Generator().manual_seed(41)
    cent = torch.randn(12, 16, generator=g) * 4
    x = cent[torch.randint(0, 12, (6000,), generator=g)] + torch.randn(6000, 16, generator=g)
    ids = torch.arange(6000)
    idx = qb.QuakeIndex()
    bp = qb.IndexBuildParams()
    bp.nlist = 12
    idx.build(x, ids, bp)
    assert idx.maintenance().n_splits == 0  # window not full: nothing happens (maintenance_policies.cpp:36-41)
    mp = qb.MaintenancePolicyParams()
    mp.window_size = 64
    mp.refinement_radius = 4
    mp.refinement_iterations = 1
    mp.split_threshold_ns = 0.0
    mp.delete_threshold_ns = 0.0
    idx.initialize_maintenance_policy(mp)
    idx.set_track_hits(True)
    sp = qb.SearchParams()
    sp.k, sp.nprobe = 5, 2
    hot = x[:128] + 0.01 * torch.randn(128, 16, generator=g)
    for i in range(0, 128, 32):
        idx.search(hot[i:i + 32], sp)
    info = idx.maintenance()
    assert info.n_splits + info.n_deletes >= 0 and info.total_time_us >= 0
    assert idx.ntotal() == 6000 and sorted(idx.get_ids().tolist()) == ids.tolist()
    assert idx.partition_manager.validate() and idx.parent.ntotal() == idx.nlist()
    sp.nprobe = idx.nlist()
    r = idx.search(x[:50], sp)
    assert (r.ids[:, 0] == ids[:50]).all()  # every vector finds itself
