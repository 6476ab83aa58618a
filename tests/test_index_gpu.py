"""The reference's index-level tests restated against quake_amd (same API names as quake._bindings):
test/cpp/quake_index.cpp:47-251 (constructor/build/flat/search shapes/get/add/remove/ntotal/save-load),
test/cpp/query_coordinator.cpp:309-371 (k > partition size -> -1 / inf padding), :459-497,
test/cpp/search_recall_tests.cpp:160-254 (flat recall >= 0.99), and full-search parity with the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

DIM, NVEC, NLIST, NQ = 32, 2000, 10, 25


@pytest.fixture(scope="module")
def quake():
    import quake_amd
    return quake_amd


@pytest.fixture()
def data():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(NVEC, DIM, generator=g)
    ids = torch.arange(NVEC, dtype=torch.int64)
    q = torch.randn(NQ, DIM, generator=g)
    return x, ids, q


def build(quake, x, ids, nlist, metric="l2"):
    idx = quake.QuakeIndex()
    p = quake.IndexBuildParams()
    p.nlist = nlist
    p.metric = metric
    info = idx.build(x, ids, p)
    return idx, info


def test_constructor(quake):  # quake_index.cpp:47-55
    idx = quake.QuakeIndex()
    assert idx.parent is None and idx.build_params_ is None and idx.maintenance_policy_params_ is None
    assert idx.ntotal() == 0 and idx.nlist() == 0
    with pytest.raises(RuntimeError):
        sp = quake.SearchParams()
        idx.search(torch.zeros(1, 4), sp)


def test_build_partitioned_and_flat(quake, data):  # :58-100
    x, ids, q = data
    idx, info = build(quake, x, ids, NLIST)
    assert idx.parent is not None and idx.build_params_ is not None
    assert info.n_vectors == NVEC and info.d == DIM
    assert idx.ntotal() == NVEC and idx.nlist() == NLIST  # :214-229
    assert idx.parent.ntotal() == NLIST and idx.parent.nlist() == 1
    flat, _ = build(quake, x, ids, 0)
    assert flat.parent is None and flat.ntotal() == NVEC and flat.nlist() == 1
    with pytest.raises(ValueError):  # str_to_metric_type, common.h:154
        build(quake, x, ids, 0, metric="cosine")


@pytest.mark.parametrize("nlist", [NLIST, 0])
def test_search_shapes_and_empty(quake, data, nlist):  # :102-156
    x, ids, q = data
    idx, _ = build(quake, x, ids, nlist)
    sp = quake.SearchParams()
    sp.k, sp.nprobe = 5, 3
    r = idx.search(q, sp)
    assert tuple(r.ids.shape) == (NQ, 5) and tuple(r.distances.shape) == (NQ, 5)
    assert r.ids.dtype == torch.int64 and r.distances.dtype == torch.float32 and not r.ids.is_cuda
    assert r.timing_info.n_queries == NQ and r.timing_info.total_time_ns > 0
    e = idx.search(torch.empty(0, DIM), sp)  # query_coordinator.cpp:476-482
    assert e.ids.numel() == 0 and e.distances.numel() == 0


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_search_matches_oracle_on_the_built_index(quake, data, metric):
    """Whatever the k-means produced, search() over it must equal the oracle's search over the same partitions."""
    x, ids, q = data
    idx, _ = build(quake, x, ids, NLIST, metric)
    s = idx._store
    pv, pi = zip(*[s.get_list(p) for p in range(NLIST)])
    vecs, aids, offs = O.csr_from_partitions(pv, pi, DIM)
    cent, cids = idx.parent._store.get_list(0)
    sp = quake.SearchParams()
    for nprobe, k in [(1, 1), (3, 10), (NLIST, 50)]:
        sp.k, sp.nprobe = k, nprobe
        r = idx.search(q, sp)
        oi, od = O.search(q.numpy(), cent, vecs, aids, offs, nprobe, k, metric, batched_scan=True, centroid_ids=cids)
        np.testing.assert_array_equal(r.ids.numpy(), oi)
        np.testing.assert_array_equal(r.distances.numpy().view(np.uint32), od.view(np.uint32))


def test_get_add_remove(quake, data):  # :158-212
    x, ids, q = data
    idx, _ = build(quake, x, ids, NLIST)
    got = idx.get(torch.tensor([3, 17, 1999]))
    np.testing.assert_array_equal(got.numpy(), x[[3, 17, 1999]].numpy())
    g = torch.Generator().manual_seed(9)
    newx = torch.randn(10, DIM, generator=g)
    newids = torch.arange(NVEC, NVEC + 10)
    info = idx.add(newx, newids)
    assert info.n_vectors == 10 and info.modify_time_us >= 0 and idx.ntotal() == NVEC + 10
    np.testing.assert_array_equal(idx.get(newids[:2]).numpy(), newx[:2].numpy())
    # a freshly added vector is its own nearest neighbour
    sp = quake.SearchParams()
    sp.k, sp.nprobe = 1, NLIST
    r = idx.search(newx, sp)
    np.testing.assert_array_equal(r.ids.reshape(-1).numpy(), newids.numpy())
    with pytest.raises(RuntimeError):  # duplicate ids (partition_manager.cpp:168-172, 178-183)
        idx.add(newx, newids)
    with pytest.raises(RuntimeError):  # ids above INT32_MAX (:163)
        idx.add(newx[:1], torch.tensor([2 ** 31 + 5]))
    rem = torch.arange(0, 100)
    info = idx.remove(rem)
    assert info.n_vectors == 100 and idx.ntotal() == NVEC + 10 - 100
    r = idx.search(x[:100], sp)
    assert not np.isin(r.ids.numpy(), rem.numpy()).any()
    assert set(idx.get_ids().tolist()) == set(range(100, NVEC + 10))
    with pytest.raises(RuntimeError):  # removing a non-resident id (:283-296)
        idx.remove(torch.tensor([5]))
    m = idx.maintenance()  # window never fills through the public API (maintenance_policies.cpp:36-41)
    assert m.n_splits == 0 and m.n_deletes == 0


def test_k_greater_than_available_pads(quake):  # query_coordinator.cpp:309-371
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 8, generator=g)
    idx, _ = build(quake, x, torch.arange(4) + 100, 0)
    sp = quake.SearchParams()
    sp.k = 7
    r = idx.search(torch.randn(3, 8, generator=g), sp)
    assert (r.ids[:, :4] >= 100).all() and (r.ids[:, 4:] == -1).all()
    assert torch.isinf(r.distances[:, 4:]).all() and (r.distances[:, 4:] > 0).all()
    idx_ip, _ = build(quake, x, torch.arange(4), 0, "ip")
    r = idx_ip.search(torch.randn(3, 8, generator=g), sp)
    assert (r.ids[:, 4:] == -1).all() and (r.distances[:, 4:] < 0).all() and torch.isinf(r.distances[:, 4:]).all()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_flat_recall(quake, metric):  # search_recall_tests.cpp:160-189,225-254: 100k x 32, recall@10 >= 0.99
    g = torch.Generator().manual_seed(11)
    x = torch.randn(100000, 32, generator=g)
    q = torch.randn(100, 32, generator=g)
    idx, _ = build(quake, x, torch.arange(100000), 0, metric)
    sp = quake.SearchParams()
    sp.k = 10
    r = idx.search(q, sp)
    d = torch.cdist(q.double(), x.double()) if metric == "l2" else -(q.double() @ x.double().T)
    gt = torch.topk(d, 10, dim=1, largest=False).indices
    rec = quake.compute_recall(r.ids, gt, 10).mean().item()
    assert rec >= 0.99


def test_ivf_recall_increases_with_nprobe(quake):
    g = torch.Generator().manual_seed(13)
    cent = torch.randn(64, 32, generator=g) * 3
    x = cent[torch.randint(0, 64, (50000,), generator=g)] + torch.randn(50000, 32, generator=g)
    q = x[torch.randperm(50000, generator=g)[:200]] + 0.1 * torch.randn(200, 32, generator=g)
    idx, _ = build(quake, x, torch.arange(50000), 64)
    gt = torch.topk(torch.cdist(q.double(), x.double()), 10, dim=1, largest=False).indices
    sp = quake.SearchParams()
    sp.k = 10
    recs = []
    for nprobe in (1, 4, 64):
        sp.nprobe = nprobe
        recs.append(quake.compute_recall(idx.search(q, sp).ids, gt, 10).mean().item())
    assert recs[0] <= recs[1] + 1e-6 <= recs[2] + 2e-6 and recs[2] >= 0.999


def test_save_load_roundtrip(quake, data, tmp_path):  # quake_index.cpp:232-251
    x, ids, q = data
    idx, _ = build(quake, x, ids, NLIST)
    d = str(tmp_path / "idx")
    idx.save(d)
    # byte layout of the reference format (dynamic_inverted_list.cpp:338-419)
    blob = open(d + "/partitions", "rb").read()
    magic, version, nl, code_size, nparts = np.frombuffer(blob, "<u4", 2)[0], np.frombuffer(blob, "<u4", 2)[1], \
        *np.frombuffer(blob, "<u8", 3, 8)
    assert magic == 0x44494E4C and version == 3 and nl == NLIST and code_size == DIM * 4 and nparts == NLIST
    assert len(blob) == 32 + 8 * (NLIST + 1) + 8 * NLIST + NVEC * (DIM * 4 + 8)
    meta = open(d + "/metadata.txt").read()
    assert "metric=1" in meta and "nlist=%d" % NLIST in meta and "ntotal=%d" % NVEC in meta
    loaded = quake.QuakeIndex()
    loaded.load(d)
    assert loaded.ntotal() == idx.ntotal() and loaded.nlist() == idx.nlist() and loaded.parent is not None
    sp = quake.SearchParams()
    sp.k, sp.nprobe = 10, 4
    a, b = idx.search(q, sp), loaded.search(q, sp)
    np.testing.assert_array_equal(a.ids.numpy(), b.ids.numpy())
    np.testing.assert_array_equal(a.distances.numpy(), b.distances.numpy())


def test_large_dimension(quake):  # quake_index.cpp d = 1024 stress case
    g = torch.Generator().manual_seed(17)
    x = torch.randn(3000, 1024, generator=g)
    q = torch.randn(8, 1024, generator=g)
    idx, _ = build(quake, x, torch.arange(3000), 8)
    sp = quake.SearchParams()
    sp.k, sp.nprobe = 10, 8
    r = idx.search(q, sp)
    gt = torch.topk(torch.cdist(q.double(), x.double()), 10, dim=1, largest=False)
    np.testing.assert_array_equal(r.ids.numpy(), gt.indices.numpy())
    np.testing.assert_allclose(r.distances.numpy(), gt.values.numpy(), atol=1e-3)


def test_refine_partitions_keeps_everything_searchable(quake, data):  # partition_manager.cpp:121-167 spirit
    x, ids, q = data
    idx, _ = build(quake, x, ids, NLIST)
    before = idx.ntotal()
    idx.refine_partitions(torch.tensor([1, 3, 5]), iterations=2)
    assert idx.ntotal() == before and idx.nlist() == NLIST and idx.parent.ntotal() == NLIST
    assert set(idx.get_ids().tolist()) == set(range(NVEC))
    sp = quake.SearchParams()
    sp.k, sp.nprobe = 1, NLIST
    r = idx.search(x[:200], sp)  # every vector still finds itself
    np.testing.assert_array_equal(r.ids.reshape(-1).numpy(), np.arange(200))


def test_recall_vs_recall_target(quake):  # search_recall_tests.cpp:284-309 RecallVsRecallTargetL2 (it only prints; bounds added)
    g = torch.Generator().manual_seed(17)
    cent = torch.randn(100, 32, generator=g) * 2
    x = cent[torch.randint(0, 100, (40000,), generator=g)] + torch.randn(40000, 32, generator=g)
    q = x[torch.randperm(40000, generator=g)[:150]] + 0.1 * torch.randn(150, 32, generator=g)
    idx, _ = build(quake, x, torch.arange(40000), 100)
    gt = torch.topk(torch.cdist(q.double(), x.double()), 10, dim=1, largest=False).indices
    sp = quake.SearchParams()
    sp.k = 10
    sp.recompute_threshold = 0.0
    sp.initial_search_fraction = 0.5
    prev_scanned, recs = 0, []
    for rt in (0.5, 0.7, 0.9, 0.99, 1.0):
        sp.recall_target = rt
        r = idx.search(q, sp)
        assert r.ids.shape == (150, 10) and r.distances.shape == (150, 10)
        assert r.timing_info.partitions_scanned >= prev_scanned  # a higher target never scans fewer partitions
        prev_scanned = r.timing_info.partitions_scanned
        recs.append(quake.compute_recall(r.ids, gt, 10).mean().item())
    assert recs[-1] >= recs[0] and recs[2] >= 0.9 and recs[-1] >= 0.99
    # same walk as the oracle's restatement, through the index object (device tensors in, device tensors out)
    sp.recall_target = 0.9
    r = idx.search(q.cuda(), sp)
    assert r.ids.is_cuda
    cids = idx.parent.get_ids()
    cvec = idx.parent.get(cids)
    pv, pi = [], []
    for p in cids.tolist():
        sub = idx._store.get_list(p)
        pv.append(sub[0])
        pi.append(sub[1])
    order = np.argsort(cids.numpy())
    vecs, ids_, offsets = O.csr_from_partitions([pv[i] for i in order], [pi[i] for i in order], 32)
    oi, od, on = O.search_aps(q.numpy(), cvec.numpy()[order], vecs, ids_, offsets, 10, "l2", 0.9, recompute_threshold=0.0,
                              initial_search_fraction=0.5, centroid_ids=cids.numpy()[order], num_threads=8)
    np.testing.assert_array_equal(r.ids.cpu().numpy(), oi)
    np.testing.assert_array_equal(r.distances.cpu().numpy().view(np.uint32), od.view(np.uint32))
    assert r.timing_info.partitions_scanned == int(on.sum())
    # batched_scan = true ignores the target and uses nprobe (query_coordinator.cpp:637-641,659-673)
    sp.batched_scan = True
    sp.nprobe = 3
    rb = idx.search(q, sp)
    sp.recall_target = -1.0
    rn = idx.search(q, sp)
    np.testing.assert_array_equal(rb.ids.numpy(), rn.ids.numpy())


def test_repeated_build_search(quake):  # quake_index.cpp:322-362 RepeatedBuildSearchTest
    g = torch.Generator().manual_seed(41)
    x = torch.randn(10000, 32, generator=g)
    ids = torch.arange(1000, 11000)
    q = torch.randn(100, 32, generator=g)
    first = None
    for _ in range(5):
        idx = quake.QuakeIndex()
        p = quake.IndexBuildParams()
        p.nlist, p.metric, p.niter = 16, "l2", 3
        idx.build(x, ids, p)
        sp = quake.SearchParams()
        sp.k, sp.nprobe = 10, 4
        r = idx.search(q, sp)
        assert tuple(r.ids.shape) == (100, 10)
        if first is None:
            first = r.ids.clone()
        else:
            assert torch.equal(first, r.ids)  # same data, same seed: the build is deterministic
        del idx


def test_rapid_add_remove_add(quake):  # quake_index.cpp:400-443 RapidAddRemoveAddTest
    g = torch.Generator().manual_seed(42)
    idx = quake.QuakeIndex()
    p = quake.IndexBuildParams()
    p.nlist = 2
    idx.build(torch.randn(1000, 16, generator=g), torch.arange(1000), p)
    for i in range(1, 10):
        av = torch.randn(1000, 16, generator=g)
        ai = torch.arange(i * 1000, (i + 1) * 1000)
        assert idx.add(av, ai).n_vectors == 1000
        assert idx.remove(ai[:500]).n_vectors == 500
        assert idx.add(av[:500], ai[:500]).n_vectors == 500
        sp = quake.SearchParams()
        sp.k = 2
        r = idx.search(torch.randn(5, 16, generator=g), sp)
        assert tuple(r.ids.shape) == (5, 2)
    assert idx.ntotal() == 10000
    # every vector is where its id says (exact self-search over both partitions)
    sp = quake.SearchParams()
    sp.k, sp.nprobe = 1, 2
    r = idx.search(av[:50], sp)
    assert torch.equal(r.ids.reshape(-1), ai[:50])


def test_search_add_remove_maintenance_loop(quake):  # quake_index.cpp:482-529 SearchAddRemoveMaintenanceTest
    g = torch.Generator().manual_seed(43)
    n = 100000
    idx = quake.QuakeIndex()
    p = quake.IndexBuildParams()
    p.nlist, p.metric, p.niter = 100, "l2", 3
    idx.build(torch.randn(n, 16, generator=g), torch.arange(n), p)
    for i in range(30):
        q = torch.randn(100, 16, generator=g) * 0.1
        sp = quake.SearchParams()
        sp.nprobe, sp.k = 1, 5
        r = idx.search(q, sp)
        assert tuple(r.ids.shape) == (100, 5)
        ai = torch.arange(i * 10 + n, i * 10 + n + 10)
        assert idx.add(torch.randn(10, 16, generator=g), ai).n_vectors == 10
        assert idx.remove(ai[:5]).n_vectors == 5
        t = idx.maintenance()  # hits are not tracked by default: the window never fills (maintenance_policies.cpp:36-41)
        assert t.n_splits == 0 and t.n_deletes == 0
    assert idx.ntotal() == n + 30 * 5


def test_load_streams_the_file(quake, tmp_path):
    """load() reads the partitions file in windows of whole partitions (dynamic_inverted_list.cpp:421-520 reads partition by
    partition): a 1.3 GB index loads with a fraction of its size in extra host memory, list contents and row order intact --
    odd dimension (ids unaligned in the file), empty lists, lists of one row, several windows"""
    import resource
    d, nlist, n = 129, 300, 2_400_000
    g = torch.Generator().manual_seed(23)
    cent = torch.randn(nlist, d, generator=g)
    assign = torch.randint(0, nlist - 3, (n,), generator=g)  # the last three lists stay empty
    assign[:2] = torch.tensor([nlist - 4, nlist - 4])
    x = cent[assign] + 0.2 * torch.randn(n, d, generator=g)
    ids = torch.randperm(n, generator=g) + 5
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nlist).numpy().astype(np.int64)
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(counts)
    from quake_amd.index import QuakeIndex
    idx = QuakeIndex.from_partitions(cent.numpy(), offsets, ids[order].numpy(), x[order].numpy(), "l2")
    path = str(tmp_path / "big")
    idx.save(path)
    assert os.path.getsize(path + "/partitions") > 1.2e9
    del x, order
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss  # KB, monotone
    loaded = quake.QuakeIndex()
    loaded.load(path)
    grown = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) / 1e6
    assert grown < 0.9, f"load() grew the peak host memory by {grown:.2f} GB for a 1.3 GB file"
    assert loaded.ntotal() == n and loaded.nlist() == nlist
    for p in (0, 7, nlist - 4, nlist - 1):
        va, ia = idx._store.get_list(p)
        vb, ib = loaded._store.get_list(p)
        np.testing.assert_array_equal(ia, ib)
        np.testing.assert_array_equal(va, vb)
    q = cent[:64] + 0.1
    sp = quake.SearchParams()
    sp.k, sp.nprobe = 10, 8
    a, b = idx.search(q, sp), loaded.search(q, sp)
    np.testing.assert_array_equal(a.ids.numpy(), b.ids.numpy())
    np.testing.assert_array_equal(a.distances.numpy(), b.distances.numpy())


@pytest.mark.parametrize("mirror", ["python", "compiled"])
@pytest.mark.parametrize("workers", [0, 2])
def test_load_of_a_hand_assembled_directory(mirror, workers):
    """An index directory assembled byte by byte from the reference's format description (tests/golden/make_disk_image.py: struct +
    numpy, nothing of this repo) -- partition ids in non-ascending file order, one empty partition, row order != id order -- is read
    by both loaders: same partitions, same rows in the same order, and it answers searches exactly (dynamic_inverted_list.cpp:338-520,
    quake_index.cpp:207-267)."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    exp = json.load(open(os.path.join(here, "golden", "disk_image.json")))
    if mirror == "python":
        import quake_amd as mod
    else:
        from quake_amd.build_ext import build_bindings
        build_bindings()
        import quake_amd.bindings as mod
    idx = mod.QuakeIndex()
    idx.load(os.path.join(here, "golden", "disk_image"), workers)
    assert idx.ntotal() == exp["ntotal"] and idx.nlist() == exp["nlist"] and idx.parent.ntotal() == 3
    allv, alli = [], []
    for pid, part in exp["partitions"].items():
        ids = torch.tensor(part["ids"], dtype=torch.int64)
        v = torch.tensor(part["vectors"], dtype=torch.float32).reshape(-1, exp["d"])
        if ids.numel():
            assert torch.equal(idx.get(ids), v)  # every row under its id
            allv.append(v)
            alli.append(ids)
    assert torch.equal(idx.parent.get(torch.tensor([9, 5, 0])), torch.tensor([exp["centroids"][k] for k in ("9", "5", "0")]))
    if mirror == "python":  # row order inside a partition = file order
        for pid, part in exp["partitions"].items():
            assert idx._store.get_list_ids(int(pid)).tolist() == part["ids"]
    x, ids = torch.cat(allv), torch.cat(alli)
    sp = mod.SearchParams()
    sp.k, sp.nprobe = 3, 3
    q = x[::2] + 0.25
    r = idx.search(q, sp)
    dist = torch.cdist(q.double(), x.double())
    key = dist * 1e6 + ids.double()[None, :] * 1e-3  # (distance, id) order: integer data, exact distances
    want = ids[torch.topk(key, 3, dim=1, largest=False).indices]
    assert torch.equal(r.ids.cpu(), want)
    np.testing.assert_allclose(r.distances.cpu().numpy(), torch.topk(dist, 3, dim=1, largest=False).values.numpy(), atol=1e-5)
