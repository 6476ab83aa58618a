"""Maintenance on the device store: split / delete / local refinement keep every vector searchable, and the dynamic
workload harness (generator + evaluator) behaves like the reference's (test/python/test_workload_generator.py:69-114;
test/cpp/maintenance.cpp spirit: after maintenance the index answers exactly and its bookkeeping is consistent)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def linear_model_estimator(d, alpha=0.9):
    from quake_amd.maintenance import ListScanLatencyEstimator, MaintenanceCostEstimator
    lat = ListScanLatencyEstimator(d, [1, 2, 4, 16, 64, 256, 1024, 4096, 16384, 65536], [1, 4, 16, 64, 256], 1,
                                   profile_fn=lambda n, k: 200.0 + 10.0 * n)  # ns: fixed cost + per-row cost
    return MaintenanceCostEstimator(d, alpha, 10, latency_estimator=lat)


def check_consistent(idx, x, ids_all, q):
    """every resident vector is in exactly one partition, centroids and partitions agree, exhaustive search is exact"""
    import quake_amd as quake
    got = idx.get_ids()
    assert sorted(got.tolist()) == sorted(ids_all.tolist())
    assert idx.parent.ntotal() == idx.nlist()
    assert sorted(idx.parent.get_ids().tolist()) == sorted(int(v) for v in idx._store.list_ids())
    sp = quake.SearchParams()
    sp.k = 10
    sp.nprobe = idx.nlist()
    r = idx.search(q, sp)
    xs = x[ids_all]
    gt = torch.topk(torch.cdist(q.double(), xs.double()), 10, dim=1, largest=False)
    # (fp64 ground truth vs the fp32 expanded form: near-equal distances may swap places, so compare as sets)
    want = ids_all[gt.indices].numpy()
    for a, b in zip(r.ids.numpy(), want):
        assert len(set(a.tolist()) & set(b.tolist())) >= 9
    np.testing.assert_allclose(r.distances.numpy(), gt.values.float().numpy(), atol=1e-3)


def test_policy_splits_hot_and_deletes_cold_partitions():
    import quake_amd as quake
    from quake_amd.maintenance import ListScanLatencyEstimator, MaintenanceCostEstimator
    g = torch.Generator().manual_seed(5)
    d = 16
    # background: 60000 vectors in 40 k-means partitions (~1500 each, nobody queries them).  The hot partitions below are
    # a little below the average size on purpose: a partition much larger than the average looks deletable to compute_delete_delta and
    # never reaches the split test (maintenance_policies.cpp:74-126 tests split only in the else branch)
    NB = 60000
    x_bg = torch.randn(NB, d, generator=g)
    idx = quake.QuakeIndex()
    bp = quake.IndexBuildParams()
    bp.nlist = 40
    idx.build(x_bg, torch.arange(NB), bp)
    # two large HOT partitions and six tiny COLD ones, far from everything else (added as whole partitions)
    hot_c = torch.stack([torch.full((d,), 8.0), torch.full((d,), -8.0)])
    cold_c = torch.stack([torch.cat([torch.full((1,), 60.0 + 10 * i), torch.zeros(d - 1)]) for i in range(6)])
    vecs, vids, nxt = [], [], NB
    for c, n in [(hot_c[0], 1200), (hot_c[1], 1200)] + [(c, 10) for c in cold_c]:
        vecs.append((c + 0.3 * torch.randn(n, d, generator=g)).numpy())
        vids.append(np.arange(nxt, nxt + n, dtype=np.int64))
        nxt += n
    cents = np.stack([v.mean(0) for v in vecs]).astype(np.float32)
    new_pids = idx._add_partitions({"centroids": cents, "vectors": vecs, "vector_ids": vids})
    idx._resident.update(range(NB, nxt))
    assert new_pids == list(range(40, 48)) and idx.nlist() == 48 and idx.ntotal() == nxt
    x = torch.cat([x_bg] + [torch.from_numpy(v) for v in vecs])
    ids = torch.arange(nxt)
    # cost model: 100 ns per partition + 1 ns per row -> splitting pays above ~800 rows, deleting a 10-row partition pays
    lat = ListScanLatencyEstimator(d, [1, 2, 4, 16, 64, 256, 1024, 4096, 16384, 65536], [1, 4, 16, 64, 256], 1,
                                   profile_fn=lambda n, k: 100.0 + 1.0 * n)
    mp = quake.MaintenancePolicyParams()
    mp.window_size = 200
    mp.refinement_radius = 4
    mp.refinement_iterations = 2
    mp.min_partition_size = 32
    mp.delete_threshold_ns = 0.1
    mp.split_threshold_ns = 0.1
    idx.initialize_maintenance_policy(mp, cost_estimator=MaintenanceCostEstimator(d, 0.9, 10, latency_estimator=lat))
    # window not full -> nothing happens (maintenance_policies.cpp:36-41)
    t = idx.maintenance()
    assert t.n_splits == 0 and t.n_deletes == 0 and idx.nlist() == 48
    idx.track_hits = True
    sp = quake.SearchParams()
    sp.k = 10
    sp.nprobe = 1
    q = torch.cat([hot_c[0] + 0.3 * torch.randn(150, d, generator=g), hot_c[1] + 0.3 * torch.randn(100, d, generator=g)])
    idx.search(q, sp)
    pol = idx._policy()
    assert pol.hit_count_tracker_.get_num_queries_recorded() == 200  # 250 recorded, the window keeps 200
    assert pol.hit_count_tracker_.get_current_scan_fraction() == pytest.approx(1200 / nxt, rel=1e-3)
    t = idx.maintenance()
    assert t.n_splits == 2 and t.n_deletes == 6, (t.n_splits, t.n_deletes)
    assert idx.nlist() == 48 + 2 - 6  # split: -1 +2, delete: -1
    live = sorted(int(v) for v in idx._store.list_ids())
    assert all(p not in live for p in range(40, 48)) and live[-4:] == [48, 49, 50, 51]  # curr_partition_id_ keeps counting
    assert idx.ntotal() == nxt
    check_consistent(idx, x, ids, q[:20])
    # the index stays dynamic afterwards
    idx.remove(ids[:100])
    nx = torch.randn(50, d, generator=g)
    idx.add(nx, torch.arange(nxt, nxt + 50))
    check_consistent(idx, torch.cat([x, nx]), torch.cat([ids[100:], torch.arange(nxt, nxt + 50)]), q[:10])


def test_device_profiled_cost_model():
    """device_profile_fn drives the scans whose time the default cost model reads; here the CLOCK is injected (45 ns per pair +
    0.085 ns per row, the shape of a recorded MI355X grid), so the grid and everything derived from it are values, not timings."""
    from quake_amd.maintenance import ListScanLatencyEstimator, MaintenanceCostEstimator, device_profile_fn
    seen = []

    def elapsed(n, k, npart):
        seen.append((n, k, npart))
        return npart * (45.0 + 0.085 * n + 0.01 * k) * 1e-9

    fn = device_profile_fn(32, 3, elapsed=elapsed)
    lat = ListScanLatencyEstimator(32, [64, 1024, 16384], [1, 16], 3, profile_fn=fn)
    assert sorted(set(seen)) == [(n, k, min(1024, max(16, (1 << 22) // n))) for n in (64, 1024, 16384) for k in (1, 16)]
    for n in (64, 1024, 16384):
        for k in (1, 16):
            assert lat.estimate_scan_latency(n, k) == pytest.approx(45.0 + 0.085 * n + 0.01 * k, rel=1e-9)
    assert lat.estimate_scan_latency(8192, 10) == pytest.approx(45.0 + 0.085 * 8192 + 0.1, rel=1e-6)
    est = MaintenanceCostEstimator(32, 0.9, 10, latency_estimator=lat)
    assert np.isfinite(est.compute_split_delta(8000, 0.5, 100))


def test_workload_generation_and_evaluation(tmp_path):  # cf. test/python/test_workload_generator.py:28-114
    """runbook schema (the on-disk contract shared with the reference's harness), determinism of the seeded stream, and a
    replay with maintenance on: the index follows the runbook's resident set, exhaustive probing finds everything."""
    from quake_amd.workload import ClusterWalk, WorkloadSpec, generate_workload, replay_workload
    import quake_amd as quake
    torch.manual_seed(0)
    base = torch.randn(1000, 16)
    queries = torch.randn(100, 16)
    wdir = tmp_path / "workload"
    spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.2, query_ratio=0.5, update_batch_size=20, query_batch_size=10,
                        number_of_operations=10, initial_size=200, cluster_size=50, seed=1738)
    rb = generate_workload(wdir, base, spec, queries=queries)
    assert (wdir / "runbook.json").exists() and (wdir / "operations").exists()
    assert rb == json.load(open(wdir / "runbook.json"), object_pairs_hook=lambda kv: {(int(k) if k.isdigit() else k): v for k, v in kv})
    assert set(rb) == {"parameters", "initialize", "operations", "summary"}
    with open(os.path.join(os.path.dirname(__file__), "golden", "runbook_schema.json")) as f:
        schema = json.load(f)
    assert sorted(rb["parameters"]) == sorted(schema["parameters"])  # the reference's parameter block, key for key
    assert len(rb["operations"]) <= 10
    for i, op in rb["operations"].items():
        assert set(op) <= set(schema["operation"]) and {"type", "sample_size", "n_resident"} <= set(op)
        assert op["type"] in ("insert", "delete", "query") and op["sample_size"] > 0 and op["n_resident"] > 0
        assert (wdir / "operations" / f"{i}.pt").exists()
        if op["type"] == "query":
            gt = torch.load(wdir / "operations" / f"{i}_gt_ids.pt", weights_only=True)
            assert gt.shape == (op["sample_size"], 100) and "gt_time" in op
    s = rb["summary"]
    assert sorted(s) == sorted(schema["summary"])
    assert s["n_inserts"] + s["n_deletes"] + s["n_queries"] == s["n_operations"] == len(rb["operations"])
    # same seed -> same operation stream and the same ids
    rb2 = generate_workload(tmp_path / "w2", base, spec, queries=queries)
    assert [o["type"] for o in rb2["operations"].values()] == [o["type"] for o in rb["operations"].values()]
    for i in rb["operations"]:
        assert torch.equal(torch.load(wdir / "operations" / f"{i}.pt", weights_only=True),
                           torch.load(tmp_path / "w2" / "operations" / f"{i}.pt", weights_only=True))
    with pytest.raises(ValueError):
        WorkloadSpec(insert_ratio=0.5, delete_ratio=0.5, query_ratio=0.5).check()

    sp = quake.SearchParams()
    sp.k, sp.nprobe = 5, 10
    mp = quake.MaintenancePolicyParams()
    mp.window_size = 20
    res = replay_workload(wdir, wdir, "quake_test", nlist=10, search_params=sp, maintenance_params=mp)
    assert isinstance(res, list) and len(res) == len(rb["operations"])
    assert json.load(open(wdir / "quake_test_results.json")) == res
    for r in res:
        assert r["latency_ms"] >= 0 and r["n_total"] == r["n_resident"]  # the index tracks the runbook's resident set
        if r["operation_type"] == "query":
            assert r["recall"] >= 0.99  # nprobe = nlist: exhaustive

    # skewed sampling: consecutive draws come from neighbouring clusters and empty the nearest cluster first
    cent = torch.tensor([[0.0, 0.0], [1.0, 0.0], [5.0, 0.0], [9.0, 0.0]])
    cluster_of = torch.tensor([0] * 5 + [1] * 5 + [2] * 5 + [3] * 5)
    torch.manual_seed(3)
    walk = ClusterWalk(cluster_of, cent)
    root0 = walk.root
    got = walk(torch.arange(20), 7)
    assert got.shape[0] == 7 and (cluster_of[got] == root0).sum() == 5  # the whole root cluster, then its neighbour
    assert walk.root != root0


def test_reassign_targets_of_many_partitions_at_once():
    """QuakeIndex._reassign_targets_many (the delete candidates of one maintenance call asked together: lists extracted on the device,
    one nearest-two search per chunk, one unique) == _reassign_targets partition by partition -- with chunks of one, several and
    all partitions, an empty partition among them."""
    import quake_amd as quake
    g = torch.Generator().manual_seed(9)
    x = torch.randn(30000, 24, generator=g)
    idx = quake.QuakeIndex()
    bp = quake.IndexBuildParams()
    bp.nlist = 50
    idx.build(x, torch.arange(30000), bp)
    pids = [int(p) for p in idx._list_ids()][:23]
    idx.remove(torch.from_numpy(idx._store.get_list_ids(pids[4])))  # an empty one
    assert idx._store.list_size(pids[4]) == 0
    one = {p: idx._reassign_targets(p) for p in pids}
    for chunk in (1, 2000, 1 << 18):
        many = idx._reassign_targets_many(pids, chunk_rows=chunk)
        assert set(many) == set(pids)
        for p in pids:
            assert dict(zip(*many[p])) == dict(zip(*one[p])), (p, chunk)
    assert many[pids[4]] == ([], [])


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_splits_on_worker_threads_change_nothing(metric, monkeypatch):
    """QuakeIndex._split_partitions_in_place runs the 2-means of its partitions on worker contexts (QUAKE_SPLIT_THREADS, default 8):
    the same library call on the same rows each -- children, their row order, their ids and the new centroids must be the ones of
    the calls made one after the other (partition_manager.cpp:402-447 splits partition by partition)."""
    import quake_amd as quake
    g = torch.Generator().manual_seed(31)
    x = torch.randn(60000, 32, generator=g)
    if metric == "ip":
        x = torch.nn.functional.normalize(x, dim=1)
    out = {}
    for threads in ("1", "8", "3"):
        monkeypatch.setenv("QUAKE_SPLIT_THREADS", threads)
        idx = quake.QuakeIndex()
        bp = quake.IndexBuildParams()
        bp.nlist = 60
        bp.metric = metric
        idx.build(x, torch.arange(60000), bp)
        pids = [int(p) for p in idx._list_ids()][:37]
        new = idx._split_partitions_in_place(pids)
        assert new is not None and len(new) == 2 * len(pids)
        lists = {int(p): (idx._store.get_list_ids(int(p)).copy(), idx._store.get_list(int(p))[0].copy()) for p in new}
        cents = idx.parent._store.get_list(0)
        out[threads] = (new, lists, cents)
    for threads in ("8", "3"):
        assert out[threads][0] == out["1"][0]
        for p in out["1"][0]:
            np.testing.assert_array_equal(out[threads][1][p][0], out["1"][1][p][0])
            np.testing.assert_array_equal(out[threads][1][p][1].view(np.uint32), out["1"][1][p][1].view(np.uint32))
        np.testing.assert_array_equal(out[threads][2][1], out["1"][2][1])
        np.testing.assert_array_equal(out[threads][2][0].view(np.uint32), out["1"][2][0].view(np.uint32))
