"""BASELINE.json configs[4] in shape at 10M x 128 on one GPU, inside the driver-run suite: a 60-operation runbook (inserts,
deletes, batches of 1024 queries) replayed with maintenance on after every operation (maintenance_policies.cpp:33-177: split /
delete / refine on the recorded hits).  The workload is one that MUST make the policy act: 8 mixture components of 10x the
members of the others are poured in by the inserts, so the lists around them grow to many times the mean while half of every
query batch asks there (quake_amd.workload.HotSamplers); the policy's thresholds -- absolute nanoseconds of modelled query cost,
10 ns by default for the reference's CPU scan -- are scaled to the device's time scale (a 2500-row list costs ~0.25 us here against
~100 us there: with the defaults no delta ever reaches them, which is why earlier at-scale replays showed no action).
Checked: the resident set after EVERY operation (the index holds exactly the runbook's live vectors), that partitions WERE split,
that maintenance PAYS against a replay of the same runbook on an unmaintained index -- in VALUES (rows a query batch scores, largest
list, recall), never in a clock: the cost model is a recorded latency grid, so every decision is a function of the runbook --, that the
policy is near its fixed point afterwards (what it still proposes after a few more rounds is a handful of partitions), recall against the exact ground truth the generator stored, the index's own invariants after the
run, and an exhaustive search of the final index against a brute-force scan of the final resident set (ids as sets, distances to
1e-4)."""
import os
import shutil
import tempfile

import numpy as np
import pytest
import torch

import bench as B

pytestmark = pytest.mark.gpu


def test_dynamic_replay_10m_with_maintenance():
    import quake_amd as quake
    from quake_amd.index import QuakeIndex
    from quake_amd.maintenance import (DEFAULT_LATENCY_ESTIMATOR_RANGE_K, DEFAULT_LATENCY_ESTIMATOR_RANGE_N,
                                       ListScanLatencyEstimator, MaintenanceCostEstimator)
    from quake_amd.workload import HotSamplers, WorkloadSpec, generate_workload, replay_workload
    n, d, n_ops = 10_000_000, 128, 60
    hot_comp, hot_each = 8, 25000
    n_cold = n - hot_comp * hot_each
    dev = torch.device("cuda", 0)
    out = tempfile.mkdtemp(prefix="quake_dyn10m_", dir="/tmp")
    try:
        xc, cent = B.gen_mixture(n_cold, d, n_cold // 2500, seed=1, device=dev)
        g = torch.Generator(device=dev).manual_seed(99)
        hot_cent = torch.randn(hot_comp, d, generator=g, device=dev)
        xh = torch.cat([hot_cent[c] + 0.3 * torch.randn(hot_each, d, generator=g, device=dev) for c in range(hot_comp)])
        q = torch.cat([B.gen_queries(10000, cent, seed=2, device=dev), B.gen_queries(10000, hot_cent, seed=3, device=dev)]).cpu()
        x = torch.cat([xc, xh]).cpu()
        del xc, xh
        torch.cuda.empty_cache()
        hs = HotSamplers(n, n_cold)
        spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.1, query_ratio=0.6, update_batch_size=hot_each // 2,
                            query_batch_size=1024, number_of_operations=n_ops, initial_size=n_cold, cluster_size=2500, seed=1738)
        rb = generate_workload(os.path.join(out, "w"), x, spec, queries=q, update_sampler=hs.update, query_sampler=hs.query)
        assert rb["summary"]["n_operations"] == n_ops and min(rb["summary"][k] for k in ("n_inserts", "n_deletes", "n_queries")) >= 3
        first = torch.load(os.path.join(out, "w", "initial_indices.pt"), weights_only=True).to(torch.int64)
        bp = quake.IndexBuildParams()
        bp.metric, bp.nlist = "l2", n_cold // 2500
        index = QuakeIndex(device=0)
        index.build(x[first], first, bp)
        # the cost model is a RECORDED grid (an MI355X profiled by device_profile_fn, tests/golden/; the reference's CSV layout,
        # maintenance_cost_estimator.cpp:259-365), not a measurement of this box at test time: what the policy decides is then a
        # function of the runbook alone
        lat = ListScanLatencyEstimator(d, DEFAULT_LATENCY_ESTIMATOR_RANGE_N, DEFAULT_LATENCY_ESTIMATOR_RANGE_K, 1,
                                       profile_filename=os.path.join(os.path.dirname(__file__), "golden", "device_latency_grid_mi355x_d128.csv"))
        ce = MaintenanceCostEstimator(d, 0.9, 10, latency_estimator=lat)
        scale = ce.get_latency_estimator().estimate_scan_latency(2500, 10) / 100_000.0  # device cost of a mean list over the CPU's
        assert 1e-4 < scale < 0.1, scale
        mp = quake.MaintenancePolicyParams()
        mp.window_size, mp.refinement_radius, mp.refinement_iterations = 2048, 8, 2
        mp.split_threshold_ns, mp.delete_threshold_ns = 10.0 * scale, 10.0 * scale
        # the lists this workload grows are hot AND large: the reference's delete model sends exactly those into its delete branch, the
        # rejection keeps them, and its split test is never reached (maintenance_policies.cpp:68-131) -- the extension examines them
        mp.split_after_delete_rejection = True
        index.initialize_maintenance_policy(mp, cost_estimator=ce)
        index.track_hits = True
        sp = quake.SearchParams()
        sp.k, sp.nprobe = 10, 8
        # the same runbook on an index nobody maintains: what the queries cost there is the bar
        static = replay_workload(os.path.join(out, "w"), os.path.join(out, "static"), "static", nlist=bp.nlist, search_params=sp)
        res = replay_workload(os.path.join(out, "w"), os.path.join(out, "run"), "with_maintenance", nlist=bp.nlist, search_params=sp,
                              maintenance_params=mp, index=index, keep_policy=True)
        assert len(res) == len(static) == n_ops
        # maintenance ACTED: partitions were split
        assert sum(r["n_splits"] for r in res) > 0, [r["n_splits"] for r in res]
        assert index.nlist() != bp.nlist
        # ... and it PAYS, in values no clock is involved in: over the second half of the run (the hot components are in) a query batch
        # scores fewer (query, row) pairs than on the static index -- the arithmetic of the scan --, the largest list is no larger than
        # the static index's, and recall at the same nprobe is no worse
        half = [i for i, r in enumerate(res) if r["operation_type"] == "query" and i >= n_ops // 2]
        assert len(half) >= 5
        pr_m, pr_s = np.median([res[i]["pair_rows"] for i in half]), np.median([static[i]["pair_rows"] for i in half])
        assert pr_m <= 0.5 * pr_s, (pr_m, pr_s)
        assert res[-1]["max_list_size"] <= static[-1]["max_list_size"], (res[-1]["max_list_size"], static[-1]["max_list_size"])
        assert np.mean([res[i]["recall"] for i in half]) >= np.mean([static[i]["recall"] for i in half]) - 0.005
        # ... and approaches the policy's fixed point: a few more windows of the same queries, then what its own cost model still wants
        # deleted or split is a handful of the 4000+ partitions (the children of the last splits whose window has just filled)
        torch.manual_seed(5)
        for rnd in range(8):
            for b in range(2):
                index.search(q[hs.query(torch.arange(q.shape[0]), 1024)], sp)
            t = index.maintenance()
            if t.n_splits == 0 and t.n_deletes == 0:
                break
        for b in range(2):
            index.search(q[hs.query(torch.arange(q.shape[0]), 1024)], sp)
        to_delete, to_split = index._policy().decide()
        assert len(to_split) + len(to_delete) <= index.nlist() // 100, (len(to_delete), len(to_split), index.nlist())
        # resident set after every operation == the runbook's
        for r in res:
            assert r["n_total"] == r["n_resident"], r
        rec = [r["recall"] for r in res if r["operation_type"] == "query"]
        # (half of every batch asks around the hot components: before their vectors arrive those queries are outliers whose true
        #  neighbours are scattered over many lists -- nprobe 8 finds about half of them; once the components are in, recall is high)
        assert len(rec) >= 5 and float(np.mean(rec)) >= 0.7 and float(np.mean(rec[-5:])) >= 0.85, rec
        # the live set, replayed on the host
        live = np.zeros(n, bool)
        live[first.numpy()] = True
        for op_id, op in rb["operations"].items():
            if op["type"] != "query":
                ids = torch.load(os.path.join(out, "w", "operations", f"{op_id}.pt"), weights_only=True).numpy()
                live[ids] = op["type"] == "insert"
        assert int(live.sum()) == index.ntotal()
        sizes = index._partition_sizes(index._list_ids())
        assert sum(sizes) == index.ntotal() and index.nlist() == len(sizes)
        # exhaustive probing of the final index == exact search over the final resident set
        qq = q[:256].contiguous()
        spx = quake.SearchParams()
        spx.k, spx.nprobe = 10, index.nlist()
        got = index.search(qq, spx)
        live_ids = torch.from_numpy(np.nonzero(live)[0])
        bi, bd2 = B.brute_force_topk(qq.to(dev), x[live_ids].to(dev), 10)
        exact = live_ids.to(dev)[bi]
        same = (got.ids.to(dev).sort(dim=1).values == exact.sort(dim=1).values).float().mean().item()
        # (the hot components are 25000 points in a ball of radius 3.4: the 10th and 11th neighbour of a query there can be closer
        #  than the fp32 matmul form of the brute force resolves)
        assert same >= 0.995, same
        np.testing.assert_allclose(got.distances.cpu().numpy(), bd2.clamp(min=0).sqrt().cpu().numpy(), atol=1e-3, rtol=1e-4)
    finally:
        shutil.rmtree(out, ignore_errors=True)
