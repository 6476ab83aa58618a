"""BASELINE.json configs[4] in shape at 10M x 128 on one GPU, inside the driver-run suite: a 60-operation runbook (inserts and
deletes of 100k vectors, batches of 1024 queries; skewed cluster sampling like the reference's workload generator,
test/python/test_workload_generator.py:69-114) replayed with maintenance on after every operation
(maintenance_policies.cpp:33-177: split / delete / refine on the recorded hits).  Checked: the resident set after EVERY
operation (the index holds exactly the runbook's live vectors), recall against the exact ground truth the generator stored,
the index's own invariants after the run, and an exhaustive search of the final index against a brute-force scan of the
final resident set (ids as sets, distances to 1e-4)."""
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
    from quake_amd.workload import WorkloadSpec, generate_workload, replay_workload
    n, d, n_ops = 10_000_000, 128, 60
    dev = torch.device("cuda", 0)
    out = tempfile.mkdtemp(prefix="quake_dyn10m_", dir="/tmp")
    try:
        x, cent = B.gen_mixture(n, d, n // 2500, seed=1, device=dev)
        q = B.gen_queries(20000, cent, seed=2, device=dev)
        x, q = x.cpu(), q.cpu()
        torch.cuda.empty_cache()
        spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.2, query_ratio=0.5, update_batch_size=n // 100,
                            query_batch_size=1024, number_of_operations=n_ops, initial_size=n // 2, cluster_size=2500,
                            cluster_sample_distribution="skewed", query_cluster_sample_distribution="skewed", seed=1738)
        rb = generate_workload(os.path.join(out, "w"), x, spec, queries=q)
        assert rb["summary"]["n_operations"] == n_ops and min(rb["summary"][k] for k in ("n_inserts", "n_deletes", "n_queries")) >= 5
        first = torch.load(os.path.join(out, "w", "initial_indices.pt"), weights_only=True).to(torch.int64)
        bp = quake.IndexBuildParams()
        bp.metric, bp.nlist = "l2", (n // 2) // 2500
        index = QuakeIndex(device=0)
        index.build(x[first], first, bp)
        mp = quake.MaintenancePolicyParams()
        mp.window_size, mp.refinement_radius, mp.refinement_iterations = 2048, 8, 2
        sp = quake.SearchParams()
        sp.k, sp.nprobe = 10, 8
        res = replay_workload(os.path.join(out, "w"), os.path.join(out, "run"), "with_maintenance", nlist=bp.nlist, search_params=sp,
                              maintenance_params=mp, index=index)
        assert len(res) == n_ops
        # resident set after every operation == the runbook's
        for r in res:
            assert r["n_total"] == r["n_resident"], r
        rec = [r["recall"] for r in res if r["operation_type"] == "query"]
        assert len(rec) >= 5 and float(np.mean(rec)) >= 0.9 and min(rec) >= 0.8, rec
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
        assert same >= 0.999, same
        np.testing.assert_allclose(got.distances.cpu().numpy(), bd2.clamp(min=0).sqrt().cpu().numpy(), atol=1e-3, rtol=1e-4)
    finally:
        shutil.rmtree(out, ignore_errors=True)
