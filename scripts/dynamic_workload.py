"""BASELINE.json configs[4] at single-GPU scale: a dynamic workload (interleaved add / remove / search with maintenance =
split + delete + k-means refine) replayed on the device index with the harness of quake_amd/workload.py (generate_workload / replay_workload).
    python scripts/dynamic_workload.py [n_base] [dim] [n_ops]
Prints one JSON summary line (per-operation-type latency, recall, partitions over time) and keeps the per-operation records
(and the runbook)
under gpurun_out/dynamic_workload/; the workload files themselves go to /tmp."""
import json, os, sys, time, shutil
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import quake_amd as quake
from quake_amd.workload import WorkloadSpec, generate_workload, replay_workload


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    out = "/tmp/dynamic_workload"  # the workload holds a copy of the corpus: keep it out of gpurun_out
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out, exist_ok=True)
    keep = os.path.join(ROOT, "gpurun_out", "dynamic_workload")
    os.makedirs(keep, exist_ok=True)
    dev = torch.device("cuda", 0)
    ncl = max(n // 2500, 16)
    x, cent = B.gen_mixture(n, d, ncl, seed=1, device=dev)
    q = B.gen_queries(20000, cent, seed=2, device=dev)
    x, q = x.cpu(), q.cpu()
    t0 = time.time()
    spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.2, query_ratio=0.5, update_batch_size=max(n // 100, 100),
                        query_batch_size=1024, number_of_operations=n_ops, initial_size=n // 2, cluster_size=2500,
                        cluster_sample_distribution="skewed", query_cluster_sample_distribution="skewed", seed=1738)
    rb = generate_workload(os.path.join(out, "w"), x, spec, queries=q)
    t_gen = time.time() - t0
    results = {}
    for name, maint in (("warmup", False), ("static_partitions", False), ("with_maintenance", True)):
        mp = quake.MaintenancePolicyParams()
        mp.window_size = 2048
        mp.refinement_radius = 8
        mp.refinement_iterations = 2
        t0 = time.time()
        sp = quake.SearchParams()
        sp.k, sp.nprobe = 10, 8
        res = replay_workload(os.path.join(out, "w"), os.path.join(out, name), name, nlist=(n // 2) // 2500, search_params=sp,
                              maintenance_params=mp if maint else None)
        wall = time.time() - t0

        def mean(key, typ):
            v = [r[key] for r in res if r["operation_type"] == typ and r.get(key) is not None]
            return round(float(np.mean(v)), 4) if v else None

        def median(key, typ):
            v = [r[key] for r in res if r["operation_type"] == typ and r.get(key) is not None]
            return round(float(np.median(v)), 4) if v else None
        qn = sum(rb["operations"][k]["sample_size"] for k in rb["operations"] if rb["operations"][k]["type"] == "query")
        qt = sum(r["latency_ms"] for r in res if r["operation_type"] == "query") / 1e3
        results[name] = {
            # means include the occasional ~85 ms operation in which a device buffer (arena, workspace, staging) is
            # re-allocated (hipMalloc + hipFree of hundreds of MB); medians are the steady state
            "insert_ms": mean("latency_ms", "insert"), "delete_ms": mean("latency_ms", "delete"),
            "query_batch_ms": mean("latency_ms", "query"), "query_recall_at_10": mean("recall", "query"),
            "insert_ms_p50": median("latency_ms", "insert"), "delete_ms_p50": median("latency_ms", "delete"),
            "query_batch_ms_p50": median("latency_ms", "query"),
            "queries_per_s_incl_host": round(qn / qt, 1) if qt > 0 else None,
            "vectors_inserted_per_s": round(rb["parameters"]["update_batch_size"] / (mean("latency_ms", "insert") / 1e3), 1)
            if mean("latency_ms", "insert") else None,
            "n_list_first_last": [res[0]["n_list"], res[-1]["n_list"]],
            "n_splits": sum(r.get("n_splits", 0) for r in res), "n_deletes": sum(r.get("n_deletes", 0) for r in res),
            "maintenance_ms_mean": mean("maintenance_ms", "query"), "evaluate_wall_s": round(wall, 2),
        }
    results.pop("warmup", None)  # first replay pays one-off costs (module load, staging / workspace growth)
    for name in results:
        shutil.copy(os.path.join(out, name, f"{name}_results.json"), os.path.join(keep, f"{name}_results.json"))
    shutil.copy(os.path.join(out, "w", "runbook.json"), os.path.join(keep, "runbook.json"))
    print(json.dumps({"workload": f"dynamic {n}x{d}, {len(rb['operations'])} ops (30% insert / 20% delete / 50% query batches of 1024), "
                                  f"skewed cluster sampling, nprobe 8, k 10", "summary": rb["summary"],
                      "generate_s": round(t_gen, 1), "results": results}), flush=True)


if __name__ == "__main__":
    main()
