"""BASELINE.json configs[4] at single-GPU scale: a dynamic workload (interleaved add / remove / search with maintenance =
split + delete + k-means refine) replayed on the device index with the harness of quake_amd/workload.py (generate_workload / replay_workload).
    python scripts/dynamic_workload.py [n_base] [dim] [n_ops]
Prints one JSON summary line (per-operation-type latency, recall, partitions over time) and keeps the per-operation records
(and the runbook)
under gpurun_out/dynamic_workload/; the workload files themselves go to /tmp.

    python scripts/dynamic_workload.py [n_base] [dim] [n_ops] hot
the workload that MUST split: the corpus has 8 "hot" mixture components of 10x the members of the others; the index is built over
the cold part, the inserts then pour the hot components in (their lists grow to 10x the mean), the deletes thin the cold part and
half of every query batch asks around the hot components.  Maintenance thresholds are scaled to the device's time scale (below)."""
import json, os, sys, time, shutil
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import quake_amd as quake
from quake_amd.workload import HotSamplers, WorkloadSpec, generate_workload, replay_workload


def hot_corpus(n, d, dev, n_hot_comp=8, factor=10, cluster=2500):
    """n vectors: cold part from n_cold / cluster components of `cluster` members, then n_hot_comp components of factor * cluster
    members each (ids: cold first, hot components one after the other).  Returns (x, queries, n_cold, comp_of_hot)"""
    hot_each = factor * cluster
    n_hot = n_hot_comp * hot_each
    n_cold = n - n_hot
    ncl = max(n_cold // cluster, 16)
    xc, cent = B.gen_mixture(n_cold, d, ncl, seed=1, device=dev)
    g = torch.Generator(device=dev).manual_seed(99)
    hot_cent = torch.randn(n_hot_comp, d, generator=g, device=dev)
    xh = torch.cat([hot_cent[c] + 0.3 * torch.randn(hot_each, d, generator=g, device=dev) for c in range(n_hot_comp)])
    q_cold = B.gen_queries(10000, cent, seed=2, device=dev)
    q_hot = B.gen_queries(10000, hot_cent, seed=3, device=dev)
    return torch.cat([xc, xh]).cpu(), torch.cat([q_cold, q_hot]).cpu(), n_cold, hot_each


def env(name, dflt, typ=int):
    return typ(os.environ[name]) if name in os.environ else dflt


def main():
    # scenario knobs (environment; the defaults are round 5's scenario): DW_HOT_FACTOR members of a hot component over a cold one's,
    # DW_HOT_COMP components, DW_QBATCH queries per batch, DW_HOT_Q fraction of a batch asking around hot components, DW_WINDOW the
    # policy's window, DW_EXT the split_after_delete_rejection extension (index.py MaintenancePolicyParams), DW_NPROBE, DW_TAG
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    hot = len(sys.argv) > 4 and sys.argv[4] == "hot"
    out = "/tmp/dynamic_workload"  # the workload holds a copy of the corpus: keep it out of gpurun_out
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out, exist_ok=True)
    keep = os.path.join(ROOT, "gpurun_out", "dynamic_workload")
    os.makedirs(keep, exist_ok=True)
    dev = torch.device("cuda", 0)
    # torch's CPU ops (the harness indexes host tensors per operation) take an OpenMP team of every hardware thread; on a box that
    # gives the process a fraction of them, a parallel region now and then stalls for ~80 ms -- that, not the index, was most of the
    # 6-10x mean / p50 ratio of the earlier replays' query latencies (the per-operation phases say so: call_ms stays ~0.6)
    torch.set_num_threads(8)
    t0 = time.time()
    if hot:
        x, q, n_cold, hot_each = hot_corpus(n, d, dev, n_hot_comp=env("DW_HOT_COMP", 8), factor=env("DW_HOT_FACTOR", 10))
        hs = HotSamplers(n, n_cold, env("DW_HOT_Q", 0.5, float))
        n_initial = n_cold
        spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.1, query_ratio=0.6, update_batch_size=hot_each // 2,
                            query_batch_size=env("DW_QBATCH", 1024), number_of_operations=n_ops, initial_size=n_cold, cluster_size=2500, seed=1738)
        rb = generate_workload(os.path.join(out, "w"), x, spec, queries=q, update_sampler=hs.update, query_sampler=hs.query)
    else:
        ncl = max(n // 2500, 16)
        x, cent = B.gen_mixture(n, d, ncl, seed=1, device=dev)
        q = B.gen_queries(20000, cent, seed=2, device=dev)
        x, q = x.cpu(), q.cpu()
        n_initial = n // 2
        spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.2, query_ratio=0.5, update_batch_size=max(n // 100, 100),
                            query_batch_size=1024, number_of_operations=n_ops, initial_size=n // 2, cluster_size=2500,
                            cluster_sample_distribution="skewed", query_cluster_sample_distribution="skewed", seed=1738)
        rb = generate_workload(os.path.join(out, "w"), x, spec, queries=q)
    t_gen = time.time() - t0
    # The policy's thresholds are ABSOLUTE nanoseconds of modelled query cost (MaintenancePolicyParams: 10 ns, common.h:116-117), set
    # for the reference's CPU scan, where a 2500-row list costs ~100 us; the device's latency grid (throughput regime: cost of one more
    # (query, list) pair in a busy batch) is ~500x smaller, so with the defaults no delta ever reaches the threshold -- which is why
    # every earlier at-scale replay showed n_splits = 0.  The thresholds are scaled by the ratio of the two time scales: the device's
    # modelled cost of a mean list over 100 us.
    from quake_amd.maintenance import MaintenanceCostEstimator
    ce = MaintenanceCostEstimator(d, 0.9, 10)
    lat = ce.get_latency_estimator()
    L_mean = lat.estimate_scan_latency(2500, 10)
    scale = L_mean / 100_000.0
    grid = {"n": lat.n_values_, "k": lat.k_values_, "ns": [[round(v, 2) for v in row] for row in lat.scan_latency_model_]}
    results = {}
    for name, maint in (("warmup", False), ("static_partitions", False), ("with_maintenance", True)):
        mp = quake.MaintenancePolicyParams()
        # the window must hold several hits of an average partition, or a partition that merely was not asked for in the last two
        # batches looks dead to the policy (the reference's default is 1000 queries -- for its 1000-list test indexes): 8 hits per
        # partition on average, in whole batches (10M / 3920 lists: 4096 queries; 50M / 19920 lists: 20480)
        qb = env("DW_QBATCH", 1024)
        mp.window_size = env("DW_WINDOW", max(2 * qb, -(-8 * (n_initial // 2500) // (env("DW_NPROBE", 8) * qb)) * qb))
        mp.split_after_delete_rejection = bool(env("DW_EXT", 1))
        mp.refinement_radius = 8
        mp.refinement_iterations = 2
        mp.split_threshold_ns = 10.0 * scale
        mp.delete_threshold_ns = 10.0 * scale
        t0 = time.time()
        sp = quake.SearchParams()
        sp.k, sp.nprobe = 10, env("DW_NPROBE", 8)
        index = None
        compiled = os.environ.get("DW_MIRROR", "python") == "compiled"
        if compiled:  # the compiled C++ mirror (`import quake`, quake_amd/cpp/) behind the same harness: same runbook, same policy
            import quake as qc
            first = torch.load(os.path.join(out, "w", "initial_indices.pt"), weights_only=True).to(torch.int64)
            bpc, spc, mpc = qc.IndexBuildParams(), qc.SearchParams(), qc.MaintenancePolicyParams()
            bpc.metric, bpc.nlist = "l2", n_initial // 2500
            spc.k, spc.nprobe = sp.k, sp.nprobe
            for f in ("window_size", "refinement_radius", "refinement_iterations", "split_threshold_ns", "delete_threshold_ns",
                      "split_after_delete_rejection"):
                setattr(mpc, f, getattr(mp, f))
            index = qc.QuakeIndex()
            index.build(x[first], first, bpc)
            if maint:
                prof_csv = os.path.join(out, "latency_grid.csv")
                lat.save_latency_profile(prof_csv)
                index.initialize_maintenance_policy(mpc)
                index.set_latency_profile(prof_csv)
                index.set_track_hits(True)
            sp, mp = spc, mpc
        elif maint:  # (the policy with the grid profiled above: one profiling pass for the whole script)
            import quake_amd
            first = torch.load(os.path.join(out, "w", "initial_indices.pt"), weights_only=True).to(torch.int64)
            bp = quake.IndexBuildParams()
            bp.metric, bp.nlist = "l2", n_initial // 2500
            index = quake_amd.QuakeIndex(device=0)
            index.build(x[first], first, bp)
            index.initialize_maintenance_policy(mp, cost_estimator=ce)
            index.track_hits = True
        prof = None
        if maint and env("DW_PROFILE", 0):
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        res = replay_workload(os.path.join(out, "w"), os.path.join(out, name), name, nlist=n_initial // 2500, search_params=sp,
                              maintenance_params=mp if maint else None, index=index, keep_policy=maint or compiled)
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
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
            "max_list_size_first_last": [res[0].get("max_list_size"), res[-1].get("max_list_size")],
            # the second half of the run, when the hot lists have grown: what the queries cost there
            "query_batch_ms_p50_second_half": round(float(np.median([r["latency_ms"] for r in res[len(res) // 2:]
                                                                      if r["operation_type"] == "query"] or [0.0])), 4),
            "query_scan_ms_p50_second_half": round(float(np.median([r["phases"].get("scan_ms", 0.0) for r in res[len(res) // 2:]
                                                                     if r["operation_type"] == "query"] or [0.0])), 4),
            # the index call alone (QuakeIndex.search, host wall clock: without the harness's tensor indexing and fences) and the sum of
            # its kernel-event phases (coarse + group + scan + merge)
            "query_call_ms_p50_second_half": round(float(np.median([r["phases"].get("call_ms", 0.0) for r in res[len(res) // 2:]
                                                                     if r["operation_type"] == "query"] or [0.0])), 4),
            "query_device_ms_p50_second_half": round(float(np.median([sum(r["phases"].get(k_, 0.0) for k_ in ("coarse_ms", "group_ms", "scan_ms", "merge_ms"))
                                                                       for r in res[len(res) // 2:] if r["operation_type"] == "query"] or [0.0])), 4),
            "pair_rows_p50_second_half": int(np.median([r.get("pair_rows", 0) for r in res[len(res) // 2:] if r["operation_type"] == "query"] or [0])),
            "unique_rows_p50_second_half": int(np.median([r.get("unique_rows", 0) for r in res[len(res) // 2:] if r["operation_type"] == "query"] or [0])),
            "maintenance_ms_p50": median("maintenance_ms", "query"),
            "maintenance_ms_mean_second_half": round(float(np.mean([r.get("maintenance_ms", 0.0) for r in res[len(res) // 2:]])), 2),
            "window_size": mp.window_size, "mirror": "compiled (import quake)" if compiled else "python (quake_amd.index)",
            "maintenance_ms_max": round(max([r.get("maintenance_ms", 0.0) for r in res] or [0.0]), 2),
            "delete_ms_max": round(max([r["latency_ms"] for r in res if r["operation_type"] == "delete"] or [0.0]), 2),
            "trace": [[r["operation_number"], r["operation_type"][0], r["n_list"], r["max_list_size"], r.get("n_splits", 0), r.get("n_deletes", 0),
                       round(r.get("maintenance_ms", 0.0), 1), r["phases"].get("scan_ms"), round(r["latency_ms"], 2), r.get("pair_rows"), r.get("unique_rows")]
                      for r in res],
            "refine_ms_total": round(sum(r.get("maintenance_phases", {}).get("refine_ms", 0.0) for r in res), 2),
            "split_ms_total": round(sum(r.get("maintenance_phases", {}).get("split_ms", 0.0) for r in res), 2),
            # the slow operations, attributed: every operation above 3x its type's median with what it paid for
            "slow_ops": [{"op": r["operation_number"], "type": r["operation_type"], "ms": round(r["latency_ms"], 2),
                          "phases": r["phases"], "store_events": r["store_events"]}
                         for r in res if r["latency_ms"] > 3.0 * (median("latency_ms", r["operation_type"]) or 1e9)][:12],
        }
    results.pop("warmup", None)  # first replay pays one-off costs (module load, staging / workspace growth)
    for name in results:
        shutil.copy(os.path.join(out, name, f"{name}_results.json"), os.path.join(keep, f"{name}_results.json"))
    shutil.copy(os.path.join(out, "w", "runbook.json"), os.path.join(keep, "runbook.json"))
    print(json.dumps({"workload": f"dynamic {n}x{d}, {len(rb['operations'])} ops (30% insert / 20% delete / 50% query batches of 1024), "
                                  f"skewed cluster sampling, nprobe 8, k 10", "summary": rb["summary"],
                      "generate_s": round(t_gen, 1), "scenario": "hot: 8 components of 10x the members poured in by the inserts" if hot else "skewed walk",
                      "thresholds_ns": {"split": 10.0 * scale, "delete": 10.0 * scale, "device_cost_of_a_2500_row_list_ns": round(L_mean, 2),
                                        "reference_cpu_scale_ns": 100000.0},
                      "device_latency_grid": grid, "results": results}), flush=True)


if __name__ == "__main__":
    main()
