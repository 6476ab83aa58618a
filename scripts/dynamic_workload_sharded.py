"""BASELINE.json configs[4] on N ranks: the dynamic workload of scripts/dynamic_workload.py (interleaved add / remove / search
batches + maintenance) replayed on a CLUSTER-SHARDED index -- quake_amd.sharded_maintenance.ShardedQuakeIndex behind the same
replay_workload harness: every rank executes the runbook; inserts land on the owner of their nearest list, deletes where the
ids live, searches are sharded, maintenance() splits / deletes / refines collectively.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/dynamic_workload_sharded.py [n_base] [dim] [n_ops]
QUAKE_BENCH_BACKEND=gloo lets the ranks share one GPU (functional run; collectives staged through the host).
Rank 0 prints one JSON summary line."""
import json, os, sys, time, shutil
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import quake_amd as quake
from quake_amd.sharded_maintenance import ShardedQuakeIndex
from quake_amd.workload import WorkloadSpec, generate_workload, replay_workload


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n_ops = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("QUAKE_BENCH_BACKEND", "nccl")
    dev_index = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    wdir = "/tmp/dynamic_workload_sharded"
    if rank == 0:
        shutil.rmtree(wdir, ignore_errors=True)
        os.makedirs(wdir, exist_ok=True)
        ncl = max(n // 2500, 16)
        x, cent = B.gen_mixture(n, d, ncl, seed=1, device=dev)
        q = B.gen_queries(20000, cent, seed=2, device=dev)
        spec = WorkloadSpec(metric="l2", insert_ratio=0.3, delete_ratio=0.2, query_ratio=0.5, update_batch_size=max(n // 100, 100),
                            query_batch_size=1024, number_of_operations=n_ops, initial_size=n // 2, cluster_size=2500,
                            cluster_sample_distribution="skewed", query_cluster_sample_distribution="skewed", seed=1738)
        generate_workload(os.path.join(wdir, "w"), x.cpu(), spec, queries=q.cpu())
        del x, q
    if world > 1:
        dist.barrier()
    root = os.path.join(wdir, "w")
    rb = json.load(open(os.path.join(root, "runbook.json")))
    base = torch.load(os.path.join(root, "base_vectors.pt"), weights_only=True).to(torch.float32)
    first = torch.load(os.path.join(root, "initial_indices.pt"), weights_only=True).to(torch.int64)
    nlist = ((n // 2) // 2500 // world) * world
    results = {}
    for name, maint in (("warmup", False), ("static_partitions", False), ("with_maintenance", True)):
        # the initial clustering is computed by every rank on the whole initial set (same seed, same kernels: identical), each
        # rank keeps the lists it owns
        bp = quake.IndexBuildParams()
        bp.metric, bp.nlist, bp.niter = "l2", nlist, 5
        full = quake.QuakeIndex(device=dev_index)
        full.build(base[first], first, bp)
        pids = [int(p) for p in full._store.list_ids()]
        cent = full.parent.get(torch.tensor(pids, dtype=torch.int64)).numpy()
        vs, is_, offs = [], [], [0]
        for p in pids:
            v, i = full._store.get_list(p)
            vs.append(v)
            is_.append(i)
            offs.append(offs[-1] + len(i))
        del full
        sh = ShardedQuakeIndex.from_global(dist if world > 1 else None, world, rank, cent, np.array(offs, np.int64),
                                           np.concatenate(is_), np.concatenate(vs), "l2", device=dev_index)
        mp = quake.MaintenancePolicyParams()
        mp.window_size = 2048
        mp.refinement_radius = 8
        mp.refinement_iterations = 2
        sp = quake.SearchParams()
        sp.k, sp.nprobe = 10, 8
        t0 = time.time()
        res = replay_workload(root, os.path.join(wdir, f"{name}_r{rank}"), name, nlist=nlist, search_params=sp,
                              maintenance_params=mp if maint else None, index=sh, device=dev_index)
        wall = time.time() - t0

        def stat(key, typ, fn):
            v = [r[key] for r in res if r["operation_type"] == typ and r.get(key) is not None]
            return round(float(fn(v)), 4) if v else None
        results[name] = {
            "insert_ms_p50": stat("latency_ms", "insert", np.median), "delete_ms_p50": stat("latency_ms", "delete", np.median),
            "query_batch_ms_p50": stat("latency_ms", "query", np.median), "query_recall_at_10": stat("recall", "query", np.mean),
            "n_list_first_last": [res[0]["n_list"], res[-1]["n_list"]], "n_total_last": res[-1]["n_total"],
            "n_resident_last": res[-1]["n_resident"],
            "n_splits": sum(r.get("n_splits", 0) for r in res), "n_deletes": sum(r.get("n_deletes", 0) for r in res),
            "maintenance_ms_mean": stat("maintenance_ms", "query", np.mean), "evaluate_wall_s": round(wall, 2),
            "local_vectors_last": int(sh.index.ntotal()),
        }
        assert res[-1]["n_total"] == res[-1]["n_resident"], (res[-1]["n_total"], res[-1]["n_resident"])  # the index tracks the runbook
    results.pop("warmup", None)
    if rank == 0:
        print(json.dumps({"workload": f"dynamic {n}x{d}, {len(rb['operations'])} ops (30% insert / 20% delete / 50% query batches of 1024), "
                                      f"skewed cluster sampling, nprobe 8, k 10, lists sharded over {world} ranks ({backend})",
                          "summary": rb["summary"], "results": results}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
