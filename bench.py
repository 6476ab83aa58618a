#!/usr/bin/env python
"""bench.py -- QuakeIndex::search() hot path on MI355X: queries/sec at recall@10 >= 0.9.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): synthetic 10M x 128 f32 Gaussian mixture (4096 centres
~N(0,1), within-cluster sigma 0.3), L2, nlist=4096 built with the GPU k-means, batch of 1024 queries, k=10.
One "step" = one qk_search() of the whole batch (coarse + partition scan + merge) with queries, index and outputs
resident in HBM.  nprobe = the smallest of {1,2,4,...,64} reaching recall@10 >= 0.9 against exact brute force.
N > 1: weak scaling -- every rank owns its own 10M-vector shard (4096 lists, cluster-sharded by list number), the
batch is N*1024 queries, centroids are replicated, each rank scans the probed lists it owns and the per-rank top-k
are all-gathered over RCCL and merged.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_scan, HBM-bound): algorithmic bytes per
launch (sum over unique probed partitions of n_p*d*4) / the kernel's mean duration measured with HIP events recorded
on the launch stream inside the timed region.  `cpu_baseline` times oracle/ (the CPU port of the reference path) on
the host cores on a bounded sample of the same queries.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def gen_mixture(n, d, ncent, seed, device, sigma=0.3, chunk=1 << 20, unit=False):
    g = torch.Generator(device=device).manual_seed(seed)
    cent = torch.randn(ncent, d, generator=g, device=device)
    x = torch.empty(n, d, device=device)
    for i0 in range(0, n, chunk):
        m = min(chunk, n - i0)
        a = torch.randint(0, ncent, (m,), generator=g, device=device)
        v = cent[a] + sigma * torch.randn(m, d, generator=g, device=device)
        x[i0:i0 + m] = torch.nn.functional.normalize(v, dim=1) if unit else v
    return x, cent


def gen_queries(nq, cent_all, seed, device, sigma=0.3, unit=False):
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randint(0, cent_all.shape[0], (nq,), generator=g, device=device)
    v = cent_all[a] + sigma * torch.randn(nq, cent_all.shape[1], generator=g, device=device)
    return (torch.nn.functional.normalize(v, dim=1) if unit else v).contiguous()


def brute_force_topk(q, x, k, id_base=0, chunk=1 << 20, metric="l2"):
    """exact top-k (fp32 matmul; squared L2 in expanded form, or negated inner product); returns (ids [Q,k], key [Q,k])
    with smaller key = better."""
    qn = (q * q).sum(1, keepdim=True)
    best_d = torch.full((q.shape[0], k), float("inf"), device=q.device)
    best_i = torch.full((q.shape[0], k), -1, dtype=torch.int64, device=q.device)
    for i0 in range(0, x.shape[0], chunk):
        xc = x[i0:i0 + chunk]
        d2 = (qn + (xc * xc).sum(1)[None, :] - 2.0 * (q @ xc.T)) if metric == "l2" else -(q @ xc.T)
        v, i = torch.topk(d2, min(k, xc.shape[0]), dim=1, largest=False)
        cd = torch.cat([best_d, v], 1)
        ci = torch.cat([best_i, i + i0 + id_base], 1)
        v2, j = torch.topk(cd, k, dim=1, largest=False)
        best_d, best_i = v2, torch.gather(ci, 1, j)
    return best_i, best_d


def recall_at_k(ids, gt, k):
    """src/python/utils.py compute_recall: per-query |set(ids) & set(gt)| / k, averaged."""
    hit = (ids[:, :k, None] == gt[:, None, :k]).any(2).float().sum(1) / k
    return hit.mean().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=200,
                    help="untimed steps before the warmup (the first ~100 steps after the build run at lower clocks)")
    ap.add_argument("--nvec", type=int, default=10_000_000, help="vectors per GPU")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=4096, help="lists per GPU")
    ap.add_argument("--batch", type=int, default=1024, help="queries per GPU per step")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", choices=("l2", "ip"), default="l2",
                    help="ip: unit-norm mixture (embedding-like), BASELINE.json configs[2] with --dim 768 --k 100")
    ap.add_argument("--nprobe", type=int, default=0, help="0 = sweep for recall@k >= target")
    ap.add_argument("--recall-target", type=float, default=0.9)
    ap.add_argument("--niter", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # QUAKE_BENCH_BACKEND=gloo lets two ranks share one GPU (functional check of the N>1 path on a 1-GPU box)
    backend = os.environ.get("QUAKE_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from quake_amd.capi import Context, Store
    ctx = Context(dev_index)
    # the library runs on torch's current stream, so that it is ordered with torch / torch.distributed work
    # (QUAKE_BENCH_STREAM=private: A/B against the context's own non-blocking stream, single GPU only)
    if not (os.environ.get("QUAKE_BENCH_STREAM") == "private" and world == 1):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    info = ctx.device_info()
    log("device", info)

    n, d, nlist, k = args.nvec, args.dim, args.nlist, args.k
    Q = args.batch * world
    t_all = time.time()

    # ---- corpus shard + index build (untimed) -------------------------------------------------------------
    t0 = time.time()
    metric = args.metric
    unit = metric == "ip"
    x, cent_true = gen_mixture(n, d, nlist, seed=1 + 100 * rank, device=dev, unit=unit)
    torch.cuda.synchronize()
    log(f"generated {n}x{d} shard in {time.time() - t0:.1f}s")
    t0 = time.time()
    centroids, assign, _ = ctx.kmeans(x, nlist, metric, niter=args.niter, seed=1234)
    torch.cuda.synchronize()
    t_kmeans = time.time() - t0
    log(f"k-means nlist={nlist} niter={args.niter}: {t_kmeans:.2f}s")
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    id_base = rank * n
    ids_sorted = (order + id_base).contiguous()
    x_sorted = x[order].contiguous()
    del order, assign
    nlist_g = nlist * world
    offsets = np.zeros(nlist_g + 1, np.int64)
    offsets[rank * nlist + 1:(rank + 1) * nlist + 1] = np.cumsum(counts)
    offsets[(rank + 1) * nlist + 1:] = offsets[(rank + 1) * nlist]
    store = Store(ctx, d)
    t0 = time.time()
    store.build_csr(offsets, ids_sorted, x_sorted)
    torch.cuda.synchronize()
    log(f"store upload {time.time() - t0:.2f}s, sizes min/mean/max = {counts.min()}/{counts.mean():.0f}/{counts.max()}, "
        f"arena {store.device_bytes() / 1e9:.2f} GB")
    host_csr = None
    if rank == 0 and world == 1 and not args.no_cpu:
        host_csr = (x_sorted.cpu().numpy(), ids_sorted.cpu().numpy(), offsets.copy(), centroids.cpu().numpy())
    del x_sorted, ids_sorted
    # replicated centroids (parent index over all ranks' lists)
    if world > 1:
        cl = [torch.empty_like(centroids) for _ in range(world)]
        dist.all_gather(cl, centroids.contiguous())
        cent_all = torch.cat(cl, 0)
        tl = [torch.empty_like(cent_true) for _ in range(world)]
        dist.all_gather(tl, cent_true.contiguous())
        cent_true_all = torch.cat(tl, 0)
    else:
        cent_all, cent_true_all = centroids, cent_true
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist_g], np.int64), torch.arange(nlist_g, device=dev), cent_all.contiguous())

    # ---- queries + exact ground truth --------------------------------------------------------------------------
    q = gen_queries(Q, cent_true_all, seed=2, device=dev, unit=unit)
    t0 = time.time()
    gi, gd2 = brute_force_topk(q, x, k, id_base=id_base, metric=metric)
    if world > 1:
        gl_i = [torch.empty_like(gi) for _ in range(world)]
        gl_d = [torch.empty_like(gd2) for _ in range(world)]
        dist.all_gather(gl_i, gi)
        dist.all_gather(gl_d, gd2)
        ci, cd = torch.cat(gl_i, 1), torch.cat(gl_d, 1)
        _, j = torch.topk(cd, k, dim=1, largest=False)
        gi = torch.gather(ci, 1, j)
    torch.cuda.synchronize()
    log(f"brute-force ground truth {time.time() - t0:.2f}s")
    del x

    out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
    out_d = torch.empty((Q, k), dtype=torch.float32, device=dev)
    sharded = None
    if world > 1:
        # ranks exchange the merge key (squared distance) with one all-gather over RCCL; sqrt after the merge
        from quake_amd.sharded import GpuEngine, ShardedIndex
        sharded = ShardedIndex(GpuEngine(ctx, parent, store, metric), dist, world, rank, result="owner")

    def step(nprobe):
        if sharded is not None:
            return sharded.search(q, nprobe, k, out=(out_i, out_d))
        return ctx.search(parent, store, q, nprobe, k, metric, out=(out_i, out_d))

    def batch_recall(ri):
        """recall@k of the batch: with N ranks every rank holds the answer of its slice of the queries."""
        if sharded is None:
            return recall_at_k(ri, gi, k)
        per = Q // world
        r = torch.tensor([recall_at_k(ri, gi[rank * per:(rank + 1) * per], k)], device=dev, dtype=torch.float64)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return r.item() / world

    # ---- nprobe: smallest reaching the recall target ----------------------------------------------------------------
    sweep = []
    nprobe = args.nprobe
    if nprobe <= 0:
        for p in (1, 2, 4, 8, 16, 32, 64):
            ri, _ = step(p)
            torch.cuda.synchronize()
            r = batch_recall(ri)
            sweep.append((p, round(r, 4)))
            if r >= args.recall_target:
                nprobe = p
                break
        if nprobe <= 0:
            nprobe = 64
    ri, rd = step(nprobe)
    torch.cuda.synchronize()
    recall = batch_recall(ri)
    log(f"nprobe={nprobe} recall@{k}={recall:.4f} sweep={sweep}")

    # per-call phase breakdown + algorithmic bytes (one synchronising call, outside the timed region)
    ctx.set_timing(1)
    _, _, tinfo = ctx.search(parent, store, q, nprobe, k, metric, timing=True)
    scan_bytes = int(tinfo["scan_bytes"])
    log("phases (ms):", {kk: round(v, 4) if isinstance(v, float) else v for kk, v in tinfo.items()})

    # ---- timed region -------------------------------------------------------------------------------------------------
    ctx.set_timing(0)
    for _ in range(max(args.settle, 0)):
        step(nprobe)
    for _ in range(args.warmup):
        step(nprobe)
    # one HIP event pair per step around the scan kernel, recorded on the launch stream, read after the region (mode 2 --
    # events around every phase -- costs the step ~10 %: it is used for the phase breakdown below, outside the region)
    ctx.set_timing(3)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(nprobe)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev = ctx.read_timing()
    ctx.set_timing(2)  # phase breakdown: a short untimed pass with events around every phase
    for _ in range(min(args.steps, 20)):
        step(nprobe)
    torch.cuda.synchronize()
    ev_ph = ctx.read_timing()
    ctx.set_timing(0)
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    ms_per_step = 1e3 * elapsed / args.steps
    qps = Q * args.steps / elapsed
    scan_ms = ev["scan_ms"] / max(ev["calls"], 1)
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0

    # measured HBM traffic of k_scan from the committed rocprofv3 PMC pass of this same command, if present
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_k_scan.json")
    if os.path.exists(pmc_path) and world == 1:
        try:
            pj = json.load(open(pmc_path))
            if pj.get("nvec") == n and pj.get("nprobe") == nprobe and pj.get("dim", 128) == d and pj.get("k", 10) == k:
                traffic = pj.get("traffic_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "queries/sec at recall@10\u22650.9 (SIFT1M, k=10); 1/2/4/8 GPU",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"Synthetic {n * world // 1_000_000}M x {d} f32 {metric.upper()} "
                        f"{'unit-norm ' if unit else ''}Gaussian mixture, nlist={nlist_g}, "
                        f"batch={Q} queries, k={k}, nprobe={nprobe} "
                        f"(BASELINE.json configs[{2 if (metric == 'ip' and d == 768) else 1}] per GPU)",
            "nvec_per_gpu": n, "dim": d, "metric_type": metric, "nlist_per_gpu": nlist, "batch": Q, "k": k, "nprobe": nprobe,
            "recall_at_k": round(recall, 4), "recall_sweep": sweep, "settle_steps": max(args.settle, 0),
            "sharding": ("lists by number across ranks, centroids replicated; all-gather of the probed-list ids, local scan, "
                         "all-to-all of the per-rank top-k, merge on the rank that owns the query") if world > 1 else "single GPU",
        },
        "roofline": {
            "kernel": "k_scan",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": scan_bytes,
            "kernel_ms_avg": round(scan_ms, 5),
            "launches": ev["calls"],
        },
        "phases_ms": {"coarse": round(ev_ph["coarse_ms"] / max(ev_ph["calls"], 1), 4),
                      "group": round(ev_ph["group_ms"] / max(ev_ph["calls"], 1), 4),
                      "scan": round(ev_ph["scan_ms"] / max(ev_ph["calls"], 1), 4),
                      "merge": round(ev_ph["merge_ms"] / max(ev_ph["calls"], 1), 4),
                      "note": "separate untimed pass with events around every phase"},
        "build": {"kmeans_s": round(t_kmeans, 2), "niter": args.niter},
    }

    # ---- CPU baseline: the oracle port on the host cores, bounded sample (rank 0, N=1 only) ---------------------------
    if host_csr is not None:
        import oracle as O
        hv, hi, ho, hc = host_csr
        qh = q.cpu().numpy()
        cores = O.max_threads()
        # bounded sample: the bench batch is replayed until ~cpu_seconds of host work have been timed, for both of the
        # reference's scan variants (serial_scan = its default, batched_serial_scan = SearchParams::batched_scan);
        # the faster one is reported as the baseline
        def time_cpu(batched, budget):
            t, n, reps, ids = 0.0, 0, 0, None
            while t < budget and reps < 10000:
                t0 = time.perf_counter()
                ids, _ = O.search(qh, hc, hv, hi, ho, nprobe, k, metric, batched_scan=batched, num_threads=cores)
                t += time.perf_counter() - t0
                n += Q
                reps += 1
            return n / t, n, reps, t, ids

        qps_serial, ns_s, reps_s, t_s, ci_ = time_cpu(False, args.cpu_seconds * 0.6)
        qps_batched, ns_b, reps_b, t_b, _ = time_cpu(True, args.cpu_seconds * 0.4)
        best_batched = qps_batched > qps_serial
        cpu_qps = max(qps_serial, qps_batched)
        ns, reps, t_cpu = (ns_b, reps_b, t_b) if best_batched else (ns_s, reps_s, t_s)
        n1 = max(1, min(Q, int(3.0 * qps_serial / max(cores, 1) * 4) or 1))  # a few seconds single-threaded
        t0 = time.perf_counter()
        O.search(qh[:n1], hc, hv, hi, ho, nprobe, k, metric, batched_scan=False, num_threads=1)
        t_cpu1 = time.perf_counter() - t0
        # the CPU path returns the same neighbours (direct-form L2 vs expanded: ids equal unless near-tied)
        same = float((ci_ == ri.cpu().numpy()).mean())
        result["cpu_baseline"] = {
            "value": round(cpu_qps, 1), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"the {Q}-query bench batch replayed {reps}x ({ns} queries), same index/nprobe/k, oracle search() = "
                      f"coarse + {'batched_serial_scan' if best_batched else 'serial_scan'} semantics on {cores} threads, "
                      f"{t_cpu:.1f}s (the faster of the reference's two scan variants)",
            "serial_scan_qps": round(qps_serial, 1), "batched_scan_qps": round(qps_batched, 1),
            "single_thread_qps": round(n1 / t_cpu1, 1),
            "ids_equal_to_gpu_frac": round(same, 5),
        }
        result["speedup_vs_cpu"] = round(qps / cpu_qps, 1)
    elif world == 1:
        result["cpu_baseline"] = None

    log(f"total bench wall {time.time() - t_all:.1f}s")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
