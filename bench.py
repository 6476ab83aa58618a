#!/usr/bin/env python
"""bench.py -- QuakeIndex::search() hot path on MI355X: queries/sec at recall@10 >= 0.9.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # N = 2, 4, 8: starts its own N ranks (launch_ranks), or, equivalently,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): synthetic 10M x 128 f32 Gaussian mixture (4096
centres ~N(0,1), within-cluster sigma 0.3), L2, nlist=4096 built with the GPU k-means, batches of 1024 queries, k=10.
One "step" = one qk_search() of a whole batch (coarse + partition scan + merge) with queries, index and outputs resident
in HBM; the timed region rotates over 4 different query batches.  nprobe = the smallest of {1,2,4,...,64} reaching
recall@10 >= 0.9 against exact brute force.

Next to the headline line the same JSON object carries (N = 1 only, `--no-extra` skips them):
  workloads.hard      the same sizes on a corpus WITHOUT cluster structure (x = zA + noise, latent dimension 10: a smooth
                      low-intrinsic-dimension density like SIFT / embeddings): the recall target NEEDS nprobe >> 1 -- the
                      regime where a probed partition is shared by many queries of the batch; reported with its own
                      `roofline` (unique bytes / k_scan time)
  workloads.configs0  BASELINE.json configs[0] shape on the S-SIFT stand-in (1M x 128 integer-valued f32,
                      nlist=1024, nprobe=10, k=10, batch=1): GPU single-query rate beside the CPU port on ONE thread (the
                      reference default: num_workers=0, num_threads=1, common.h:73,175), same ids

N > 1 (BASELINE.json configs[3] shape, one rank per GPU): the corpus is 12.5M vectors per rank, ONE global k-means over
all ranks (local assign + accumulate, all-reduce of sums and counts per iteration: quake_amd/sharded.py) produces
nlist = 8192*N global centroids, replicated; list p lives on rank p % N; the batch is 512*N queries; every rank computes
the coarse step, scans the probed lists it owns and the per-rank top-k meet in one all-to-all + merge.

`roofline` is for the dominant kernel (k_scan, HBM-bound): algorithmic bytes per launch (sum over unique probed
partitions of n_p*d*4) / the kernel's mean duration measured with HIP events recorded on the launch stream inside the
timed region.  `cpu_baseline` times oracle/ (the CPU port of the reference path) on the host cores on a bounded sample
of the same queries; the ids of the bench batch must equal the batched oracle's, bit for bit, or the run fails.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak (MI355X_MICROARCH.md); the 16x16x4 f32 instruction reaches 142 in isolation
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
METRIC_NAME = "queries/sec at recall@10≥0.9 (SIFT1M, k=10); 1/2/4/8 GPU"
N_BATCHES = 4  # query batches the timed region rotates over


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ---- synthetic corpora ---------------------------------------------------------------------------
def gen_mixture(n, d, ncent, seed, device, sigma=0.3, chunk=1 << 20, unit=False, cent=None):
    g = torch.Generator(device=device).manual_seed(seed)
    if cent is None:
        cent = torch.randn(ncent, d, generator=g, device=device)
    x = torch.empty(n, d, device=device)
    for i0 in range(0, n, chunk):
        m = min(chunk, n - i0)
        a = torch.randint(0, cent.shape[0], (m,), generator=g, device=device)
        v = cent[a] + sigma * torch.randn(m, d, generator=g, device=device)
        x[i0:i0 + m] = torch.nn.functional.normalize(v, dim=1) if unit else v
    return x, cent


def gen_queries(nq, cent_all, seed, device, sigma=0.3, unit=False):
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randint(0, cent_all.shape[0], (nq,), generator=g, device=device)
    v = cent_all[a] + sigma * torch.randn(nq, cent_all.shape[1], generator=g, device=device)
    return (torch.nn.functional.normalize(v, dim=1) if unit else v).contiguous()


def gen_manifold(n, d, seed, device, latent=10, noise=0.05, basis=None, chunk=1 << 20):
    """Low-intrinsic-dimension corpus: x = z A + noise, z ~ N(0, I_latent), A a fixed [latent, d] basis.  No cluster structure for
    k-means to find: the lists are Voronoi cells of a smooth density, a query's neighbours straddle several cells and the
    recall target needs nprobe >> 1 -- the regime real embedding / SIFT-like data put an IVF index in."""
    g = torch.Generator(device=device).manual_seed(seed)
    if basis is None:
        gb = torch.Generator(device=device).manual_seed(777)
        basis = torch.randn(latent, d, generator=gb, device=device) / (latent ** 0.5)
    x = torch.empty(n, d, device=device)
    for i0 in range(0, n, chunk):
        m = min(chunk, n - i0)
        z = torch.randn(m, basis.shape[0], generator=g, device=device)
        x[i0:i0 + m] = z @ basis + noise * torch.randn(m, d, generator=g, device=device)
    return x, basis


def gen_ssift(n, device, seed=1234, d=128, ncomp=1024, sigma=25.0, cent=None, chunk=1 << 20):
    """S-SIFT, the stand-in for SIFT1M: integer-valued f32 in [0, 218] drawn from a 1024-component Gaussian
    mixture (centres ~U[20,120]^d, sigma 25, clipped, rounded).  Every fp32 partial sum of a distance is an exact integer
    (< 2^24), so the direct and the expanded L2 forms agree bit for bit."""
    g = torch.Generator(device=device).manual_seed(seed)
    if cent is None:
        cent = 20.0 + 100.0 * torch.rand(ncomp, d, generator=g, device=device)
    x = torch.empty(n, d, device=device)
    for i0 in range(0, n, chunk):
        m = min(chunk, n - i0)
        a = torch.randint(0, cent.shape[0], (m,), generator=g, device=device)
        v = cent[a] + sigma * torch.randn(m, d, generator=g, device=device)
        x[i0:i0 + m] = torch.clamp(torch.round(v), 0.0, 218.0)
    return x, cent


def read_vecs(path, dtype):
    """.fvecs / .ivecs / .bvecs (TEXMEX layout: every vector is an int32 dimension followed by d components) -> [n, d] array.  Cf. the
    reference's readers, src/python/utils.py (fvecs_read / ivecs_read) and datasets/ann_datasets.py:44."""
    dtype = np.dtype(dtype)
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 4:
        raise SystemExit(f"{path}: empty")
    d = int(raw[:4].view(np.int32)[0])
    rec = 4 + d * dtype.itemsize
    if d <= 0 or raw.size % rec:
        raise SystemExit(f"{path}: not a vecs file of {dtype} (d={d}, {raw.size} bytes)")
    return np.ascontiguousarray(raw.reshape(-1, rec)[:, 4:]).view(dtype).reshape(-1, d)


def load_sift_dir(path):
    """a TEXMEX directory (sift/: sift_base.fvecs, sift_query.fvecs, sift_groundtruth.ivecs; any <name>_base / _query / _groundtruth)"""
    def one(suffix, dtype):
        for f in sorted(os.listdir(path)):
            if f.endswith(suffix):
                return read_vecs(os.path.join(path, f), dtype)
        raise SystemExit(f"--sift-dir {path}: no *{suffix}")
    base, query = one("_base.fvecs", np.float32), one("_query.fvecs", np.float32)
    try:
        gt = one("_groundtruth.ivecs", np.int32)
    except SystemExit:
        gt = None
    return base, query, gt


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


# ---- single-GPU index build (untimed) -------------------------------------------------------------------------------------
def build_single(ctx, dev, x, nlist, metric, niter, keep_host):
    """k-means on the GPU, vectors bucketed by list, device store + flat parent over the centroids."""
    from quake_amd.capi import Store
    n, d = x.shape
    t0 = time.time()
    centroids, assign, _ = ctx.kmeans(x, nlist, metric, niter=niter, seed=1234)
    torch.cuda.synchronize()
    t_kmeans = time.time() - t0
    kt = ctx.kmeans_last_timing()  # last Lloyd iteration, HIP events inside the library
    km_kernels = None
    if kt["rows"] > 0 and kt["assign_ms"] > 0 and kt["update_ms"] > 0:
        a_tf = 2.0 * kt["rows"] * kt["m"] * d / (kt["assign_ms"] * 1e-3) / 1e12
        u_gbs = kt["rows"] * d * 4 / (kt["update_ms"] * 1e-3) / 1e9
        km_kernels = {"rows": kt["rows"], "centroids": kt["m"],
                      # the assign step settles WHICH key decides on bf16 MFMA (two passes over all pairs: qk_assign_pf.hip) and
                      # computes only the deciding keys exactly: `achieved` counts one key per (row, centroid) pair, `executed` the
                      # two bf16 passes, priced against the dense bf16 MFMA peak; an all-fp32 kernel is capped at 157.3 TFLOP/s
                      # (rows wider than 128 columns take the all-fp32 kernel k_assign: priced against the fp32 MFMA peak)
                      "assign": ({"ms": round(kt["assign_ms"], 3), "achieved": round(a_tf, 1), "executed": round(2 * a_tf, 1),
                                  "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(2 * a_tf / MFMA_BF16_PEAK_TFLOPS, 3),
                                  "bound": "mfma (bf16 prefilter, exact fp32 keys for the candidates)",
                                  "achieved_over_fp32_mfma_peak": round(a_tf / MFMA_F32_PEAK_TFLOPS, 2)} if d <= 128 else
                                 {"ms": round(kt["assign_ms"], 3), "achieved": round(a_tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(a_tf / MFMA_F32_PEAK_TFLOPS, 3), "bound": "mfma (fp32: k_assign, rows wider than 128 columns)"}),
                      "update": {"ms": round(kt["update_ms"], 3), "achieved": round(u_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(u_gbs / HBM_PEAK_GBS, 3), "bound": "hbm",
                                 "note": "rows x d x 4 bytes (each training row read once) / (bucketing by assignment + k_accumulate)"}}
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    ids_sorted = order.contiguous()
    x_sorted = x[order].contiguous()
    del order, assign
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(counts)
    store = Store(ctx, d)
    store.build_csr(offsets, ids_sorted, x_sorted)
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
    torch.cuda.synchronize()
    host = None
    if keep_host:
        host = (x_sorted.cpu().numpy(), ids_sorted.cpu().numpy(), offsets.copy(), centroids.cpu().numpy())
    log(f"index: {n}x{d} nlist={nlist} k-means {t_kmeans:.2f}s, list sizes min/mean/max = "
        f"{counts.min()}/{counts.mean():.0f}/{counts.max()}, arena {store.device_bytes() / 1e9:.2f} GB")
    return dict(parent=parent, store=store, host=host, kmeans_s=t_kmeans, counts=counts, kmeans_kernels=km_kernels)


def pick_nprobe(step, batches, gts, k, target, fixed, recall_fn):
    """smallest nprobe of {1,2,4,...,64} whose recall@k over the batches reaches the target (or the fixed one)."""
    sweep = []

    def rec(p):
        r = 0.0
        for b in range(len(batches)):
            ri, _ = step(p, b)
            torch.cuda.synchronize()
            r += recall_fn(ri, b)
        return r / len(batches)

    nprobe = fixed
    if nprobe <= 0:
        for p in (1, 2, 4, 8, 16, 32, 64, 128):
            r = rec(p)
            sweep.append((p, round(r, 4)))
            if r >= target:
                nprobe = p
                break
        if nprobe <= 0:
            nprobe = 128
    return nprobe, rec(nprobe), sweep


MIN_TIMED_STEPS = 200  # a 20-step region of a 0.3 ms step is 6 ms: one clock excursion moves it by 2-3 %


def timed_region(ctx, step, nprobe, steps, warmup, settle, dist, dev, ctxs=None, groups=None, sync=None):
    """settle + warmup untimed, then EXACTLY `steps` steps between barrier + synchronize; one HIP event pair per step around
    the scan kernel (timing mode 3).  ctxs: the contexts the steps rotate over (step i runs on ctxs[i % len]; each has its own
    stream and output buffers, so len(ctxs) batches are in flight at a time); default [ctx].
    The region is run `groups` times (default: as many as it takes to time MIN_TIMED_STEPS steps in all, at most 10) -- every
    group is exactly `steps` steps between barrier + synchronize on both sides -- and the MEDIAN group is the one reported:
    `elapsed` is its wall time (max over ranks); group_times lists them all.
    sync: extra synchronisation of the path under test (the device group's members), called beside the contexts'.
    Returns (elapsed seconds, scan-kernel event sums over all groups, phase sums, group_times)."""
    ctxs = ctxs or [ctx]
    nc = len(ctxs)
    if groups is None:
        groups = max(1, min(10, -(-MIN_TIMED_STEPS // max(steps, 1))))

    def sync_all():
        for c in ctxs:
            c.synchronize()
        if sync is not None:
            sync()
        torch.cuda.synchronize()

    for c in ctxs:
        c.set_timing(0)
    # form feedback (qk_ctx_set_form_feedback): a context measures the admissible forms of the scan on the first calls of a
    # shape -- each twice, one measurement in flight at a time, read back at a later call.  Give it synchronised calls to finish
    # that before anything is timed (a few of them run a form that loses: they belong to the untimed part)
    for c_i in range(nc):
        for i in range(16):
            step(nprobe, i % N_BATCHES, c_i)
            ctxs[c_i].synchronize()
            torch.cuda.synchronize()
    for i in range(max(settle, 0)):
        step(nprobe, i % N_BATCHES, i % nc)
    for i in range(warmup):
        step(nprobe, i % N_BATCHES, i % nc)
    for c in ctxs:
        c.set_timing(3)
    times = []
    for _ in range(groups):
        if dist is not None:
            dist.barrier()
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step(nprobe, i % N_BATCHES, i % nc)
        if dist is not None:
            dist.barrier()
        sync_all()
        times.append(time.perf_counter() - t0)
    if dist is not None:  # a group's time is the slowest rank's
        tt = torch.tensor(times, dtype=torch.float64, device=dev if str(dist.get_backend()).lower() != "gloo" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        times = tt.tolist()
    elapsed = sorted(times)[(len(times) - 1) // 2]  # the median group (the lower one of an even count)
    ev = None
    for c in ctxs:  # the scan-kernel event pairs of all streams
        e = c.read_timing()
        ev = e if ev is None else {kk: ev[kk] + e[kk] for kk in ev}
    ctx.set_timing(2)  # phase breakdown: a short untimed pass on ONE stream with events around every phase
    for i in range(min(steps, 20)):
        step(nprobe, i % N_BATCHES, 0)
    sync_all()
    ev_ph = ctx.read_timing()
    for c in ctxs:
        c.set_timing(0)
    return elapsed, ev, ev_ph, times


def groups_of(times, steps, Q):
    """the timed groups of a region as a field of the result: every group = exactly `steps` steps between synchronisations"""
    return {"groups": len(times), "steps_each": steps, "reported": "median group",
            "queries_per_s": [round(Q * steps / t, 1) for t in times],
            "min": round(Q * steps / max(times), 1), "max": round(Q * steps / min(times), 1)}


def roofline_of(scan_bytes, ev, traffic=None, kernel="k_scan", pair_rows=None, d=None, traffic_source=None):
    """the launch against both roofs: unique bytes / time against the HBM peak, and -- when `pair_rows` (rows x probing queries,
    summed over the probed lists) is given -- 2*d flops per (row, query) against the dense fp32 MFMA peak.  `bound` names the roof
    that gives the LONGER minimum time for this batch under the arithmetic the launched form really does (see below: the mixed
    form prefilters in bf16 and is always priced against HBM); the top-level achieved / peak / frac are that roof's, the other
    one is kept under its own key."""
    scan_ms = ev["scan_ms"] / max(ev["calls"], 1)
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    r = {
        "kernel": kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
        "traffic_over_algorithmic": round(traffic / scan_bytes, 3) if traffic and scan_bytes else None,
        "algorithmic_bytes_per_launch": int(scan_bytes), "kernel_ms_avg": round(scan_ms, 5), "launches": ev["calls"],
    }
    if pair_rows is not None and scan_ms > 0:
        flops = 2.0 * d * pair_rows
        tf = flops / (scan_ms * 1e-3) / 1e12
        r["mfma"] = {"algorithmic_flops_per_launch": int(flops), "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                     "queries_per_scanned_row": round(pair_rows * d * 4 / max(scan_bytes, 1), 2),
                     "min_ms_hbm": round(scan_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, 4),
                     "min_ms_mfma": round(flops / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3, 4)}
        # which roof binds: the tile / row-per-lane forms do every product on fp32 matrix instructions, so a batch whose flops take
        # longer than its bytes lives under the fp32-MFMA roof.  The MIXED form does most products of its hot lists on
        # v_mfma_f32_16x16x32_bf16 (16x the fp32 rate, exact fp32 chains only where a candidate is possible): its roof is HBM
        # whatever the algorithmic flop count says -- the fp32 figure stays in `mfma` as information only
        if "(mixed)" in str(kernel):
            r["mfma"]["note"] = ("informational: algorithmic flops against the fp32-MFMA peak; the mixed form computes most of them "
                                 "in bf16 behind a one-sided bound, so the launch is priced against HBM")
        elif r["mfma"]["min_ms_mfma"] > r["mfma"]["min_ms_hbm"]:
            r["hbm"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"]}
            r.update(bound="mfma", achieved=r["mfma"]["achieved"], peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=r["mfma"]["frac"])
    return r


def phases_of(ev_ph):
    c = max(ev_ph["calls"], 1)
    return {"coarse": round(ev_ph["coarse_ms"] / c, 4), "group": round(ev_ph["group_ms"] / c, 4),
            "scan": round(ev_ph["scan_ms"] / c, 4), "merge": round(ev_ph["merge_ms"] / c, 4),
            "note": "separate untimed pass with events around every phase"}


def _form_matches(form, kernel_name):
    """does a kernel symbol of the profile belong to the scan form qk_ctx_last_scan_kernel named?"""
    kn = kernel_name.replace(" ", "")
    if form is None:
        return "k_scan" in kn
    if "(mixed)" in form:
        return "k_scan_rl<" in kn and kn.split("(")[0].endswith(",true>")
    if form.startswith("k_scan_rl"):
        return "k_scan_rl<" in kn and kn.split("(")[0].endswith(",false>")
    return "k_scan<" in kn


def measured_traffic(args, nprobe, manifold, timeout=420, form=None):
    """HBM bytes per partition-scan launch, measured in THIS run: a rocprofv3 pass (--pmc FETCH_SIZE, kernel trace only: PMC
    collection gets its own process, MI355X_MICROARCH.md "HBM / rocprofv3") over a short replay of the same workload
    (`bench.py --traffic-probe`: same corpus, index, batches and nprobe; 32 synchronised searches, so that the form feedback
    settles as it did in the timed run); the launches counted are those of `form`, the form the timed region ran.  FETCH_SIZE is reported in KB and, on
    gfx950, tallies the 128-B requests of wide streaming reads at 64 B: x 1024 x 2.  None when rocprofv3 is not on the box,
    the pass fails, or QUAKE_BENCH_NO_PMC is set."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("QUAKE_BENCH_NO_PMC"):
        return None
    tmp = tempfile.mkdtemp(prefix="quake_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", tmp, "--", sys.executable,
           os.path.abspath(__file__), "--traffic-probe", "--nprobe", str(nprobe), "--nvec", str(args.nvec), "--dim", str(args.dim),
           "--nlist", str(args.nlist), "--batch", str(args.batch), "--k", str(args.k), "--metric", args.metric,
           "--sigma", str(args.sigma), "--niter", str(args.niter), "--manifold", str(manifold)]
    try:
        env = dict(os.environ, TMPDIR="/tmp", QUAKE_BENCH_NO_PMC="1")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            log("rocprofv3 traffic pass failed:", r.stderr[-400:])
            return None
        tot, n = {}, {}
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                # (the launches of the form that was TIMED: with form feedback on, the replay also runs the forms it compares)
                if row.get("Counter_Name") == "FETCH_SIZE" and _form_matches(form, row.get("Kernel_Name", "")):
                    kn = row["Kernel_Name"].split("(")[0]
                    tot[kn] = tot.get(kn, 0.0) + float(row["Counter_Value"])
                    n[kn] = n.get(kn, 0) + 1
        if not tot:
            return None
        kn = max(tot, key=tot.get)
        return {"bytes_per_launch": int(tot[kn] / n[kn] * 1024 * 2), "kernel": kn.replace("void ", ""), "launches": n[kn]}
    except Exception as e:  # (a profiler problem must never cost the bench line)
        log("rocprofv3 traffic pass skipped:", repr(e)[:200])
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_traffic_probe(ctx, dev, args):
    """`--traffic-probe`: the replay the rocprofv3 pass of measured_traffic() wraps -- build, then 8 searches at the given nprobe"""
    n, d, nlist, k, Q, metric = args.nvec, args.dim, args.nlist, args.k, args.batch, args.metric
    unit = metric == "ip"
    if args.manifold:
        x, basis = gen_manifold(n, d, seed=1, device=dev, latent=args.manifold)
        batches = [gen_manifold(Q, d, seed=2 + b, device=dev, latent=args.manifold, basis=basis)[0] for b in range(N_BATCHES)]
    else:
        x, cent_true = gen_mixture(n, d, nlist, seed=1, device=dev, sigma=args.sigma, unit=unit)
        batches = [gen_queries(Q, cent_true, seed=2 + b, device=dev, sigma=args.sigma, unit=unit) for b in range(N_BATCHES)]
    idx = build_single(ctx, dev, x, nlist, metric, args.niter, keep_host=False)
    del x
    for i in range(32):  # (synchronised: the form feedback reads one measurement per call)
        ctx.search(idx["parent"], idx["store"], batches[i % N_BATCHES], max(args.nprobe, 1), k, metric)
        torch.cuda.synchronize()


def committed_traffic(name, n, d, k, nprobe):
    """HBM bytes per k_scan launch from the committed rocprofv3 PMC pass of this command (profiles/<name>), if it matches
    the configuration that just ran; a PMC pass cannot run inside this process."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        pj = json.load(open(path))
        if pj.get("nvec") == n and pj.get("nprobe") == nprobe and pj.get("dim", 128) == d and pj.get("k", 10) == k:
            return pj.get("traffic_bytes_per_launch")
    except Exception:
        pass
    return None


def cpu_sgemm_search(qh, hc, hv, hi, ho, nprobe, k, metric, threads):
    """SURVEY 8(d) CPU leg (iii): batched_serial_scan as the reference's back end computes it (list_scanning.h:313-366 ->
    faiss::knn_L2sqr / knn_inner_product): queries grouped by probed partition (query_coordinator.cpp:707-721); a group of >= 20
    queries against its partition is ONE sgemm (torch.mm on the host = MKL) + the |x|^2 + |y|^2 - 2xy fix-up clamped at 0, a
    smaller group the direct form; k_max = min(k, n_p) best rows per (query, partition), merged per query.  The coarse step is
    the same thing over the centroids.  Not the checker (summation order is the BLAS's): a timed port of the reference's
    arithmetic speed.  Returns (ids [Q, k], dist [Q, k])."""
    torch.set_num_threads(int(threads))
    q = torch.from_numpy(qh)
    Q, d = q.shape
    cent = torch.from_numpy(hc)
    l2 = metric == "l2"

    def keys(a, b):  # smaller = better
        if not l2:
            return -(a @ b.T)
        if a.shape[0] >= 20:
            return ((a * a).sum(1, keepdim=True) + (b * b).sum(1)[None, :] - 2.0 * (a @ b.T)).clamp_(min=0)
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(2)

    pids = torch.topk(keys(q, cent), min(nprobe, cent.shape[0]), dim=1, largest=False).indices  # [Q, P]
    flat = pids.reshape(-1)
    order = torch.argsort(flat, stable=True)
    uniq, cnt = torch.unique_consecutive(flat[order], return_counts=True)
    qidx = torch.div(order, pids.shape[1], rounding_mode="floor")
    best_k = torch.full((Q, pids.shape[1] * k), float("inf"))
    best_i = torch.full((Q, pids.shape[1] * k), -1, dtype=torch.int64)
    fill = torch.zeros(Q, dtype=torch.int64)
    vt, it = torch.from_numpy(hv), torch.from_numpy(hi)
    pos = 0
    for p, c in zip(uniq.tolist(), cnt.tolist()):
        rows = qidx[pos:pos + c]
        pos += c
        a, b = int(ho[p]), int(ho[p + 1])
        if b == a:
            continue
        kk = min(k, b - a)
        v, j = torch.topk(keys(q[rows], vt[a:b]), kk, dim=1, largest=False)
        col = fill[rows][:, None] + torch.arange(kk)[None, :]
        best_k[rows[:, None], col] = v
        best_i[rows[:, None], col] = it[a:b][j]
        fill[rows] += kk
    v, j = torch.topk(best_k, k, dim=1, largest=False)
    ids = torch.gather(best_i, 1, j)
    dist = torch.sqrt(v) if l2 else -v
    return ids.numpy(), dist.numpy()


def run_single_workload(ctx, dev, args, name, sigma, fixed_nprobe, steps, warmup, settle, cpu_seconds, traffic_file=None,
                        manifold=0, sweep_nprobes=(), with_aps=None):
    """Build, sweep nprobe, time, verify against the oracle.  Returns the result dict of one single-GPU workload.
    manifold > 0: the low-intrinsic-dimension corpus (gen_manifold, latent dimension `manifold`) instead of the mixture."""
    n, d, nlist, k, Q, metric = args.nvec, args.dim, args.nlist, args.k, args.batch, args.metric
    unit = metric == "ip"
    t0 = time.time()
    sift = None
    if getattr(args, "sift_dir", "") and name == "headline":
        # the metric's NAMED dataset, verbatim, when it is on the box: SIFT1M base / query files, nlist from --nlist
        base, query, gt_file = load_sift_dir(args.sift_dir)
        n, d = base.shape
        x = torch.from_numpy(base).to(dev)
        nb_avail = query.shape[0] // Q
        if nb_avail < 1:
            raise SystemExit(f"--sift-dir: {query.shape[0]} queries, fewer than one batch of {Q}")
        sift = {"query": torch.from_numpy(query).to(dev), "gt": gt_file}
        desc = f"read from {args.sift_dir}"
    elif manifold:
        x, basis = gen_manifold(n, d, seed=1, device=dev, latent=manifold)
        desc = f"x = zA + noise, latent dimension {manifold}"
    else:
        x, cent_true = gen_mixture(n, d, nlist, seed=1, device=dev, sigma=sigma, unit=unit)
        desc = f"Gaussian mixture (sigma {sigma})"
    torch.cuda.synchronize()
    log(f"[{name}] generated {n}x{d} ({desc}) in {time.time() - t0:.1f}s")
    want_cpu = not args.no_cpu
    idx = build_single(ctx, dev, x, nlist, metric, args.niter, keep_host=want_cpu)
    parent, store = idx["parent"], idx["store"]
    if sift is not None:
        nq = sift["query"].shape[0] // Q
        batches = [sift["query"][(b % nq) * Q:(b % nq + 1) * Q].contiguous() for b in range(N_BATCHES)]
    elif manifold:
        batches = [gen_manifold(Q, d, seed=2 + b, device=dev, latent=manifold, basis=basis)[0] for b in range(N_BATCHES)]
    else:
        batches = [gen_queries(Q, cent_true, seed=2 + b, device=dev, sigma=sigma, unit=unit) for b in range(N_BATCHES)]
    t0 = time.time()
    gts = [brute_force_topk(q, x, k, metric=metric)[0] for q in batches]
    torch.cuda.synchronize()
    log(f"[{name}] brute-force ground truth of {N_BATCHES} batches {time.time() - t0:.2f}s")
    if sift is not None and sift["gt"] is not None and sift["gt"].shape[1] >= k:
        # the dataset's own ground truth file against the brute force of this run (ids as sets: ties may be ordered differently)
        g0 = torch.from_numpy(sift["gt"][:Q, :k].astype(np.int64)).to(dev)
        log(f"[{name}] agreement of the brute force with the dataset's ground-truth file, batch 0: {recall_at_k(gts[0], g0, k):.4f}")
    del x
    # (extra measurement) `inflight` batches at a time: step i runs on context i % inflight -- own HIP stream, own output buffers, the same
    # stores -- so the small kernels of one batch (prep, nearest centroid, group + seed, merge) run under the partition
    # scan of another.  (The extra contexts use the library's private non-blocking streams.)
    from quake_amd.capi import Context
    inflight = max(1, int(args.inflight))
    ctxs = [ctx] + [Context(dev.index) for _ in range(inflight - 1)]
    outs = [(torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev))
            for _ in range(inflight)]

    def step(nprobe, b, slot=0):
        return ctxs[slot].search(parent, store, batches[b], nprobe, k, metric, out=outs[slot])

    nprobe, recall, sweep = pick_nprobe(step, batches, gts, k, args.recall_target, fixed_nprobe,
                                        lambda ri, b: recall_at_k(ri, gts[b], k))
    log(f"[{name}] nprobe={nprobe} recall@{k}={recall:.4f} sweep={sweep}")
    ctx.set_timing(1)
    _, _, tinfo = ctx.search(parent, store, batches[0], nprobe, k, metric, timing=True)
    log(f"[{name}] phases (ms):", {kk: round(v, 4) if isinstance(v, float) else v for kk, v in tinfo.items()})
    # the timed region of the contract: one batch at a time on one stream
    elapsed, ev, ev_ph, gtimes = timed_region(ctx, step, nprobe, steps, warmup, settle, None, dev, ctxs=[ctx])
    piped = None
    if inflight > 1:  # the same steps with `inflight` batches in flight, reported beside the headline figure (never `value`)
        n2 = max(steps // 2, 10)
        e2, ev2, _, _ = timed_region(ctx, step, nprobe, n2, min(warmup, 10), 0, None, dev, ctxs=ctxs, groups=3)
        piped = {"batches_in_flight": inflight, "value": round(Q * n2 / e2, 1), "unit": "queries/s",
                 "ms_per_step": round(1e3 * e2 / n2, 4), "steps": n2,
                 "scan_kernel_ms_avg": round(ev2["scan_ms"] / max(ev2["calls"], 1), 5),
                 "note": f"step i runs on HIP stream i % {inflight} (own context and output buffers, same index): the small kernels "
                         "of one batch run under the partition scan of another; every step is still one complete qk_search"}
    # algorithmic bytes: mean over the rotated batches (each launch reports the unique rows it scanned)
    scan_bytes = 0
    ctx.set_timing(1)
    for b in range(N_BATCHES):
        scan_bytes += int(ctx.search(parent, store, batches[b], nprobe, k, metric, timing=True)[2]["scan_bytes"])
    ctx.set_timing(0)
    scan_bytes //= N_BATCHES
    # rows x probing queries over the probed lists (mean over the rotated batches): the flops side of the same launch
    scan_kernel = ctx.last_scan_kernel()  # (before the coarse calls below: they are launches of their own)
    cnt_t = torch.as_tensor(idx["counts"], device=dev)

    def pair_rows_of(npb):
        return int(sum(int(cnt_t[ctx.coarse(parent, batches[b], npb, metric)[0]].sum().item()) for b in range(N_BATCHES)) // N_BATCHES)

    pair_rows = pair_rows_of(nprobe)
    # HBM traffic of the scan launch: measured in this run (a rocprofv3 PMC pass over a replay of the workload) when the profiler is
    # on the box, else the committed pass of the same configuration (profiles/), marked as such
    traffic, traffic_source = None, None
    if not args.no_pmc:
        mt = measured_traffic(args, nprobe, manifold, form=scan_kernel)
        if mt is not None:
            traffic, traffic_source = mt["bytes_per_launch"], f"rocprofv3 --pmc FETCH_SIZE in this run ({mt['kernel']}, {mt['launches']} launches; x1024 x2: KB, gfx950 wide-read tally)"
    if traffic is None and traffic_file:
        traffic = committed_traffic(traffic_file, n, d, k, nprobe)
        traffic_source = f"committed: profiles/{traffic_file}" if traffic is not None else None
    # the caller-visible rate through the reference's CPU-tensor API: host buffers in and out (QK_MEM_HOST: the queries cross
    # PCIe, the answers come back, one synchronisation per call) -- reported beside `value`, never as it
    host_api = None
    if not args.no_host_api:
        hq = [b.cpu().numpy() for b in batches]
        for i in range(20):
            ctx.search(parent, store, hq[i % N_BATCHES], nprobe, k, metric)
        nh = max(20, min(steps, 100))
        t1 = time.perf_counter()
        for i in range(nh):
            ctx.search(parent, store, hq[i % N_BATCHES], nprobe, k, metric)
        dt = time.perf_counter() - t1
        host_api = {"value": round(Q * nh / dt, 1), "unit": "queries/s", "ms_per_step": round(1e3 * dt / nh, 4), "steps": nh,
                    "note": "numpy queries in, numpy ids / distances out (QK_MEM_HOST): H2D of the batch, the same qk_search, D2H of the "
                            "answers and a synchronisation per call -- what a caller of the reference's CPU-tensor API sees"}
    # the shared-list regime on the same index: fixed nprobe values, each with its own roofline (short timed regions)
    sweep_res = []
    for npb in sweep_nprobes:
        if npb == nprobe:
            continue
        e_s, ev_s, _, _ = timed_region(ctx, step, npb, 30, 5, 20, None, dev, ctxs=[ctx], groups=5)
        ctx.set_timing(1)
        sb = int(sum(int(ctx.search(parent, store, batches[b], npb, k, metric, timing=True)[2]["scan_bytes"]) for b in range(N_BATCHES)) // N_BATCHES)
        ctx.set_timing(0)
        kern = ctx.last_scan_kernel()
        ri = step(npb, 0)[0]
        torch.cuda.synchronize()
        mt = None if args.no_pmc else measured_traffic(args, npb, manifold, form=kern)
        sweep_res.append({"nprobe": npb, "value": round(Q * 30 / e_s, 1), "unit": "queries/s", "ms_per_step": round(1e3 * e_s / 30, 4),
                          "steps": 30, "recall_at_k": round(recall_at_k(ri, gts[0], k), 4),
                          "roofline": roofline_of(sb, ev_s, mt["bytes_per_launch"] if mt else None, kernel=kern, pair_rows=pair_rows_of(npb), d=d,
                                                  traffic_source=f"rocprofv3 --pmc FETCH_SIZE in this run ({mt['kernel']}, {mt['launches']} launches)" if mt else None)})
    res = {
        "value": round(Q * steps / elapsed, 1), "unit": "queries/s", "ms_per_step": round(1e3 * elapsed / steps, 4),
        "steps": steps, "warmup": warmup, "timed_groups": groups_of(gtimes, steps, Q),
        "config": {
            "workload": (f"{os.path.basename(os.path.normpath(args.sift_dir))} {n} x {d} f32 {metric.upper()} ({desc}), " if sift is not None else
                         f"Synthetic {n // 1_000_000}M x {d} f32 {metric.upper()} {'unit-norm ' if unit else ''}{desc}, ")
                        + f"nlist={nlist}, batch={Q} queries, k={k}, nprobe={nprobe}",
            "data": "dataset files (--sift-dir)" if sift is not None else "synthetic",
            "nvec": n, "dim": d, "metric_type": metric, "nlist": nlist, "batch": Q, "k": k, "nprobe": nprobe,
            "sigma": None if manifold else sigma, "latent_dim": manifold or None,
            "recall_at_k": round(recall, 4), "recall_sweep": sweep, "settle_steps": max(settle, 0),
            "query_batches_rotated": N_BATCHES,
        },
        "batches_in_flight": piped,
        "roofline": roofline_of(scan_bytes, ev, traffic, kernel=scan_kernel, pair_rows=pair_rows, d=d, traffic_source=traffic_source),
        "host_api": host_api,
        "phases_ms": phases_of(ev_ph),
        "build": {"kmeans_s": round(idx["kmeans_s"], 2), "niter": args.niter, "last_iteration": idx.get("kmeans_kernels")},
    }
    if sweep_res:
        res["nprobe_sweep"] = sweep_res
    # the recall-target search (adaptive partition scanning, SearchParams.recall_target > 0) on the same index: rounds on the
    # device, the reference's defaults (initial_search_fraction 0.02, recompute_threshold 0.001, precomputed table)
    aps_res = None
    if (bool(sweep_nprobes) if with_aps is None else with_aps) and k <= 32:
        rt = 0.9
        for _ in range(8):  # (the form feedback compares its forms per round shape first)
            ai, ad, an, atm = ctx.search_aps(parent, store, batches[0], k, metric, rt, timing=True)
        torch.cuda.synchronize()
        tt = []
        for gidx in range(5):
            t1 = time.perf_counter()
            for i in range(10):
                ctx.search_aps(parent, store, batches[i % N_BATCHES], k, metric, rt)
            torch.cuda.synchronize()
            tt.append((time.perf_counter() - t1) / 10)
        tt.sort()
        ai, ad, an = ctx.search_aps(parent, store, batches[0], k, metric, rt)
        torch.cuda.synchronize()
        npe = max(1, int(round(float(an.float().mean().item()))))  # the fixed search that scans as many lists per query
        for i in range(8):
            step(npe, i % N_BATCHES)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(20):
            step(npe, i % N_BATCHES)
        torch.cuda.synchronize()
        fixed_qps = Q * 20 / (time.perf_counter() - t1)
        aps_res = {"recall_target": rt, "value": round(Q / tt[2], 1), "unit": "queries/s", "ms_per_step": round(1e3 * tt[2], 4),
                   "steps": "median of 5 groups of 10", "recall_at_k": round(recall_at_k(ai, gts[0], k), 4),
                   "partitions_scanned_mean": round(float(an.float().mean().item()), 2), "partitions_scanned_max": int(an.max().item()),
                   "rounds": int(atm["n_items"]), "candidates": max(int(np.float32(nlist) * np.float32(0.02)), 1),
                   "fixed_nprobe_of_equal_work": {"nprobe": npe, "value": round(fixed_qps, 1), "unit": "queries/s"},
                   "note": "qk_search_aps: results and partitions visited are those of the reference's sequential walk "
                           "(query_coordinator.cpp:529-579)"}
        res["recall_target_search"] = aps_res
    # ---- parity + CPU baseline: the oracle port on the host cores (test infrastructure used as checker / timed port) ----
    if want_cpu:
        import oracle as O
        hv, hi, ho, hc = idx["host"]
        qh = batches[0].cpu().numpy()
        gi0, gd0 = step(nprobe, 0)
        torch.cuda.synchronize()
        gi0, gd0 = gi0.cpu().numpy(), gd0.cpu().numpy()
        cores = O.max_threads()
        eff_cores = round(O.effective_cores(cores), 1)  # what the host really gives this process (shared / sandboxed boxes)

        def time_cpu(batched, budget, threads, nq, fast=1):
            t, nn, reps, ids, dist = 0.0, 0, 0, None, None
            while reps == 0 or (t < budget and reps < 10000):
                t1 = time.perf_counter()
                ids, dist = O.search(qh[:nq], hc, hv, hi, ho, nprobe, k, metric, batched_scan=batched, num_threads=threads,
                                     fast=fast)
                t += time.perf_counter() - t1
                nn += nq
                reps += 1
            return nn / t, nn, reps, t, ids, dist

        # the thread count of the timed legs is CHOSEN BY MEASUREMENT: the boxes report 128 hardware threads and give the process 15-36
        # cores' worth (effective_cores), where 128 OpenMP threads run slower than 32 -- candidates around the measured figure and the
        # reported one, a short run each, the fastest is used (cores_reported says what the box claimed)
        cores_reported = cores
        cand_t = sorted({max(1, int(round(eff_cores))), max(1, min(cores, 2 * int(round(eff_cores)))), max(1, cores // 2), cores})
        probe = {t_: time_cpu(False, min(1.0, cpu_seconds * 0.03), t_, min(Q, 256), fast=2)[0] for t_ in cand_t}
        cores = max(probe, key=probe.get)
        log(f"[{name}] cpu legs: threads {cores} of {cores_reported} reported (probe q/s: { {t_: round(v) for t_, v in probe.items()} })")
        qps_b, n_b, reps_b, t_b, ids_b, dist_b = time_cpu(True, cpu_seconds * 0.35, cores, Q)
        # parity at the bench size: the canonical (batched, expanded-form) oracle must give the SAME bits as the GPU
        same_ids = float((ids_b == gi0).mean())
        same_dist = float((dist_b.view(np.uint32) == gd0.view(np.uint32)).mean())
        if same_ids != 1.0 or same_dist != 1.0:
            raise SystemExit(f"[{name}] PARITY FAILURE at the bench size: ids equal {same_ids:.6f}, distance bits equal "
                             f"{same_dist:.6f} (HIP qk_search vs oracle batched_serial_scan)")
        # serial_scan timed with FAISS's way of computing a row (SIMD over the dimensions, lane partial sums: fast=2) -- not the
        # canonical summation order, so its ids may differ on near-ties; it is the timed port, never the checker
        qps_s, n_s, reps_s, t_s, ids_s, _ = time_cpu(False, cpu_seconds * 0.35, cores, Q, fast=2)
        n1 = max(8, min(Q, int(qps_s / max(cores, 1) * cpu_seconds * 0.3) or 8))  # a few seconds on one thread
        qps_1, _, _, t_1, _, _ = time_cpu(False, 0.0, 1, n1, fast=2)
        # leg (iii): the batched path at the reference's arithmetic speed -- per-partition sgemm (MKL through torch.mm) on all
        # threads and on one; ids compared with the GPU's under the near-tie rule (a BLAS sums in its own order)
        nt0 = torch.get_num_threads()
        sg = {}
        for label, th, budget in (("all", cores, cpu_seconds * 0.15), ("one", 1, cpu_seconds * 0.15)):
            t, reps, ids_g, dist_g = 0.0, 0, None, None
            nq_g = Q if label == "all" else max(32, Q // 8)
            while reps == 0 or (t < budget and reps < 1000):
                t1 = time.perf_counter()
                ids_g, dist_g = cpu_sgemm_search(qh[:nq_g], hc, hv, hi, ho, nprobe, k, metric, th)
                t += time.perf_counter() - t1
                reps += 1
            sg[label] = (nq_g * reps / t, ids_g, dist_g, nq_g, reps, t)
        torch.set_num_threads(nt0)
        qps_g, ids_g, dist_g = sg["all"][0], sg["all"][1], sg["all"][2]
        agree = ids_g == gi0
        with np.errstate(invalid="ignore"):
            dist_ok = (np.abs(dist_g - gd0) <= 1e-4 * np.maximum(1.0, np.abs(gd0))) | (dist_g == gd0)
        best_batched = qps_b > qps_s
        best_cpu = max(qps_b, qps_s, qps_g)
        res["cpu_baseline"] = {
            "value": round(best_cpu, 1), "unit": "queries/s", "cores": cores, "cores_reported_by_the_box": cores_reported,
            "thread_count_probe_qps": {str(t_): round(v, 1) for t_, v in probe.items()}, "kind": "port",
            "fastest_leg": "batched_sgemm" if best_cpu == qps_g else ("batched_scan" if best_batched else "serial_scan"),
            "batched_sgemm_qps": round(qps_g, 1), "batched_sgemm_qps_one_thread": round(sg["one"][0], 1),
            "batched_sgemm_sample": f"{sg['all'][3]} queries x {sg['all'][4]} on {cores} torch threads in {sg['all'][5]:.1f}s; one "
                                    f"thread: {sg['one'][3]} queries x {sg['one'][4]} in {sg['one'][5]:.1f}s (torch.mm = MKL sgemm per "
                                    "partition group of >= 20 queries, direct form below: list_scanning.h:335-338 -> faiss::knn_L2sqr)",
            "batched_sgemm_ids_equal_to_gpu_frac": round(float(agree.mean()), 5),
            "batched_sgemm_dist_within_1e-4_frac": round(float(dist_ok.mean()), 5),
            "sample": f"(oracle legs) the {Q}-query bench batch 0 replayed {reps_b if best_batched else reps_s}x, same index/nprobe/k, oracle "
                      f"search() = coarse + {'batched_serial_scan' if best_batched else 'serial_scan'} semantics on {cores} "
                      f"threads, {t_b if best_batched else t_s:.1f}s (the faster of the reference's two scan variants); "
                      f"single thread: {n1} queries in {t_1:.1f}s.  NOTE: the box reports {cores_reported} threads and delivers "
                      f"{eff_cores} cores' worth of arithmetic (effective_cores_measured); {cores} threads is the fastest of the probed "
                      f"counts (thread_count_probe_qps) and is {max(qps_b, qps_s) / max(qps_1, 1e-9):.1f}x one thread.  `value` is the "
                      f"fastest of the three legs (fastest_leg)",
            "serial_scan_qps": round(qps_s, 1), "batched_scan_qps": round(qps_b, 1),
            "single_thread_qps": round(qps_1, 1), "threads_speedup": round(max(qps_b, qps_s) / max(qps_1, 1e-9), 1),
            "effective_cores_measured": eff_cores,
            "ids_equal_to_gpu_frac": same_ids, "distance_bits_equal_to_gpu_frac": same_dist,
            # (NOT a parity figure: the timed serial leg sums like FAISS's AVX2 row loop -- 8 lane partial sums -- which is not the
            #  canonical order, so a near-tie may land the other way round; the checker is the batched oracle above)
            "timed_serial_leg_ids_agree_frac": round(float((ids_s == gi0).mean()), 5),
        }
        res["speedup_vs_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        if aps_res is not None:  # in-run parity of the recall-target search: the oracle's walk over a sample of the batch
            nqa = 128
            oi, od, on = O.search_aps(qh[:nqa], hc, hv, hi, ho, k, metric, aps_res["recall_target"], expanded=True, num_threads=cores)
            ga, gn = ai[:nqa].cpu().numpy(), an[:nqa].cpu().numpy()
            gdd = ad[:nqa].cpu().numpy()
            okp = bool(np.array_equal(ga, oi) and np.array_equal(gn, on) and np.array_equal(gdd.view(np.uint32), od.view(np.uint32)))
            aps_res["parity_sample"] = f"{nqa} queries vs the oracle's walk: ids, distance bits and partitions visited equal = {okp}"
            if not okp:
                raise SystemExit(f"[{name}] PARITY FAILURE: qk_search_aps differs from the oracle's walk on the bench batch")
        if inflight > 1:  # the other stream's answer for the same batch: the same bits
            gi1, gd1 = step(nprobe, 0, 1)
            ctxs[1].synchronize()
            if not (np.array_equal(gi1.cpu().numpy(), gi0) and np.array_equal(gd1.cpu().numpy().view(np.uint32), gd0.view(np.uint32))):
                raise SystemExit(f"[{name}] PARITY FAILURE: the second stream's answer differs from the first")
    for c in ctxs[1:]:
        c.close()
    store.close()
    parent.close()
    del idx, batches, gts
    torch.cuda.empty_cache()
    return res


def run_configs0(ctx, dev, args):
    """BASELINE.json configs[0] shape on S-SIFT: 1M x 128, nlist=1024, nprobe=10, k=10, L2, batch = 1."""
    n, d, nlist, nprobe, k, nq = 1_000_000, 128, 1024, 10, 10, 1000
    x, cent = gen_ssift(n, dev, seed=1234)
    q, _ = gen_ssift(nq, dev, seed=4321, cent=cent)
    idx = build_single(ctx, dev, x, nlist, "l2", args.niter, keep_host=not args.no_cpu)
    parent, store = idx["parent"], idx["store"]
    gt, _ = brute_force_topk(q, x, k)
    del x
    out_i = torch.empty((1, k), dtype=torch.int64, device=dev)
    out_d = torch.empty((1, k), dtype=torch.float32, device=dev)
    qs = [q[i:i + 1].contiguous() for i in range(nq)]
    ids_all = torch.empty((nq, k), dtype=torch.int64, device=dev)
    for i in range(nq):  # warm-up pass, also the answers
        ctx.search(parent, store, qs[i], nprobe, k, "l2", out=(ids_all[i:i + 1], out_d))
    torch.cuda.synchronize()
    recall = recall_at_k(ids_all, gt, k)
    # (a) back-to-back single-query searches, device buffers, one synchronisation at the end: launch throughput
    t0 = time.perf_counter()
    for i in range(nq):
        ctx.search(parent, store, qs[i], nprobe, k, "l2", out=(out_i, out_d))
    torch.cuda.synchronize()
    t_pipe = time.perf_counter() - t0
    # (a') the same searches spread over four contexts (own HIP stream each, the same stores): what several client threads of the
    # reference's batch = 1 traffic amount to -- single-query launches of ~50 workgroups run side by side on the 256 CUs
    from quake_amd.capi import Context
    NS = 4
    cs = [ctx] + [Context(dev.index) for _ in range(NS - 1)]
    ids_ms = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dist_ms = torch.empty((nq, k), dtype=torch.float32, device=dev)
    for i in range(4 * NS):
        cs[i % NS].search(parent, store, qs[i], nprobe, k, "l2", out=(ids_ms[i:i + 1], dist_ms[i:i + 1]))
    for c in cs:
        c.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(nq):
        cs[i % NS].search(parent, store, qs[i], nprobe, k, "l2", out=(ids_ms[i:i + 1], dist_ms[i:i + 1]))
    for c in cs:
        c.synchronize()
    torch.cuda.synchronize()
    t_ms = time.perf_counter() - t0
    if not torch.equal(ids_ms, ids_all):
        raise SystemExit("[configs0] PARITY FAILURE: the four-stream run's ids differ from the one-stream run's")
    for c in cs[1:]:
        c.close()
    # (b) one query at a time, synchronised: latency as a caller sees it
    lat = []
    for i in range(min(nq, 200)):
        t1 = time.perf_counter()
        ctx.search(parent, store, qs[i], nprobe, k, "l2", out=(out_i, out_d))
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    lat = np.array(lat) * 1e6
    # (c) whole 1000-query batch in one call
    ob_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    ob_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
    for _ in range(3):
        ctx.search(parent, store, q, nprobe, k, "l2", out=(ob_i, ob_d))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.search(parent, store, q, nprobe, k, "l2", out=(ob_i, ob_d))
    torch.cuda.synchronize()
    t_batch = (time.perf_counter() - t0) / 20
    res = {
        "config": {"workload": "S-SIFT (SIFT1M stand-in) 1M x 128 integer-valued f32 L2, nlist=1024, nprobe=10, k=10, batch=1 "
                               "(BASELINE.json configs[0])", "recall_at_k": round(recall, 4), "queries": nq},
        "value": round(nq / t_pipe, 1), "unit": "queries/s",
        "note": "batch=1 searches issued back to back on one stream (device buffers), one synchronisation at the end",
        "latency_us_synchronised": {"p50": round(float(np.percentile(lat, 50)), 1), "p99": round(float(np.percentile(lat, 99)), 1)},
        "batch_1000_qps": round(nq / t_batch, 1),
        "four_streams": {"value": round(nq / t_ms, 1), "unit": "queries/s",
                         "note": "the same batch = 1 searches round-robin over four contexts (one HIP stream each), ids equal to the one-stream run's"},
    }
    if not args.no_cpu:
        import oracle as O
        hv, hi, ho, hc = idx["host"]
        qh = q.cpu().numpy()
        # the reference default: serial_scan, one query per call, ONE thread
        t0 = time.perf_counter()
        cpu_ids = np.empty((nq, k), np.int64)
        for i in range(nq):
            cpu_ids[i], _ = O.search(qh[i:i + 1], hc, hv, hi, ho, nprobe, k, "l2", batched_scan=False, num_threads=1, fast=2)
        t_cpu = time.perf_counter() - t0
        same = float((cpu_ids == ids_all.cpu().numpy()).mean())
        same_b = float((cpu_ids == ob_i.cpu().numpy()).mean())
        if same != 1.0 or same_b != 1.0:
            raise SystemExit(f"[configs0] PARITY FAILURE: batch=1 ids equal {same:.6f}, batch=1000 ids equal {same_b:.6f} "
                             f"(HIP vs oracle serial_scan, integer data)")
        res["cpu_baseline"] = {"value": round(nq / t_cpu, 1), "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": f"the same {nq} queries, one search() call per query, serial_scan semantics, 1 thread, "
                                         f"{t_cpu:.1f}s", "ids_equal_to_gpu_frac": same}
        res["speedup_vs_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
    store.close()
    parent.close()
    torch.cuda.empty_cache()
    return res


# ---- N > 1: BASELINE.json configs[3] shape --------------------------------------------------------------------------------
def _all_to_all(dist, out, inp, out_splits=None, in_splits=None):
    """all_to_all_single; staged through the host under gloo (QUAKE_BENCH_BACKEND=gloo: two ranks on one GPU, functional check)"""
    if str(dist.get_backend()).lower() == "gloo" and inp.is_cuda:
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(ho, inp.cpu(), out_splits, in_splits)
        out.copy_(ho)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits)


def _all_gather(dist, tensors, t):
    if str(dist.get_backend()).lower() == "gloo" and t.is_cuda:
        hl = [torch.empty(x.shape, dtype=x.dtype) for x in tensors]
        dist.all_gather(hl, t.cpu())
        for a, b in zip(tensors, hl):
            a.copy_(b)
    else:
        dist.all_gather(tensors, t)


def _all_reduce(dist, t):
    if str(dist.get_backend()).lower() == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)


def run_sharded(ctx, dev, args, dist, rank, world):
    from quake_amd.capi import Store
    from quake_amd.sharded import GpuEngine, ShardedIndex, sharded_kmeans
    n, d, k, metric = args.nvec_sharded, args.dim, args.k, args.metric
    nlist_g = args.nlist_sharded * world
    Q = args.batch_sharded * world
    unit = metric == "ip"
    # the SAME mixture on every rank (nlist_g components), a different draw per rank
    g0 = torch.Generator(device=dev).manual_seed(1)
    cent_true = torch.randn(nlist_g, d, generator=g0, device=dev)
    x, _ = gen_mixture(n, d, nlist_g, seed=1000 + rank, device=dev, unit=unit, cent=cent_true)
    id_base = rank * n
    torch.cuda.synchronize()
    t0 = time.time()
    centroids, assign = sharded_kmeans(ctx, dist, x, nlist_g, metric, niter=args.niter, seed=1234, rank=rank, world=world)
    torch.cuda.synchronize()
    t_kmeans = time.time() - t0
    log(f"sharded k-means over {world} ranks: {n * world} vectors, nlist={nlist_g}, niter={args.niter}: {t_kmeans:.2f}s")
    # route every vector to the owner of its list (list p lives on rank p % world): one all-to-all of rows + ids
    t0 = time.time()
    owner = assign % world
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world)
    recv_counts = torch.empty_like(send_counts)
    _all_to_all(dist, recv_counts, send_counts)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    xs, as_, is_ = x[order].contiguous(), assign[order].contiguous(), (order + id_base).contiguous()
    del x, order, owner
    nr = int(sum(rc))
    xr = torch.empty((nr, d), device=dev)
    ar = torch.empty((nr,), dtype=torch.int64, device=dev)
    ir = torch.empty((nr,), dtype=torch.int64, device=dev)
    _all_to_all(dist, xr, xs, rc, sc)
    _all_to_all(dist, ar, as_, rc, sc)
    _all_to_all(dist, ir, is_, rc, sc)
    del xs, as_, is_
    o2 = torch.argsort(ar, stable=True)
    counts = torch.bincount(ar, minlength=nlist_g).cpu().numpy().astype(np.int64)
    offsets = np.zeros(nlist_g + 1, np.int64)
    offsets[1:] = np.cumsum(counts)
    x_local, ids_local = xr[o2].contiguous(), ir[o2].contiguous()
    del xr, ar, ir, o2
    store = Store(ctx, d)
    store.build_csr(offsets, ids_local, x_local)
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist_g], np.int64), torch.arange(nlist_g, device=dev), centroids.contiguous())
    torch.cuda.synchronize()
    own = counts[counts > 0]
    log(f"rank 0 holds {int(counts.sum())} vectors in {len(own)} lists after the exchange ({time.time() - t0:.1f}s), "
        f"arena {store.device_bytes() / 1e9:.2f} GB")
    batches = [gen_queries(Q, cent_true, seed=2 + b, device=dev, unit=unit) for b in range(N_BATCHES)]
    per = Q // world
    gts = []
    for q in batches:  # exact ground truth: local brute force, all-gather, merge
        gi, gd2 = brute_force_topk(q, x_local, k, metric=metric)
        gi = torch.where(gi >= 0, ids_local[gi.clamp(min=0)], gi)
        gl_i = [torch.empty_like(gi) for _ in range(world)]
        gl_d = [torch.empty_like(gd2) for _ in range(world)]
        _all_gather(dist, gl_i, gi)
        _all_gather(dist, gl_d, gd2)
        ci, cd = torch.cat(gl_i, 1), torch.cat(gl_d, 1)
        _, j = torch.topk(cd, k, dim=1, largest=False)
        gts.append(torch.gather(ci, 1, j)[rank * per:(rank + 1) * per])
    del x_local
    # scratch of the LOCAL scan (every rank scans the whole batch against the lists it owns: [Q, k]); the merged answer of
    # this rank's slice comes back from search() as [Q / N, k]
    out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
    out_d = torch.empty((Q, k), dtype=torch.float32, device=dev)
    sharded = ShardedIndex(GpuEngine(ctx, parent, store, metric), dist, world, rank, result="owner")

    def step(nprobe, b, slot=0):  # (one batch in flight: the collectives order the steps)
        return sharded.search(batches[b], nprobe, k, out=(out_i, out_d))

    def rec(ri, b):
        r = torch.tensor([recall_at_k(ri, gts[b], k)], device=dev, dtype=torch.float64)
        _all_reduce(dist, r)
        return r.item() / world

    nprobe, recall, sweep = pick_nprobe(step, batches, gts, k, args.recall_target, args.nprobe, rec)
    log(f"nprobe={nprobe} recall@{k}={recall:.4f} sweep={sweep}")
    elapsed, ev, ev_ph, gtimes = timed_region(ctx, step, nprobe, args.steps, args.warmup, args.settle, dist, dev)
    ctx.set_timing(1)
    sb = torch.tensor([float(sharded.engine.last_scan_bytes(batches[0], nprobe, k))], device=dev, dtype=torch.float64)
    ctx.set_timing(0)
    # what makes the first run on real hardware self-describing: every rank's own roofline line (its scan kernel's mean duration from
    # its own HIP events, its own unique bytes), the device each rank sits on (N distinct PCI bus ids = N GPUs really took part),
    # and the bytes of the two collectives of a step
    my_roof = roofline_of(int(sb.item()), ev, kernel=ctx.last_scan_kernel())
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device_index": dev.index, "device": props.name,
          "pci_bus_id": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', -1):02x}:{getattr(props, 'pci_device_id', 0):02x}",
          "uuid": str(getattr(props, "uuid", "")), "vectors": int(counts.sum()), "lists": int(len(own)),
          "roofline": my_roof, "phases_ms": phases_of(ev_ph)}
    seen = [None] * world
    if world > 1:
        try:
            dist.all_gather_object(seen, me)
        except Exception as e:  # (the description of the run must not cost the run its line: the timed region is over)
            log(f"per-rank description not gathered: {type(e).__name__}: {e}")
            seen = [me]
    else:
        seen = [me]
    kk = min(nprobe, nlist_g)
    from quake_amd.sharded import topk_block_bytes
    exchange = {
        "all_gather_list_numbers": {"send_bytes_per_rank": per * kk * 8, "recv_bytes_per_rank": Q * kk * 8},
        "all_to_all_topk": {"send_bytes_per_rank": world * topk_block_bytes(per, k), "recv_bytes_per_rank": world * topk_block_bytes(per, k),
                            "record": "12 bytes per entry (int64 id + float32 key), block j = the results for the queries rank j owns"},
        "collectives_per_step": 2,
    }
    return {
        "value": round(Q * args.steps / elapsed, 1), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "timed_groups": groups_of(gtimes, args.steps, Q),
        "per_rank": seen, "exchange": exchange,
        "rccl_ranks_seen": {"backend": str(dist.get_backend()), "ranks": world,
                            "distinct_devices": len({(r_["pci_bus_id"], r_["uuid"], r_["device_index"]) for r_ in seen}),
                            "ranks_described": len(seen)},
        "config": {
            "workload": f"Synthetic {n * world // 1_000_000}M x {d} f32 {metric.upper()} Gaussian mixture, nlist={nlist_g} "
                        f"(one k-means over all ranks), lists sharded by number over {world} ranks, batch={Q} queries, k={k}, "
                        f"nprobe={nprobe} (BASELINE.json configs[3] shape: 12.5M vectors, 8192 lists and 512 queries per GPU)",
            "nvec_per_gpu": n, "dim": d, "metric_type": metric, "nlist": nlist_g, "batch": Q, "k": k, "nprobe": nprobe,
            "recall_at_k": round(recall, 4), "recall_sweep": sweep, "settle_steps": max(args.settle, 0),
            "query_batches_rotated": N_BATCHES,
            "sharding": "list p on rank p % N, centroids replicated; every rank computes the coarse step, scans the probed "
                        "lists it owns; all-to-all of the per-rank top-k, merge on the rank that owns the query",
        },
        "roofline": my_roof,
        "phases_ms": phases_of(ev_ph),
        "build": {"sharded_kmeans_s": round(t_kmeans, 2), "niter": args.niter},
    }


# ---- one rank's step of BASELINE.json configs[3] (8 x MI355X), on ONE GPU, without the exchange ---------------------------------
def run_rank_step(ctx, dev, args, nprobe, world=8):
    """What rank 0 of an 8-GPU configs[3] run does per batch, minus the collectives: 100M x 128 / 65536 lists / 4096 queries over 8
    GPUs = 12.5M local vectors in the 8192 lists p % 8 == 0, all 65536 centroids replicated.  Step = coarse for the rank's 512-query
    slice against the 65536 centroids + scan of the WHOLE 4096-query batch over the local lists (a probe of another rank's list
    finds nothing here) with squared keys + qk_pack_topk of the [4096][k] result into the 8 blocks of the all-to-all.  The local
    centroids come from a k-means of the local vectors (8192 clusters: what the global k-means finds in this rank's region), the
    other 57344 are the true centres of the other ranks' components."""
    from quake_amd.capi import Store
    n, d, k, metric = args.nvec_sharded, 128, args.k, "l2"
    nl_local, Q = args.nlist_sharded, args.batch_sharded * world
    nl = nl_local * world
    per = Q // world
    g0 = torch.Generator(device=dev).manual_seed(1)
    cent_true = torch.randn(nl, d, generator=g0, device=dev)
    local = torch.arange(0, nl, world, device=dev)
    x, _ = gen_mixture(n, d, nl_local, seed=1000, device=dev, cent=cent_true[local].contiguous())
    t0 = time.time()
    c_local, assign, _ = ctx.kmeans(x, nl_local, metric, niter=args.niter, seed=1234)
    torch.cuda.synchronize()
    t_km = time.time() - t0
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nl_local).cpu().numpy().astype(np.int64)
    offsets = np.zeros(nl + 1, np.int64)
    sizes = np.zeros(nl, np.int64)
    sizes[::world] = counts  # local cluster i is global list i * world
    offsets[1:] = np.cumsum(sizes)
    store = Store(ctx, d)
    store.build_csr(offsets, order.contiguous(), x[order].contiguous())
    del x, order, assign
    cent_all = cent_true.clone()
    cent_all[local] = c_local
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nl], np.int64), torch.arange(nl, device=dev), cent_all.contiguous())
    batches = [gen_queries(Q, cent_true, seed=2 + b, device=dev) for b in range(N_BATCHES)]
    pids = [ctx.coarse(parent, q, nprobe, metric, values=False)[0].contiguous() for q in batches]  # what the ranks exchange
    out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
    out_d = torch.empty((Q, k), dtype=torch.float32, device=dev)
    packed = torch.empty((world, ctx.topk_block_bytes(per, k)), dtype=torch.uint8, device=dev)
    slice_p = torch.empty((per, nprobe), dtype=torch.int64, device=dev)
    import ctypes as C
    from quake_amd.capi import _ptr, check, metric_code
    from quake_amd._lib import QK_MEM_DEVICE

    def coarse_slice(b):
        check(ctx.lib.qk_coarse(ctx.h, parent.h, _ptr(batches[b]), per, nprobe, metric_code(metric), _ptr(slice_p), None, QK_MEM_DEVICE))

    def step(b):
        coarse_slice(b)
        ctx.set_squared_l2(True)
        ctx.scan_into(store, batches[b], pids[b], k, metric, (out_i, out_d))
        ctx.set_squared_l2(False)
        ctx.pack_topk(out_i, out_d, world, out=packed)

    def timeit(fn, reps):
        for i in range(10):
            fn(i % N_BATCHES)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # (the context is bound to torch's current stream)
        for i in range(reps):
            fn(i % N_BATCHES)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for i in range(32):  # form feedback of the scan shape: synchronised calls
        step(i % N_BATCHES)
        torch.cuda.synchronize()
    step_ms = timeit(step, 100)
    coarse_ms = timeit(coarse_slice, 100)
    ctx.set_timing(3)
    for i in range(50):
        step(i % N_BATCHES)
    ev = ctx.read_timing()
    ctx.set_timing(1)
    sb, pairs = 0, 0
    for b in range(N_BATCHES):
        tm = ctx.scan(store, batches[b], pids[b], k, metric, timing=True)[2]
        sb += int(tm["scan_bytes"])
        pairs += int(tm["partitions_scanned"])
    ctx.set_timing(0)
    kern = ctx.last_scan_kernel()
    # the step's answer against brute force over the local vectors, restricted to the local probes: ids of batch 0 (parity of the
    # scan itself is the headline's and the tests' business; this guards the emulation)
    cf = 2.0 * per * nl * d
    res = {
        "config": {"workload": f"one rank's step of BASELINE.json configs[3] (100M x 128 L2, nlist=65536, batch=4096, k={k}, 8 GPUs) on one "
                               f"GPU without the exchange: {n} local vectors in the {nl_local} lists p % {world} == 0, {nl} replicated "
                               f"centroids, coarse for the rank's {per}-query slice + scan of all {Q} queries over the local lists + "
                               f"qk_pack_topk; nprobe={nprobe}",
                   "nvec_local": n, "nlist": nl, "nlist_local": nl_local, "batch": Q, "slice": per, "k": k, "nprobe": nprobe,
                   "local_kmeans_s": round(t_km, 2)},
        "ms_per_step": round(step_ms, 4), "value": round(Q / (step_ms * 1e-3), 1), "unit": "queries/s",
        "note": "value = batch / step time of ONE rank: the whole-job rate of 8 such ranks if the two collectives (all-gather of the "
                "list numbers, all-to-all of 12-byte records: 491 KB per rank) were free; device events on the launch stream",
        "probed_local_pairs_per_batch": pairs // N_BATCHES,
        "roofline": roofline_of(sb // N_BATCHES, ev, kernel=kern),
        "coarse": {"kernel_ms": round(coarse_ms, 4), "flops": int(cf), "achieved": round(cf / (coarse_ms * 1e-3) / 1e12, 1),
                   "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(cf / (coarse_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 3),
                   "bound": "mfma" if nprobe == 1 else "mfma (bf16 prefilter + exact fp32 finish: algorithmic flops over the fp32 peak)",
                   "note": f"{per} queries x {nl} centroids x {d}: 2*Q*nlist*d flops / the coarse call's device time (prep + rank + select)"},
    }
    store.close()
    parent.close()
    torch.cuda.empty_cache()
    return res


# ---- N > 1 in ONE process: the device group (IndexBuildParams::num_workers = N) ---------------------------------------------------
def run_group(ctx, dev, args, world):
    """`--gpus N --single-process`: BASELINE.json configs[3] shape through the device group of the C ABI (qk_group_*: what
    QuakeIndex does with num_workers = N): one process, member j on GPU j % #GPUs, list p in member p % N, the centroids
    replicated by the library; a step = one qk_group_search of the whole batch with queries and answers in the lead's HBM."""
    from quake_amd.capi import Group, Store
    n, d, k, metric = args.nvec_sharded * world, args.dim, args.k, args.metric
    nlist, Q = args.nlist_sharded * world, args.batch_sharded * world
    unit = metric == "ip"
    ndev = torch.cuda.device_count()
    devices = [j % ndev for j in range(world)]
    x, cent_true = gen_mixture(n, d, nlist, seed=1, device=dev, unit=unit)
    torch.cuda.synchronize()
    t0 = time.time()
    centroids, assign, _ = ctx.kmeans(x, nlist, metric, niter=args.niter, seed=1234)
    torch.cuda.synchronize()
    t_kmeans = time.time() - t0
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(counts)
    ids_sorted, x_sorted = order.contiguous(), x[order].contiguous()
    del order, assign
    t0 = time.time()
    grp = Group(devices, d)
    grp.build_csr(offsets, ids_sorted, x_sorted)
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
    log(f"group of {world} members on devices {devices}: {n} vectors, nlist={nlist}, k-means {t_kmeans:.2f}s, distribution {time.time() - t0:.1f}s")
    del ids_sorted, x_sorted
    batches = [gen_queries(Q, cent_true, seed=2 + b, device=dev, unit=unit) for b in range(N_BATCHES)]
    gts = [brute_force_topk(q, x, k, metric=metric)[0] for q in batches]
    del x
    torch.cuda.empty_cache()
    grp.set_stream(torch.cuda.current_stream().cuda_stream)
    out = (torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev))

    def step(nprobe, b, slot=0):
        return grp.search(parent, batches[b], nprobe, k, metric, out=out)

    nprobe, recall, sweep = pick_nprobe(step, batches, gts, k, args.recall_target, args.nprobe, lambda ri, b: recall_at_k(ri, gts[b], k))
    log(f"[group] nprobe={nprobe} recall@{k}={recall:.4f} sweep={sweep}")
    # parity: the group's answer == the one-store search over the same lists (when the whole index also fits one store: N small)
    elapsed, _, _, gtimes = timed_region(ctx, step, nprobe, args.steps, args.warmup, args.settle, None, dev, sync=grp.synchronize)
    _, _, tm = grp.search(parent, batches[0], nprobe, k, metric, timing=True)
    sb = 0
    for b in range(N_BATCHES):
        sb += int(grp.search(parent, batches[b], nprobe, k, metric, timing=True)[2]["scan_bytes"])
    sb //= N_BATCHES
    # every member streams its share of the unique bytes concurrently: aggregate bytes / the lead's scan phase against N x peak
    scan_ms = tm["scan_ms"]
    agg = sb / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    distinct = len(set(devices))
    res = {
        "value": round(Q * args.steps / elapsed, 1), "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "timed_groups": groups_of(gtimes, args.steps, Q),
        "config": {
            "workload": f"Synthetic {n // 1_000_000}M x {d} f32 {metric.upper()} Gaussian mixture, nlist={nlist}, device group of {world} "
                        f"members in ONE process (qk_group_search = QuakeIndex with num_workers={world}), list p in member p % {world}, "
                        f"batch={Q} queries, k={k}, nprobe={nprobe} (BASELINE.json configs[3] shape: 12.5M vectors, 8192 lists and "
                        f"512 queries per member)",
            "nvec": n, "dim": d, "metric_type": metric, "nlist": nlist, "batch": Q, "k": k, "nprobe": nprobe,
            "recall_at_k": round(recall, 4), "recall_sweep": sweep, "settle_steps": max(args.settle, 0),
            "query_batches_rotated": N_BATCHES, "devices": devices,
            "sharding": "single process: coarse split by queries, list numbers and packed top-k peer-written over xGMI, merge on "
                        "the lead; " + ("members SHARE devices: a functional run, not a scaling figure" if distinct < world else
                                        "one member per GPU"),
        },
        "roofline": {"kernel": "partition scan of every member (lead's scan phase: entry of the scans to the arrival of the last block)",
                     "bound": "hbm", "achieved": round(agg, 1), "peak": HBM_PEAK_GBS * distinct, "unit": "GB/s",
                     "frac": round(agg / (HBM_PEAK_GBS * distinct), 4), "traffic": None, "algorithmic_bytes_per_launch": int(sb),
                     "kernel_ms_avg": round(scan_ms, 5),
                     "note": "unique probed bytes over ALL members / the lead's scan phase (HIP events on the lead's stream), against "
                             f"{distinct} x the HBM peak"},
        "phases_ms": {"coarse": round(tm["coarse_ms"], 4), "group": 0.0, "scan": round(tm["scan_ms"], 4), "merge": round(tm["merge_ms"], 4),
                      "note": "events on the lead's stream: coarse = batch in to list numbers everywhere; scan = to the last packed block; merge"},
        "build": {"kmeans_s": round(t_kmeans, 2), "niter": args.niter},
    }
    grp.close()
    parent.close()
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks of one node (static rendezvous on
    127.0.0.1 and a free port -- the container hostname may not resolve).  Returns the exit code of the job.  Fails before
    anything is started when the node shows fewer than N GPUs (RCCL needs one per rank; QUAKE_BENCH_BACKEND=gloo lets ranks
    share a GPU for a functional run)."""
    import socket
    backend = os.environ.get("QUAKE_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print(f"bench.py --gpus {n}: this node shows {have} GPU(s); RCCL needs one per rank "
              f"(QUAKE_BENCH_BACKEND=gloo runs the ranks on shared GPUs, functionally)", file=sys.stderr, flush=True)
        return 2
    if have < 1:
        print(f"bench.py --gpus {n}: no GPU visible", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching", n, "ranks:", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=200,
                    help="untimed steps before the warmup (the first ~100 steps after the build run at lower clocks)")
    ap.add_argument("--nvec", type=int, default=10_000_000, help="vectors (single GPU)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=4096, help="lists (single GPU)")
    ap.add_argument("--batch", type=int, default=1024, help="queries per step (single GPU)")
    ap.add_argument("--nvec-sharded", type=int, default=12_500_000, help="vectors per GPU when N > 1 (configs[3]: 100M / 8)")
    ap.add_argument("--nlist-sharded", type=int, default=8192, help="lists per GPU when N > 1 (configs[3]: 65536 / 8)")
    ap.add_argument("--batch-sharded", type=int, default=512, help="queries per GPU per step when N > 1 (configs[3]: 4096 / 8)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", choices=("l2", "ip"), default="l2",
                    help="ip: unit-norm mixture (embedding-like), BASELINE.json configs[2] with --dim 768 --k 100")
    ap.add_argument("--sigma", type=float, default=0.3, help="within-cluster sigma of the headline mixture")
    ap.add_argument("--hard-latent", type=int, default=10, help="latent dimension of the second workload's corpus (gen_manifold)")
    ap.add_argument("--manifold", type=int, default=0, help="headline corpus = gen_manifold with this latent dimension (probe)")
    ap.add_argument("--nprobe", type=int, default=0, help="0 = sweep for recall@k >= target")
    ap.add_argument("--recall-target", type=float, default=0.9)
    ap.add_argument("--niter", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs (and the oracle parity check)")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only")
    ap.add_argument("--inflight", type=int, default=2,
                    help="extra measurement beside the timed region: the same steps rotated over this many HIP streams (1 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 FETCH_SIZE pass (roofline.traffic falls back to profiles/)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-buffer (QK_MEM_HOST) rate")
    ap.add_argument("--traffic-probe", action="store_true", help="internal: the short replay measured_traffic() profiles")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in ONE process through the device group of the C ABI (qk_group_*: QuakeIndex with num_workers = N) "
                         "instead of N torch.distributed ranks")
    ap.add_argument("--sift-dir", default="", help="directory with <name>_base.fvecs / _query.fvecs [/ _groundtruth.ivecs] (e.g. TEXMEX sift/): "
                    "the headline workload runs on these vectors verbatim instead of the synthetic mixture (use --nlist 1024 for configs[0]'s index)")
    ap.add_argument("--only", default="", help="comma-separated subset of the extra workloads (hard,configs0,configs2,configs3_rank_step)")
    args = ap.parse_args()

    if args.single_process:
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (the coordinator starts its own workers,
        # query_coordinator.cpp:50-95) -- one process per GPU through torch.distributed.run, rank 0 prints the JSON line
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus}: the launcher started {world} rank(s) (WORLD_SIZE); launch one rank per GPU with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py "
                         f"--gpus {args.gpus} ...`")
    # QUAKE_BENCH_BACKEND=gloo lets two ranks share one GPU (functional check of the N>1 path on a 1-GPU box)
    backend = os.environ.get("QUAKE_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # QUAKE_BENCH_FORCE_SHARDED=1 (with QUAKE_FORCE_COLLECTIVES=1): the N > 1 path with ONE rank -- every collective of it runs
    # on the real backend (tests/test_rccl_world1_gpu.py)
    force_sharded = world == 1 and os.environ.get("QUAKE_BENCH_FORCE_SHARDED", "0") not in ("", "0")
    if world > 1 or force_sharded:
        import torch.distributed as dist_
        dist = dist_
        if force_sharded:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: the {backend} process group has {dist.get_world_size()} rank(s)")
        if backend == "nccl" and world > 1 and torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py --gpus {args.gpus}: RCCL needs one GPU per rank, this node shows {torch.cuda.device_count()}")

    from quake_amd.capi import Context
    ctx = Context(dev_index)
    # the library runs on torch's current stream, so that it is ordered with torch / torch.distributed work
    # (QUAKE_BENCH_STREAM=private: A/B against the context's own non-blocking stream, single GPU only)
    if not (os.environ.get("QUAKE_BENCH_STREAM") == "private" and world == 1):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    log("device", ctx.device_info())
    t_all = time.time()
    if args.traffic_probe:
        run_traffic_probe(ctx, dev, args)
        return

    if world == 1 and not force_sharded:
        main_res = run_single_workload(ctx, dev, args, "headline", args.sigma, args.nprobe, args.steps, args.warmup, args.settle,
                                       args.cpu_seconds, traffic_file="r06_pmc_k_scan.json", manifold=args.manifold,
                                       sweep_nprobes=() if (args.no_extra or args.manifold) else (8, 16, 32))
        cfg_no = 2 if (args.metric == "ip" and args.dim == 768) else 1
        main_res["config"]["workload"] += f" (BASELINE.json configs[{cfg_no}])"
        main_res["config"]["sharding"] = "single GPU"
    else:
        main_res = run_sharded(ctx, dev, args, dist, rank, world)

    result = {
        "metric": METRIC_NAME, "value": main_res["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": main_res["config"].get("data", "synthetic"), "config": main_res["config"],
        # (`steps` is the contract's K: `value` is the median over `timed_steps_total` / K groups of exactly K steps, each between
        #  barrier + synchronise -- timed_groups lists every group)
        "timed_steps_total": (main_res.get("timed_groups") or {}).get("groups", 1) * args.steps if isinstance(main_res.get("timed_groups"), dict) else args.steps,
        "roofline": main_res["roofline"], "phases_ms": main_res["phases_ms"], "build": main_res["build"],
        "timed_groups": main_res.get("timed_groups"),
    }
    if main_res.get("batches_in_flight"):
        result["batches_in_flight"] = main_res["batches_in_flight"]
    if main_res.get("host_api"):
        result["host_api"] = main_res["host_api"]
    for key in ("per_rank", "exchange", "rccl_ranks_seen"):  # the N > 1 path: every rank's own roofline, the devices, the exchange
        if main_res.get(key):
            result[key] = main_res[key]
    if world == 1 and not force_sharded:
        result["cpu_baseline"] = main_res.get("cpu_baseline")
        if "speedup_vs_cpu" in main_res:
            result["speedup_vs_cpu"] = main_res["speedup_vs_cpu"]
        if not args.no_extra:
            extra = {}
            only = set(v for v in args.only.split(",") if v)
            hard_steps = max(20, min(args.steps, 100))
            if not only or "hard" in only:
                extra["hard"] = run_single_workload(ctx, dev, args, "hard", 0.0, 0, hard_steps, min(args.warmup, 10),
                                                    min(args.settle, 50), args.cpu_seconds * 0.5, traffic_file="r05_pmc_k_scan_hard.json",
                                                    manifold=args.hard_latent, with_aps=True)
            if not only or "configs0" in only:
                extra["configs0"] = run_configs0(ctx, dev, args)
            if (not only or "configs2" in only) and cfg_no == 1:
                # BASELINE.json configs[2]: 10M x 768 unit-norm IP, nlist 4096, batch 1024, k = 100 -- its own value, roofline (with
                # the in-run FETCH_SIZE pass), cpu_baseline and in-run parity
                a2 = argparse.Namespace(**vars(args))
                a2.dim, a2.metric, a2.k, a2.inflight = 768, "ip", 100, 1
                extra["configs2"] = run_single_workload(ctx, dev, a2, "configs2", args.sigma, 0, hard_steps, min(args.warmup, 10),
                                                        min(args.settle, 50), args.cpu_seconds * 0.5)
                extra["configs2"]["config"]["workload"] += " (BASELINE.json configs[2])"
            if (not only or "configs3_rank_step" in only) and cfg_no == 1:
                extra["configs3_rank_step"] = run_rank_step(ctx, dev, args, int(main_res["config"]["nprobe"]))
            if main_res.get("nprobe_sweep"):
                # the headline index at fixed nprobe 8 / 16 / 32: the regime where a probed list is shared by many queries of the
                # batch (the mixed work sequence of the row-per-lane scan), each line with its own roofline
                extra["nprobe_sweep"] = main_res["nprobe_sweep"]
            if main_res.get("recall_target_search"):
                extra["recall_target_search"] = main_res["recall_target_search"]
            if extra.get("hard", {}).get("recall_target_search"):
                # the fair APS benchmark: on the mixture every boundary lies beyond the query radius and the reference's estimate falls
                # back to uniform probabilities (geometry.h:389-393) -- 48 lists scanned where nprobe 1 already has recall 0.95; on the
                # low-intrinsic-dimension corpus the probabilities discriminate
                extra["recall_target_search_hard"] = extra["hard"].pop("recall_target_search")
            result["workloads"] = extra

    log(f"total bench wall {time.time() - t_all:.1f}s")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_single_process(args):
    world = max(1, args.gpus)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from quake_amd.capi import Context
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    log("device", ctx.device_info(), "| GPUs visible:", torch.cuda.device_count())
    t_all = time.time()
    r = run_group(ctx, dev, args, world)
    result = {
        "metric": METRIC_NAME, "value": r["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "config": r["config"], "roofline": r["roofline"], "phases_ms": r["phases_ms"], "build": r["build"],
        "timed_groups": r["timed_groups"], "launch": "single process, device group (qk_group_*)",
    }
    log(f"total bench wall {time.time() - t_all:.1f}s")
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
