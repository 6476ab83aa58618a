"""Partition-scan kernel time against both roofs over nprobe, one index build per process.

    python scripts/nprobe_sweep.py --nprobes 8,16,32 [--corpus mixture|hard] [--steps 50] [--tag NAME] [--parity]

One JSON line per nprobe: the form that ran (qk_ctx_last_scan_kernel), the scan kernel's mean duration from the deferred HIP
events of the timed steps (timing mode 3), unique bytes / time against the HBM peak, 2 d flops per (row, probing query)
against the fp32 MFMA peak, fraction of the roof that gives the longer minimum time, and the whole-step time.  --parity also
checks batch 0 against the oracle's batched path (ids and distance bits).  QK_* switches need the probe build."""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context

ap = argparse.ArgumentParser()
ap.add_argument("--nprobes", default="8,16,32")
ap.add_argument("--corpus", default="mixture")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--tag", default="")
ap.add_argument("--parity", action="store_true")
ap.add_argument("--nvec", type=int, default=10_000_000)
ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--k", type=int, default=10)
a = ap.parse_args()
n, d, nlist, Q, k = a.nvec, 128, a.nlist, a.batch, a.k
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
if a.corpus == "mixture":
    x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev)
    batches = [B.gen_queries(Q, cent_true, seed=2 + b, device=dev) for b in range(B.N_BATCHES)]
else:
    x, basis = B.gen_manifold(n, d, seed=1, device=dev)
    batches = [B.gen_manifold(Q, d, seed=2 + b, device=dev, basis=basis)[0] for b in range(B.N_BATCHES)]
idx = B.build_single(ctx, dev, x, nlist, "l2", 5, keep_host=a.parity)
del x
parent, store = idx["parent"], idx["store"]
out = (torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev))
cnt_t = torch.as_tensor(idx["counts"], device=dev)


def step(nprobe, b, slot=0):
    return ctx.search(parent, store, batches[b], nprobe, k, "l2", out=out)


results = []
for nprobe in [int(v) for v in a.nprobes.split(",")]:
    elapsed, ev, ev_ph, _ = B.timed_region(ctx, step, nprobe, a.steps, 10, 30, None, dev, groups=1)
    ctx.set_timing(1)
    sb = 0
    for b in range(B.N_BATCHES):
        sb += int(ctx.search(parent, store, batches[b], nprobe, k, "l2", timing=True)[2]["scan_bytes"])
    ctx.set_timing(0)
    sb //= B.N_BATCHES
    kern = ctx.last_scan_kernel()
    pair_rows = int(sum(int(cnt_t[ctx.coarse(parent, batches[b], nprobe, "l2")[0]].sum().item()) for b in range(B.N_BATCHES)) // B.N_BATCHES)
    r = B.roofline_of(sb, ev, kernel=kern, pair_rows=pair_rows, d=d)
    m = r["mfma"]
    bind_ms = max(m["min_ms_hbm"], m["min_ms_mfma"])
    res = {"tag": a.tag, "corpus": a.corpus, "nprobe": nprobe, "kernel": kern, "scan_ms": r["kernel_ms_avg"],
           "hbm_frac_unique": round(sb / (r["kernel_ms_avg"] * 1e-3) / 1e9 / B.HBM_PEAK_GBS, 4), "mfma_frac": m["frac"],
           "bound": r["bound"], "frac_of_binding_roof": round(bind_ms / r["kernel_ms_avg"], 4),
           "min_ms_hbm": m["min_ms_hbm"], "min_ms_mfma": m["min_ms_mfma"], "unique_GB": round(sb / 1e9, 3),
           "queries_per_row": m["queries_per_scanned_row"], "step_ms": round(1e3 * elapsed / a.steps, 4),
           "phases_ms": B.phases_of(ev_ph)}
    results.append(res)
# every timing first, the oracle afterwards: its 128 threads keep spinning for a while after a call and starve the launching thread
# of the next timed region (whole-step times 0.25 ms too long on the second corpus when the two alternated)
for res in results:
    if a.parity:
        import oracle as O
        hv, hi, ho, hc = idx["host"]
        gi, gd = step(res["nprobe"], 0)
        torch.cuda.synchronize()
        oi, od = O.search(batches[0].cpu().numpy(), hc, hv, hi, ho, res["nprobe"], k, "l2", batched_scan=True, num_threads=O.max_threads())
        res["ids_equal"] = bool((oi == gi.cpu().numpy()).all())
        res["dist_bits_equal"] = bool((od.view(np.uint32) == gd.cpu().numpy().view(np.uint32)).all())
    print(json.dumps(res), flush=True)
