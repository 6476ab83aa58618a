"""Recall-target search (adaptive partition scanning) at the bench scale: QPS, recall reached, partitions visited, rounds,
with the oracle's APS walk timed on the host cores beside it.  Run on the GPU box:
    python scripts/aps_probe.py [nvec] [nlist] [recall_target ...]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from quake_amd.capi import Context, Store


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    targets = [float(a) for a in sys.argv[3:]] or [0.8, 0.9, 0.99]
    d, k, Q = 128, 10, int(os.environ.get("APS_Q", "1024"))
    frac = float(os.environ.get("APS_FRACTION", "0.02"))
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if os.environ.get("APS_FEEDBACK") == "0":
        ctx.set_form_feedback(False)
    x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev)
    centroids, assign, _ = ctx.kmeans(x, nlist, "l2", niter=5, seed=1234)
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    offsets = np.zeros(nlist + 1, np.int64); offsets[1:] = np.cumsum(counts)
    xs = x[order].contiguous(); ids = order.contiguous()
    store = Store(ctx, d); store.build_csr(offsets, ids, xs)
    parent = Store(ctx, d); parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
    q = B.gen_queries(Q, cent_true, seed=2, device=dev)
    gi, _ = B.brute_force_topk(q, x, k)
    host = None
    if not os.environ.get("APS_NO_CPU"):
        host = (xs.cpu().numpy(), ids.cpu().numpy(), offsets, centroids.cpu().numpy(), q.cpu().numpy())
    del x, xs
    for rt in targets:
        for _ in range(int(os.environ.get("APS_WARMUP", "8"))):  # (the form feedback compares three forms twice per round shape)
            ri, rd, rn, tm = ctx.search_aps(parent, store, q, k, "l2", rt, initial_search_fraction=frac, timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 10
        for _ in range(reps):
            ri, rd, rn = ctx.search_aps(parent, store, q, k, "l2", rt, initial_search_fraction=frac)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        out = {"recall_target": rt, "qps": round(Q / el, 1), "ms_per_batch": round(el * 1e3, 3), "recall": round(B.recall_at_k(ri, gi, k), 4),
               "nscan_mean": round(rn.float().mean().item(), 2), "nscan_max": int(rn.max().item()), "rounds": int(tm["n_items"]),
               "M": max(int(np.float32(nlist) * np.float32(frac)), 1)}
        if os.environ.get("APS_ONLY"):  # (for a kernel trace: the last call's launches are the trace's tail)
            print(json.dumps(out), flush=True)
            continue
        # fixed nprobe with the same mean work, for reference
        npb = max(1, int(round(out["nscan_mean"])))
        fi, _ = ctx.search(parent, store, q, npb, k, "l2")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.search(parent, store, q, npb, k, "l2")
        torch.cuda.synchronize()
        out["fixed_nprobe"] = npb
        out["fixed_qps"] = round(Q * reps / (time.perf_counter() - t0), 1)
        out["fixed_recall"] = round(B.recall_at_k(fi, gi, k), 4)
        if host is not None:
            import oracle as O
            hv, hi, ho, hc, hq = host
            nq = 256
            t0 = time.perf_counter()
            oi, od, on = O.search_aps(hq[:nq], hc, hv, hi, ho, k, "l2", rt, initial_search_fraction=frac, expanded=False,
                                      num_threads=O.max_threads())
            tc = time.perf_counter() - t0
            out["cpu_qps"] = round(nq / tc, 1)
            out["cpu_threads"] = O.max_threads()
            out["nscan_equal_frac"] = round(float((on == rn[:nq].cpu().numpy()).mean()), 4)
            out["ids_equal_frac"] = round(float((oi == ri[:nq].cpu().numpy()).mean()), 4)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
