"""Single-query latency (BASELINE.json configs[0] shape: 1M x 128, nlist 1024, nprobe 10, k 10, batch 1) through the C ABI with
host buffers (what the reference's CPU tensors are) and with device buffers; the oracle's serial scan beside it."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from quake_amd.capi import Context, Store


def main():
    n, d, nlist, nprobe, k = 1_000_000, 128, 1024, 10, 10
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev)
    centroids, assign, _ = ctx.kmeans(x, nlist, "l2", niter=5, seed=1234)
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    offsets = np.zeros(nlist + 1, np.int64); offsets[1:] = np.cumsum(counts)
    xs, ids = x[order].contiguous(), order.contiguous()
    store = Store(ctx, d); store.build_csr(offsets, ids, xs)
    parent = Store(ctx, d); parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
    q = B.gen_queries(2000, cent_true, seed=2, device=dev)
    gi, _ = B.brute_force_topk(q, x, k)
    qh = q.cpu().numpy()
    out = {}
    for Q in (1, 4, 16, 64):
        for mode in ("host", "device"):
            lat, hits = [], 0
            nrep = 400 if Q == 1 else 100
            for i in range(nrep):
                sl = slice((i * Q) % 1900, (i * Q) % 1900 + Q)
                qq = qh[sl] if mode == "host" else q[sl]
                if mode == "device":
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                ri, rd = ctx.search(parent, store, qq, nprobe, k, "l2")
                if mode == "device":
                    torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
                ri = ri if mode == "host" else ri.cpu().numpy()
                hits += sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ri, gi[sl].cpu().numpy()))
            lat = np.array(lat[nrep // 10:]) * 1e6
            out[f"Q{Q}_{mode}"] = {"mean_us": round(float(lat.mean()), 1), "p50_us": round(float(np.median(lat)), 1),
                                   "p99_us": round(float(np.percentile(lat, 99)), 1), "qps": round(Q / lat.mean() * 1e6, 1),
                                   "recall": round(hits / (nrep * Q * k), 4)}
    # device time of one search from the library's own HIP events (no Python / launch-queue time)
    ctx.set_timing(1)
    for Q in (1, 4, 16, 64):
        ts = [ctx.search(parent, store, q[i * Q:(i + 1) * Q], nprobe, k, "l2", timing=True)[2]["total_ms"] for i in range(20)]
        out[f"Q{Q}_device_event_us"] = round(float(np.median(ts[5:])) * 1e3, 1)
    ctx.set_timing(0)
    if not os.environ.get("LAT_NO_CPU"):
        import oracle as O
        hv, hi, hc = xs.cpu().numpy(), ids.cpu().numpy(), centroids.cpu().numpy()
        t0 = time.perf_counter()
        for i in range(200):
            O.search(qh[i:i + 1], hc, hv, hi, offsets, nprobe, k, "l2", batched_scan=False, num_threads=1)
        out["cpu_oracle_Q1_1thread_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
