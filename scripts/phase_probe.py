"""Phase breakdown (library HIP events) of one search at small and medium batch sizes on the configs[0] shape
(1M x 128, nlist 1024, nprobe 10, k 10): where a batch of 8..256 queries spends its time."""
import json, os, sys
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
    store = Store(ctx, d); store.build_csr(offsets, order.contiguous(), x[order].contiguous())
    parent = Store(ctx, d); parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
    q = B.gen_queries(8192, cent_true, seed=2, device=dev)
    if os.environ.get("PHASE_PROBE_TIMING", "1") == "0":   # no library events: what the kernel trace of a plain call looks like
        for Q in [int(v) for v in sys.argv[1:]]:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for i in range(30):
                if i == 10:
                    ev[0].record()
                ctx.search(parent, store, q[(i * Q) % 4096:(i * Q) % 4096 + Q], nprobe, k, "l2")
            ev[1].record(); torch.cuda.synchronize()
            print(json.dumps({"Q": Q, "kernel": ctx.last_scan_kernel(), "back_to_back_us_per_call": round(ev[0].elapsed_time(ev[1]) * 1e3 / 20, 1)}), flush=True)
        return
    ctx.set_timing(1)
    for Q in ([int(v) for v in sys.argv[1:]] or (1, 4, 8, 16, 32, 64, 128, 256, 1024)):
        rows = []
        for i in range(24):
            tm = ctx.search(parent, store, q[(i * Q) % 4096:(i * Q) % 4096 + Q], nprobe, k, "l2", timing=True)[2]
            rows.append(tm)
        med = {kk: round(float(np.median([r[kk] for r in rows[6:]])) * 1e3, 1) for kk in ("coarse_ms", "group_ms", "scan_ms", "merge_ms", "total_ms")}
        print(json.dumps({"Q": Q, "kernel": ctx.last_scan_kernel(), **{kk.replace("_ms", "_us"): v for kk, v in med.items()},
                          "scan_MB": round(rows[-1]["scan_bytes"] / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
