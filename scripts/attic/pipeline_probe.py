"""Throughput of back-to-back search batches with 1, 2 or 3 contexts (each on its own non-blocking stream, its own
workspace) taking the batches in turn -- the small grouping / coarse kernels of batch i+1 run beside the scan of batch i.
Run on the GPU box:  python scripts/pipeline_probe.py [nvec] [nlist] [nprobe]"""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B


def main():
    nvec = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    d, k, Q = 128, 10, 1024
    from quake_amd.capi import Context, Store
    dev = torch.device("cuda", 0)
    ctxs = [Context(0) for _ in range(3)]
    c0 = ctxs[0]
    c0.set_stream(torch.cuda.current_stream().cuda_stream)
    x, cent_true = B.gen_mixture(nvec, d, nlist, seed=1, device=dev, unit=False)
    centroids, assign, _ = c0.kmeans(x, nlist, "l2", niter=5, seed=1234)
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
    offsets = np.zeros(nlist + 1, np.int64); offsets[1:] = np.cumsum(counts)
    store = Store(c0, d); store.build_csr(offsets, order.contiguous(), x[order].contiguous())
    parent = Store(c0, d); parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), centroids.contiguous())
    torch.cuda.synchronize()
    c0.set_stream(None)
    qs = [B.gen_queries(Q, cent_true, seed=2 + i, device=dev, unit=False) for i in range(3)]
    outs = [(torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev)) for _ in range(3)]
    ref = c0.search(parent, store, qs[0], nprobe, k, "l2")
    torch.cuda.synchronize()
    ref_i = ref[0].clone()
    for nctx in (1, 2, 3, 1, 2, 3):
        for w in range(30):
            ctxs[w % nctx].search(parent, store, qs[w % nctx], nprobe, k, "l2", out=outs[w % nctx])
        torch.cuda.synchronize()
        steps = 300
        t0 = time.perf_counter()
        for i in range(steps):
            ctxs[i % nctx].search(parent, store, qs[i % nctx], nprobe, k, "l2", out=outs[i % nctx])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        same = bool((outs[0][0] == ref_i).all().item())
        print(json.dumps({"contexts": nctx, "ms_per_step": round(1e3 * el / steps, 4), "qps": round(Q * steps / el, 1), "ids_equal": same}), flush=True)


if __name__ == "__main__":
    main()
