"""Two batches in flight: alternate the steps of the bench loop between two contexts (two HIP streams) over the SAME stores.
The small kernels of one batch (prep, nearest centroid, group + seed, merge) then run under the partition scan of the other.
python scripts/pipeline2_probe.py [nprobe]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from quake_amd.capi import Context, Store

nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, d, nlist, k, Q = 10_000_000, 128, 4096, 10, 1024
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
del x
batches = [B.gen_queries(Q, cent_true, seed=2 + b, device=dev) for b in range(4)]
torch.cuda.synchronize()
ctx2 = Context(0)  # private non-blocking stream
outs = [(torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev)) for _ in range(2)]
ref = [ctx.search(parent, store, batches[b], nprobe, k, "l2") for b in range(4)]
torch.cuda.synchronize()


def run(ctxs, steps):
    for i in range(100):
        c = ctxs[i % len(ctxs)]
        c.search(parent, store, batches[i % 4], nprobe, k, "l2", out=outs[i % len(ctxs)])
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        c = ctxs[i % len(ctxs)]
        c.search(parent, store, batches[i % 4], nprobe, k, "l2", out=outs[i % len(ctxs)])
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for name, cs in (("one stream", [ctx]), ("two streams", [ctx, ctx2]), ("one stream", [ctx]), ("two streams", [ctx, ctx2])):
    t = run(cs, 400)
    print(json.dumps({"mode": name, "nprobe": nprobe, "ms_per_step": round(t * 1e3, 4), "qps": round(Q / t, 1)}), flush=True)
# results of the pipelined form are the same answers
for i in range(8):
    c = (ctx, ctx2)[i % 2]
    gi, gd = c.search(parent, store, batches[i % 4], nprobe, k, "l2", out=outs[i % 2])
    c.synchronize()
    assert torch.equal(gi, ref[i % 4][0]) and torch.equal(gd, ref[i % 4][1])
print("pipelined answers identical")
