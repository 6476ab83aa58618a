"""PCIe-inclusive rate of the headline workload: the same 10M x 128 index and 1024-query batches as bench.py, but the caller hands
over HOST buffers (numpy queries in, numpy ids / distances out: QK_MEM_HOST, what the reference's CPU tensors are) -- every call
stages the queries to the device, runs the search, copies the answers back and synchronises.  Never bench.py's `value`.
    python scripts/host_buffers_probe.py"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from quake_amd.capi import Context

dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n, d, nlist, k = 10_000_000, 128, 4096, 10
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev, sigma=0.3)
idx = B.build_single(ctx, dev, x, nlist, "l2", niter=5, keep_host=False)
del x
qs_dev = [B.gen_queries(1024, cent, seed=2 + b, device=dev, sigma=0.3) for b in range(4)]
qs_host = [q.cpu().numpy() for q in qs_dev]
out = {}
for name, qs in (("device_buffers", qs_dev), ("host_buffers", qs_host)):
    for nprobe in (1, 8):
        for i in range(100):
            ctx.search(idx["parent"], idx["store"], qs[i % 4], nprobe, k, "l2")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 300
        for i in range(steps):
            ctx.search(idx["parent"], idx["store"], qs[i % 4], nprobe, k, "l2")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"{name}_nprobe{nprobe}"] = {"queries_per_s": round(1024 * steps / dt, 1), "ms_per_batch": round(1e3 * dt / steps, 4)}
print(json.dumps(out))
