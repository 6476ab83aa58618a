"""The host-buffer entry point (what a caller of the reference's CPU-tensor API sees): numpy queries in, numpy ids / distances out,
one synchronisation per call.  ms per call on the 10M bench index for 1024 / 64 / 1 queries at nprobe 1 and 8, with whatever library
QUAKE_HIP_LIB names (A/B: scripts/build_variant.sh nopin qk_api.hip -DQK_HOST_PINNED_IO=0)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context

dev = torch.device("cuda", 0)
ctx = Context(0)
n, d, nlist = 10_000_000, 128, 4096
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev)
idx = B.build_single(ctx, dev, x, nlist, "l2", 5, keep_host=False)
del x
out = {"lib": os.environ.get("QUAKE_HIP_LIB", "product")}
for Q in (1024, 64, 1):
    qs = [B.gen_queries(Q, cent, seed=2 + b, device=dev).cpu().numpy() for b in range(4)]
    for nprobe in (1, 8):
        for i in range(50):
            ctx.search(idx["parent"], idx["store"], qs[i % 4], nprobe, 10, "l2")
        ts = []
        for rep in range(7):
            t0 = time.perf_counter()
            for i in range(100):
                ri, rd = ctx.search(idx["parent"], idx["store"], qs[i % 4], nprobe, 10, "l2")
            ts.append((time.perf_counter() - t0) / 100 * 1e3)
        out[f"Q{Q}_nprobe{nprobe}_ms"] = round(sorted(ts)[3], 4)
        out[f"Q{Q}_nprobe{nprobe}_checksum"] = int(ri.sum())
print(json.dumps(out))
