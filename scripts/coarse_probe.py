"""Coarse step alone (qk_coarse: k_prep_queries + k_dense_ord + k_select_rows) vs number of centroids -- what a rank pays per
step when the centroids of all ranks are replicated (bench.py --gpus N: nlist = 4096 N).  python scripts/coarse_probe.py"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd.capi import Context, Store

ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
d, Q = 128, int(sys.argv[3]) if len(sys.argv) > 3 else 1024
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(Q, d, generator=g, device=dev)
NLS = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (4096, 8192, 16384, 32768, 65536)
NPS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 32, 100, 400)
for nl in NLS:
    c = torch.randn(nl, d, generator=g, device=dev)
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nl], np.int64), torch.arange(nl, device=dev), c)
    for nprobe in NPS:
        for _ in range(3):
            ctx.coarse(parent, q, nprobe, "l2")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.coarse(parent, q, nprobe, "l2")
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 50 * 1e6
        print(json.dumps({"nlist": nl, "Q": Q, "nprobe": nprobe, "kernel": ctx.last_scan_kernel(), "coarse_us": round(us, 1), "TFLOPs": round(2.0 * Q * nl * d / us / 1e6, 1)}), flush=True)
