"""Where the prefiltered assign stops paying: 262144 rows through qk_assign_pf.hip against the fp32 kernel (30000 rows, scaled), over the
number of centroids (64 ... 1024) and d = 128 / 64 / 32.  Round 4: the prefiltered form wins from 64 centroids on (0.15 against 0.42 ms).
python scripts/assign_crossover.py"""
import json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context
dev = torch.device("cuda", 0)
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
def t(x, c, reps=10):
    ctx.kmeans_assign(x, c, "l2", values=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ctx.kmeans_assign(x, c, "l2", values=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for d in (128, 64, 32):
    xb, _ = B.gen_mixture(1 << 20, d, 1024, seed=1, device=dev)
    for m in (64, 128, 192, 256, 384, 512, 1024):
        c = xb[torch.randperm(1 << 20, device=dev)[:m]].contiguous()
        big = t(xb[:1 << 18].contiguous(), c)             # 262144 rows: prefiltered
        small = t(xb[:30000].contiguous(), c) * (262144 / 30000.0)   # fp32 kernel, scaled to the same rows
        print(json.dumps({"d": d, "m": m, "pf_ms_262144": round(big, 4), "fp32_ms_scaled": round(small, 4)}), flush=True)
