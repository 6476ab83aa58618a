"""k-means update step alone (bucketing + accumulate) on 2^20 x 128 rows: device ms per call for both summation orders and three
assignment patterns -- uniform random (every bucket's rows scattered over the array: the Lloyd iteration's case), contiguous
(rows of a centroid adjacent: what the gather would cost without the scatter) and the skewed mixture of the bench build.
python scripts/accum_probe.py [n] [m] [d]"""
import json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quake_amd.capi import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda", 0)
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, d, device=dev, generator=g)
pats = {"uniform": torch.randint(0, m, (n,), device=dev, generator=g),
        "contiguous": (torch.arange(n, device=dev) * m // n),
        "skewed": (torch.randn(n, device=dev, generator=g).abs() * m / 3).long().clamp(max=m - 1)}
for name, a in pats.items():
    for blocked in (False, True):
        for _ in range(3):
            ctx.kmeans_accumulate(x, a, m, blocked=blocked)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            ctx.kmeans_accumulate(x, a, m, blocked=blocked)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"n": n, "m": m, "d": d, "pattern": name, "blocked": blocked, "ms_per_call_incl_alloc": round(ms, 4),
                          "max_bucket": int(torch.bincount(a, minlength=m).max())}), flush=True)
