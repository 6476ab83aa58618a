"""k-means assign alone on the bench shape (2^20 rows x 4096 centroids x 128) and two neighbours: ms per call (HIP events around
qk_kmeans_assign on device buffers); wrap in `rocprofv3 --kernel-trace --stats` for the kernel's own time."""
import json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context
dev = torch.device("cuda", 0)
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
shapes = [(1 << 20, 4096, 128), (1 << 20, 1024, 128), (1 << 20, 4096, 64), (1 << 18, 4096, 128), (1 << 17, 4096, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for n, m, d in shapes:
    x, _ = B.gen_mixture(n, d, m, seed=1, device=dev)
    c = x[torch.randperm(n, device=dev)[:m]].contiguous()
    for metric, values in (("l2", True), ("l2", False)):
        ctx.kmeans_assign(x, c, metric, values=values)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            a, v = ctx.kmeans_assign(x, c, metric, values=values)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"n": n, "m": m, "d": d, "metric": metric, "values": values, "ms": round(ms, 4),
                          "tflops": round(2.0 * n * m * d / ms / 1e9, 1)}), flush=True)
    del x, c
