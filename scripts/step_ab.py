"""ms per step of the headline search (10M x 128, 4096 lists, 1024 queries, nprobe 1, k = 10) with whatever library QUAKE_HIP_LIB
names: for A/B runs of a side library (scripts/build_variant.sh) against the product on one box, processes alternating:
    for i in 1 2 3; do python scripts/step_ab.py; QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_x.so python scripts/step_ab.py; done"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context

nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 1
k_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
if os.environ.get("STEP_FEEDBACK") == "0":  # the static form rule alone
    ctx.set_form_feedback(False)
n, d, nlist, Q, k = int(os.environ.get("STEP_N", "10000000")), int(os.environ.get("STEP_DIM", "128")), int(os.environ.get("STEP_NLIST", "4096")), int(os.environ.get("STEP_Q", "1024")), k_arg
metric = os.environ.get("STEP_METRIC", "l2")
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev)
idx = B.build_single(ctx, dev, x, nlist, metric, 5, keep_host=False)
del x
qs = [B.gen_queries(Q, cent, seed=2 + b, device=dev) for b in range(4)]
out = (torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev))


def block(steps=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ctx.search(idx["parent"], idx["store"], qs[i % 4], nprobe, k, metric, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


block(300)
ms = sorted(block(200) for _ in range(9))
chk = int(out[0].sum().item())
print(json.dumps({"lib": os.environ.get("QUAKE_HIP_LIB", "product"), "n": n, "d": d, "nlist": nlist, "Q": Q, "metric": metric, "nprobe": nprobe, "k": k, "kernel": ctx.last_scan_kernel(), "feedback": os.environ.get("STEP_FEEDBACK", "1"), "ms_per_step_median": round(ms[4], 5), "min": round(ms[0], 5),
                  "max": round(ms[-1], 5), "ids_checksum": chk}))
