"""Phases of one search when a whole batch concentrates on a few neighbouring lists (the "skewed" sampling of the dynamic
workload, BASELINE.json configs[4]: workload.ClusterWalk): n x 128 in nlist lists, 1024 queries drawn around `nc` neighbouring
cluster centres.  python scripts/skew_probe.py [n] [nlist] [nprobe]   (QK_* switches need the probe build)"""
import json, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 400
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
x, cent = B.gen_mixture(n, 128, max(n // 2500, 16), seed=1, device=dev)
idx = B.build_single(ctx, dev, x, nlist, "l2", 5, keep_host=False)
g = torch.Generator(device=dev).manual_seed(5)
for nc in (2, 8, 64, cent.shape[0]):
    root = cent[7:8]
    near = torch.cdist(root, cent)[0].argsort()[:nc]            # nc neighbouring clusters
    q = B.gen_queries(1024, cent[near], seed=9, device=dev)
    ctx.set_timing(1)
    rows = [ctx.search(idx["parent"], idx["store"], q, nprobe, 10, "l2", timing=True)[2] for _ in range(12)]
    ctx.set_timing(0)
    med = {k: round(float(np.median([r[k] for r in rows[4:]])) * 1e3, 1) for k in ("coarse_ms", "group_ms", "scan_ms", "merge_ms")}
    print(json.dumps({"n": n, "nlist": nlist, "nprobe": nprobe, "query_clusters": nc, "kernel": ctx.last_scan_kernel(), **med,
                      "scan_MB": round(rows[-1]["scan_bytes"] / 1e6, 1)}), flush=True)
