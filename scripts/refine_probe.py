"""qk_store_refine_lists on the bench index: wall time of a call over 100 / 400 / 1600 neighbouring partitions (2 iterations, what a
maintenance call does around its splits) -- wrap in `rocprofv3 --kernel-trace --stats` to set the kernels' sum beside it."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context

dev = torch.device("cuda", 0)
ctx = Context(0)
n, d, nlist = int(os.environ.get("PROBE_N", "10000000")), 128, int(os.environ.get("PROBE_NLIST", "4096"))
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev)
idx = B.build_single(ctx, dev, x, nlist, "l2", 5, keep_host=False)
del x
parent, store = idx["parent"], idx["store"]
c_all = parent.get_list(0)[0]
for m in (100, 400, 1600):
    near, _ = ctx.coarse(parent, c_all[:1], m, "l2")
    pids = np.sort(near.reshape(-1))
    cents = c_all[pids]
    rows = int(store.list_sizes(pids).sum())
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        newc = store.refine_lists(pids, cents, "l2", 2)
        ts.append((time.perf_counter() - t0) * 1e3)
        cents = newc
    print(json.dumps({"partitions": m, "rows": rows, "ms": [round(t, 2) for t in ts], "counters": store.counters()}), flush=True)
