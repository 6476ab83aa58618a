"""Device group on ONE GPU: what the single-process orchestration costs.  For G = 1, 2, 4, 8 members sharing device 0: host time
to enqueue a qk_group_search (no synchronisation) and the device time per search, beside the one-store qk_search of the same index
(the members' scans then run concurrently on one device: their sum is the one-store scan).  python scripts/group_probe.py [n] [nlist] [Q] [nprobe]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context, Group, Store

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nprobe = int(sys.argv[4]) if len(sys.argv) > 4 else 4
d, k = 128, 10
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev)
idx = B.build_single(ctx, dev, x, nlist, "l2", 5, keep_host=False)
order_ids = None
qs = [B.gen_queries(Q, cent, seed=2 + b, device=dev) for b in range(4)]
out = (torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev))


def measure(fn, sync, reps=100):
    for i in range(40):
        fn(i % 4)
        sync()
    for i in range(20):
        fn(i % 4)
    sync()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i % 4)
    t_host = time.perf_counter() - t0
    sync()
    t_all = time.perf_counter() - t0
    return round(1e3 * t_host / reps, 4), round(1e3 * t_all / reps, 4)


ref = ctx.search(idx["parent"], idx["store"], qs[0], nprobe, k, "l2")
torch.cuda.synchronize()
ref = (ref[0].clone(), ref[1].clone())
h, a = measure(lambda b: ctx.search(idx["parent"], idx["store"], qs[b], nprobe, k, "l2", out=out), torch.cuda.synchronize)
print(json.dumps({"path": "one store (qk_search)", "host_enqueue_ms": h, "ms_per_search": a, "n": n, "nlist": nlist, "Q": Q, "nprobe": nprobe}), flush=True)
# the lists once more, as host arrays for the groups
counts = idx["counts"]
offsets = np.zeros(nlist + 1, np.int64)
offsets[1:] = np.cumsum(counts)
vec = torch.empty((n, d), device=dev)
ids = torch.empty((n,), dtype=torch.int64, device=dev)
import ctypes as C
from quake_amd.capi import _ptr, check
from quake_amd._lib import QK_MEM_DEVICE
for p in range(nlist):
    a0, a1 = int(offsets[p]), int(offsets[p + 1])
    if a1 > a0:
        check(ctx.lib.qk_store_get_list(idx["store"].h, p, _ptr(vec[a0:a1]), _ptr(ids[a0:a1]), QK_MEM_DEVICE))
torch.cuda.synchronize()
for G in (1, 2, 4, 8):
    grp = Group([0] * G, d)
    grp.build_csr(offsets, ids, vec)
    grp.set_stream(torch.cuda.current_stream().cuda_stream)
    gi, gd = grp.search(idx["parent"], qs[0], nprobe, k, "l2")
    grp.synchronize()
    same = bool((gi == ref[0]).all().item() and (gd.view(torch.int32) == ref[1].view(torch.int32)).all().item())
    h, a = measure(lambda b: grp.search(idx["parent"], qs[b], nprobe, k, "l2", out=out), grp.synchronize)
    _, _, tm = grp.search(idx["parent"], qs[0], nprobe, k, "l2", timing=True)
    print(json.dumps({"path": f"group of {G} members on device 0", "host_enqueue_ms": h, "ms_per_search": a, "bits_equal_one_store": same,
                      "lead_phases_ms": {kk: round(v, 4) for kk, v in tm.items() if kk.endswith("_ms")}}), flush=True)
    grp.close()
