"""What ONE rank does per step of bench.py --gpus N (without the collectives): coarse for its slice of the batch against the
N*4096 replicated centroids, scan of the whole batch of N*1024 queries over its own 4096 lists (queries whose partition lives
on another rank find an empty list), per-rank merge of N*1024 rows, final merge of its slice.  python scripts/rank_step_probe.py [N]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context, Store

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, d, nlist, k, per = 10_000_000, 128, 4096, 10, 1024
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_squared_l2(True)
x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev)
centroids, assign, _ = ctx.kmeans(x, nlist, "l2", niter=5, seed=1234)
order = torch.argsort(assign, stable=True)
counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
G = nlist * N
offsets = np.zeros(G + 1, np.int64)
offsets[1:nlist + 1] = np.cumsum(counts)
offsets[nlist + 1:] = offsets[nlist]
store = Store(ctx, d)
store.build_csr(offsets, order.contiguous(), x[order].contiguous())
g = torch.Generator(device=dev).manual_seed(7)
# the other ranks' centroids: same distribution, elsewhere in space (their queries never pick ours)
others = [centroids + 100.0 * (r + 1) for r in range(N - 1)]
cent_all = torch.cat([centroids] + others, 0).contiguous()
parent = Store(ctx, d)
parent.build_csr(np.array([0, G], np.int64), torch.arange(G, device=dev), cent_all)
q_local = B.gen_queries(per, cent_true, seed=2, device=dev)
q_all = torch.cat([q_local] + [q_local + 100.0 * (r + 1) for r in range(N - 1)], 0).contiguous()
Q = q_all.shape[0]
out_i = torch.empty((Q, k), dtype=torch.int64, device=dev)
out_d = torch.empty((Q, k), dtype=torch.float32, device=dev)


def step():
    pl = ctx.coarse(parent, q_all[:per], 1, "l2")[0]          # this rank's slice
    pids = ctx.coarse(parent, q_all, 1, "l2")[0] if False else torch.cat([pl] + [pl + nlist * (r + 1) for r in range(N - 1)], 0)
    ids, keys = ctx.scan_into(store, q_all, pids.contiguous(), k, "l2", (out_i, out_d))
    xi = ids.view(N, per, k)   # stand-in for the all-to-all output (same shapes)
    xk = keys.view(N, per, k)
    return ctx.merge_topk(xi, xk, "l2")


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {"N": N, "step_ms_without_collectives": round(timeit(step), 4),
       "coarse_ms": round(timeit(lambda: ctx.coarse(parent, q_all[:per], 1, "l2")), 4)}
pl = ctx.coarse(parent, q_all[:per], 1, "l2")[0]
pids = torch.cat([pl] + [pl + nlist * (r + 1) for r in range(N - 1)], 0).contiguous()
if os.environ.get("PROBE_REMOTE") == "skip":      # remote pairs marked "skip" instead of naming an empty list
    pids = torch.cat([pl] + [torch.full_like(pl, -1) for r in range(N - 1)], 0).contiguous()
if os.environ.get("PROBE_REMOTE") == "first":     # the local queries last in the batch
    pids = torch.cat([pl + nlist * (r + 1) for r in range(N - 1)] + [pl], 0).contiguous()
    q_all = torch.cat([q_all[per:], q_all[:per]], 0).contiguous()
out["scan_ms"] = round(timeit(lambda: ctx.scan_into(store, q_all, pids, k, "l2", (out_i, out_d))), 4)
out["merge_ms"] = round(timeit(lambda: ctx.merge_topk(out_i.view(N, per, k), out_d.view(N, per, k), "l2")), 4)
print(json.dumps(out), flush=True)
ctx.set_timing(1)
_, _, tm = ctx.scan(store, q_all, pids, k, "l2", timing=True)
ctx.set_timing(0)
print(json.dumps({"scan_phases_ms": {kk: round(v, 4) for kk, v in tm.items() if kk.endswith("_ms")}}), flush=True)
