"""Where an add() / remove() call spends its time: QuakeIndex on 5M x 128, batches of 100k / 500k rows from host or device
tensors.  python scripts/add_probe.py"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
import quake_amd as quake

dev = torch.device("cuda", 0)
n, d = 5_000_000, 128
x, cent = B.gen_mixture(n + 2_000_000, d, 2000, seed=1, device=dev)
ix = quake.QuakeIndex()
bp = quake.IndexBuildParams()
bp.nlist, bp.metric, bp.niter = 2000, "l2", 3
ix.build(x[:n], torch.arange(n, dtype=torch.int64), bp)
nxt = n
for where in ("device", "host", "host"):
    for m in (100_000, 500_000):
        xs = x[nxt:nxt + m]
        ids = torch.arange(nxt, nxt + m, dtype=torch.int64)
        if where == "host":
            xs = xs.cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = ix.add(xs, ids)
        torch.cuda.synchronize()
        t_add = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        rinfo = ix.remove(ids)
        torch.cuda.synchronize()
        t_rm = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"from": where, "rows": m, "add_ms": round(t_add, 2), "validate_ms": round(info.input_validation_time_us / 1e3, 2),
                          "find_partition_ms": round(info.find_partition_time_us / 1e3, 2), "modify_ms": round(info.modify_time_us / 1e3, 2),
                          "remove_ms": round(t_rm, 2), "remove_validate_ms": round(rinfo.input_validation_time_us / 1e3, 2),
                          "remove_modify_ms": round(rinfo.modify_time_us / 1e3, 2)}), flush=True)
        nxt += m
