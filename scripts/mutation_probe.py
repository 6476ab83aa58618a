"""Small and large add / remove calls on a 10M x 128 index through QuakeIndex (python mirror): ms per call (synchronised).
python scripts/mutation_probe.py"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
if os.environ.get("MIRROR", "python") == "compiled":
    import quake
else:
    import quake_amd as quake

n, d, nlist = 10_000_000, 128, 4096
dev = torch.device("cuda", 0)
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev)
idx = quake.QuakeIndex()
bp = quake.IndexBuildParams()
bp.nlist = nlist
t0 = time.perf_counter()
idx.build(x if quake.__name__ != "quake" else x.cpu(), torch.arange(n, device=dev) if quake.__name__ != "quake" else torch.arange(n), bp)
torch.cuda.synchronize()
print(json.dumps({"build_s": round(time.perf_counter() - t0, 2)}), flush=True)
g = torch.Generator(device=dev).manual_seed(5)
next_id = n
for m in (1, 16, 256, 4096, 65536, 1048576):
    ts_a, ts_r = [], []
    for rep in range(5):
        v = cent[torch.randint(0, nlist, (m,), generator=g, device=dev)] + 0.3 * torch.randn(m, d, generator=g, device=dev)
        ids = torch.arange(next_id, next_id + m, device=dev)
        if quake.__name__ == "quake":  # (the compiled mirror takes the reference's CPU tensors)
            v, ids = v.cpu(), ids.cpu()
        next_id += m
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ti = idx.add(v, ids)
        torch.cuda.synchronize()
        ts_a.append(time.perf_counter() - t0)
        if rep == 0:
            first = {"validate_us": ti.input_validation_time_us, "find_partition_us": ti.find_partition_time_us, "modify_us": ti.modify_time_us,
                     "store_events": ({k_: v_ for k_, v_ in idx._store.counters().items() if v_} if hasattr(idx, "_store") else None)}
        t0 = time.perf_counter()
        idx.remove(ids)
        torch.cuda.synchronize()
        ts_r.append(time.perf_counter() - t0)
    print(json.dumps({"vectors": m, "add_ms_p50": round(1e3 * sorted(ts_a)[2], 3), "remove_ms_p50": round(1e3 * sorted(ts_r)[2], 3),
                      "add_ms_first": round(1e3 * ts_a[0], 3), "remove_ms_first": round(1e3 * ts_r[0], 3), "first_add": first}), flush=True)
