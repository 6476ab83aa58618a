"""The two mirrors of the reference's API on the bench index (10M x 128, 4096 lists): ms per QuakeIndex.search call with CPU tensors
in and out (what a program written against the reference does), 1024 queries and 1 query, nprobe 1 / 8; build time beside it.
python scripts/mirror_probe.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B

n, d, nlist = 10_000_000, 128, 4096
dev = torch.device("cuda", 0)
x, cent = B.gen_mixture(n, d, nlist, seed=1, device=dev)
xc = x.cpu()
ids = torch.arange(n)
q = B.gen_queries(1024, cent, seed=2, device=dev).cpu()
del x
for name in ("compiled", "python"):
    if name == "compiled":
        import quake as Q
    else:
        import quake_amd as Q
    idx = Q.QuakeIndex()
    bp = Q.IndexBuildParams()
    bp.nlist = nlist
    t0 = time.perf_counter()
    idx.build(xc, ids, bp)
    out = {"mirror": name, "build_s_from_cpu_tensors": round(time.perf_counter() - t0, 2)}
    for nq in (1024, 1):
        for nprobe in (1, 8):
            sp = Q.SearchParams()
            sp.k = 10
            sp.nprobe = nprobe
            sp.batched_scan = True
            qq = q[:nq].contiguous()
            for _ in range(5):
                r = idx.search(qq, sp)
            ts = []
            for _ in range(30):
                t1 = time.perf_counter()
                r = idx.search(qq, sp)
                ts.append(time.perf_counter() - t1)
            ts.sort()
            out[f"search_ms_q{nq}_nprobe{nprobe}"] = round(1e3 * ts[len(ts) // 2], 3)
    print(json.dumps(out), flush=True)
    del idx
