"""Microbenchmark of the scan kernel alone (qk_scan with given partition lists) under env-var variants.
Run on the GPU box:  python scripts/scan_probe.py [nvec] [nlist] [P]"""
import os, sys, time, subprocess, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    nvec = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    from quake_amd.capi import Context, Store
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(0)
    d = int(os.environ.get("PROBE_D", "128"))
    kk = int(os.environ.get("PROBE_K", "10"))
    metric = os.environ.get("PROBE_METRIC", "l2")
    x = torch.randn(nvec, d, generator=g, device=dev)
    # partition sizes: lognormal-ish spread around nvec/nlist, like a k-means build
    w = torch.exp(0.5 * torch.randn(nlist, generator=g, device=dev))
    sizes = torch.clamp((w / w.sum() * nvec).long(), min=1)
    sizes[0] += nvec - sizes.sum()
    offsets = np.zeros(nlist + 1, np.int64); offsets[1:] = np.cumsum(sizes.cpu().numpy())
    ids = torch.arange(nvec, device=dev)
    s = Store(ctx, d); s.build_csr(offsets, ids, x)
    Q = 1024
    q = torch.randn(Q, d, generator=g, device=dev)
    pids = torch.stack([torch.randperm(nlist, generator=g, device=dev)[:P] for _ in range(Q)]).contiguous()
    uniq = torch.unique(pids)
    bytes_alg = int(sizes[uniq].sum().item()) * d * 4
    ctx.set_timing(0)
    for _ in range(3):
        ctx.scan(s, q, pids, kk, metric)
    ctx.set_timing(2)
    for _ in range(20):
        ctx.scan(s, q, pids, kk, metric)
    t = ctx.read_timing()
    ms = t["scan_ms"] / t["calls"]
    print(json.dumps({"mode": os.environ.get("QK_SCAN_MODE", "0"), "wpc": os.environ.get("QK_SCAN_WAVES_PER_CU", "auto"),
                      "d": d, "k": kk, "metric": metric, "P": P, "scan_ms": round(ms, 4), "GBs": round(bytes_alg / ms / 1e6, 1), "group_ms": round(t["group_ms"] / t["calls"], 4),
                      "merge_ms": round(t["merge_ms"] / t["calls"], 4)}), flush=True)

if __name__ == "__main__":
    main()
