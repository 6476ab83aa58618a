"""Phase stamps of the one-launch small-batch search (k_search_small) on the configs[0] shape: probe build of qk_small.hip
(QUAKE_HIP_LIB=quake_amd/lib/libquake_hip_smallprobe.so QK_SMALL_CLOCK=1)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from quake_amd.capi import Context, Store
dev = torch.device("cuda", 0)
ctx = Context(0)
x, cent = B.gen_ssift(1_000_000, dev, seed=1234)
q, _ = B.gen_ssift(64, dev, seed=4321, cent=cent)
idx = B.build_single(ctx, dev, x, 1024, "l2", 5, keep_host=False)
for Q in (1, 1, 1, 4):
    for i in range(3):
        ctx.search(idx["parent"], idx["store"], q[i * Q:(i + 1) * Q].contiguous(), 10, 10, "l2")
    ctx.synchronize()
