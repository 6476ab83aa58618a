"""Every multi-GPU branch on RCCL before an 8-GPU node runs it: one process, `init_process_group("nccl", world_size=1)` on
GPU 0 and QUAKE_FORCE_COLLECTIVES=1, so that the exchange paths execute instead of being short-circuited for one rank --
device all_gather_into_tensor, async all_to_all_single on device buffers, all_reduce / ordered all-gather of the k-means
partials, broadcasts and uneven all-to-all of the maintenance routing, the library's kernels ordered with the collectives on
torch's stream (qk_ctx_set_stream).  With one rank the sharded result must equal the plain C-ABI result bit for bit
(query_coordinator.cpp:243-469, partition_manager.cpp:557-603).  Each scenario runs in its own interpreter: a process group
is global state, and a hang must not take the suite with it."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
from quake_amd.sharded import collectives_active
assert collectives_active(dist, 1)
'''


def _run(body, timeout=600):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), QUAKE_FORCE_COLLECTIVES="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = "ROOT = %r\n" % ROOT + PRELUDE + body + "\ndist.barrier()\ndist.destroy_process_group()\nprint('RCCL-WORLD1-OK')\n"
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0 or "RCCL-WORLD1-OK" not in r.stdout:
        err = r.stderr
        i = err.rfind("Traceback")
        raise AssertionError("RCCL world-1 scenario failed:\n" + (err[i:i + 3000] if i >= 0 else err[-3000:]))
    return r.stdout


def test_sharded_search_both_layouts_on_rccl():
    _run(r'''
from helpers import make_ivf, make_queries
from quake_amd.capi import Context, Store
from quake_amd.sharded import GpuEngine, ShardedIndex
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for metric in ("l2", "ip"):
    d, nlist = 64, 48
    ivf = make_ivf(60000, d, nlist, seed=81, metric=metric, empty=(5,))
    q = make_queries(256, d, seed=82, like=ivf["x"], metric=metric)
    qd = torch.from_numpy(q).cuda()
    a, b = Store(ctx, d), Store(ctx, d)
    a.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"]); b.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), ivf["centroids"])
    eng = GpuEngine(ctx, parent, a, metric)
    for result in ("all", "owner"):
        sh = ShardedIndex(eng, dist, 1, 0, result=result)
        assert not sh._stage_host           # device buffers straight into the collectives
        for rep in range(3):                # (buffers are reused from the second call on)
            for nprobe, k in [(1, 10), (8, 10), (48, 100)]:
                gi, gd = sh.search(qd, nprobe, k)
                ri, rd = ctx.search(parent, b, qd, nprobe, k, metric)
                torch.cuda.synchronize()
                assert torch.equal(gi, ri), (metric, result, nprobe, k)
                assert torch.equal(gd.view(torch.int32), rd.view(torch.int32)), (metric, result, nprobe, k)
        assert sh._g_pids is not None and sh._g_pids.is_cuda
        assert (sh._x_recv if result == "owner" else sh._g_ids).is_cuda  # (owner: ONE packed all-to-all, uint8 [G][block])
    # a side stream: the library follows torch's current stream, the collectives order themselves against it
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx.set_stream(st.cuda_stream)
        sh = ShardedIndex(eng, dist, 1, 0, result="owner")
        gi, gd = sh.search(qd, 8, 10)
        st.synchronize()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ri, rd = ctx.search(parent, b, qd, 8, 10, metric)
    torch.cuda.synchronize()
    assert torch.equal(gi, ri) and torch.equal(gd.view(torch.int32), rd.view(torch.int32))
''')


def test_sharded_kmeans_ordered_and_all_reduce_on_rccl():
    _run(r'''
from quake_amd.capi import Context
from quake_amd.sharded import sharded_kmeans
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(83)
for metric in ("l2", "ip"):
    cent = rng.standard_normal((32, 48)).astype(np.float32)
    x = torch.from_numpy((cent[rng.integers(0, 32, 40000)] + 0.4 * rng.standard_normal((40000, 48))).astype(np.float32)).cuda()
    rc, ra, _ = ctx.kmeans(x, 64, metric, niter=3, seed=9)       # with one rank the sharded k-means IS qk_kmeans
    for ordered in (True, False):                                # all-gather + rank-order sum / one all_reduce
        # (IP: sharded_kmeans normalises the rows it is handed IN PLACE, like qk_kmeans does with its own copy)
        c, a = sharded_kmeans(ctx, dist, x.clone(), 64, metric, niter=3, seed=9, rank=0, world=1, ordered=ordered)
        torch.cuda.synchronize()
        assert torch.equal(c.view(torch.int32), rc.view(torch.int32)), (metric, ordered)
        assert torch.equal(a, ra), (metric, ordered)
''')


def test_sharded_quake_index_build_update_maintenance_on_rccl():
    _run(r'''
import quake_amd as quake
from quake_amd.index import QuakeIndex
from quake_amd.sharded_maintenance import ShardedQuakeIndex
from test_sharded_maintenance_gpu import _corpus, _plain, _apply, _same_lists, _policy_params, _cost, _search_plain
metric = "l2"
ivf = _corpus(metric)
nlist, d = ivf["nlist"], ivf["d"]
ix = _plain(ivf, metric)
sh = ShardedQuakeIndex(_plain(ivf, metric), dist, 1, 0)
assert sh.partitions.comm.active and sh.partitions.comm.device.type == "cuda"
ops = [("split", [3, 8]), ("delete", [5, 10]), ("refine", [0, 1, 2, nlist, nlist + 1], 3), ("refine", [4, 6, nlist + 2], 0)]
_apply(sh.partitions, False, ops)
_apply(ix, True, ops)
_same_lists(sh, ix, 0, 1)
rng = np.random.default_rng(5)
nx = (ivf["x"][rng.integers(0, len(ivf["x"]), 500)] + 0.01 * rng.standard_normal((500, d))).astype(np.float32)
nid = np.arange(900000, 900500, dtype=np.int64)
assert sh.add(nx, nid) == 500
ix.add(torch.from_numpy(nx), torch.from_numpy(nid))
rm = np.concatenate([ivf["ids"][:200], nid[:50]])
assert sh.remove(rm) == 250
ix.remove(torch.from_numpy(rm))
assert sh.ntotal() == ix.ntotal()
qd = torch.from_numpy(ivf["x"][:128].copy()).cuda()
for nprobe, k in [(1, 10), (4, 10), (sh.nlist(), 20)]:
    gi, gd = sh.search(qd, nprobe, k)
    ri, rd = _search_plain(ix, qd, nprobe, k)
    torch.cuda.synchronize()
    assert torch.equal(gi.cpu(), ri.cpu()) and torch.equal(gd.cpu().view(torch.int32), rd.cpu().view(torch.int32)), (nprobe, k)
# the policy: same decisions, same index afterwards (hit tracking through the all-gathered partition lists)
for s_ in (sh, ix):
    s_.initialize_maintenance_policy(_policy_params(256, 2), cost_estimator=_cost(d))
    s_.track_hits = True
q2 = torch.from_numpy(ivf["x"][:256].copy()).cuda()
sh.search(q2, 2, 10)
_search_plain(ix, q2, 2, 10)
ta, tb = sh.maintenance(), ix.maintenance()
assert (ta.n_splits, ta.n_deletes) == (tb.n_splits, tb.n_deletes)
_same_lists(sh, ix, 0, 1, ordered=False)
gi, gd = sh.search(q2, 4, 10)
ri, rd = _search_plain(ix, q2, 4, 10)
assert torch.equal(gi.cpu(), ri.cpu()) and torch.equal(gd.cpu().view(torch.int32), rd.cpu().view(torch.int32))
# sharded BUILD over the collectives: exhaustive probing of the built index == exact k-NN
xb = torch.from_numpy(ivf["x"][:20000].copy())
shb = ShardedQuakeIndex.build(dist, 1, 0, xb, np.arange(20000, dtype=np.int64), 16, metric, niter=3, seed=3)
assert shb.ntotal() == 20000
gi, gd = shb.search(qd, shb.nlist(), 5)
d2 = torch.cdist(qd.double(), xb.cuda().double()) ** 2
ref = torch.topk(d2, 5, dim=1, largest=False).indices
assert (gi.sort(dim=1).values == ref.sort(dim=1).values).float().mean().item() > 0.999
''')


def test_bench_sharded_path_on_one_rank():
    """bench.py forced down its N > 1 path with one rank on RCCL (sharded k-means, row routing with all_to_all_single, sharded
    search with result='owner', all-gathered ground truth): one JSON line, recall as good as the single-GPU path's"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               QUAKE_FORCE_COLLECTIVES="1", QUAKE_BENCH_FORCE_SHARDED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--settle", "5",
                        "--nvec-sharded", "400000", "--nlist-sharded", "256", "--batch-sharded", "256", "--no-cpu", "--no-extra"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    import json
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["recall_at_k"] >= 0.9
    assert "lists sharded by number over 1 ranks" in line["config"]["workload"]


def test_bench_refuses_a_rank_count_mismatch():
    """`bench.py --gpus N` must exit non-zero with a clear message when the launcher gave it a different number of ranks"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)


@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_bench_starts_its_own_ranks(nranks):
    """plain `python bench.py --gpus N` (no launcher, WORLD_SIZE unset): bench.py starts the N ranks itself; over gloo they share
    the one GPU of the test box.  One JSON line from rank 0, n_gpus = N, recall as good as the single-GPU path's (N = 4: the
    list p % N ownership, the N-block packed exchange and the rank-ordered reductions beyond two ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QUAKE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "5", "--warmup", "2", "--settle", "5",
                        "--nvec-sharded", "300000", "--nlist-sharded", "128", "--batch-sharded", "128", "--no-cpu", "--no-extra"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    import json
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == nranks and line["config"]["recall_at_k"] >= 0.9 and line["config"]["batch"] == 128 * nranks
    assert f"lists sharded by number over {nranks} ranks" in line["config"]["workload"]
    # the N > 1 line describes itself: one roofline per rank, the devices the ranks sat on, the bytes of the step's two collectives
    # (8 ranks = the rank count of BASELINE.json configs[3] / [4]; on this box they share one GPU: distinct_devices == 1)
    assert [r_["rank"] for r_ in line["per_rank"]] == list(range(nranks))
    assert all(r_["roofline"]["bound"] == "hbm" and r_["roofline"]["achieved"] > 0 and r_["vectors"] > 0 for r_ in line["per_rank"])
    assert line["rccl_ranks_seen"] == {"backend": "gloo", "ranks": nranks, "distinct_devices": 1, "ranks_described": nranks}
    kk, per, k = min(line["config"]["nprobe"], 128 * nranks), 128, line["config"]["k"]
    assert line["exchange"]["all_gather_list_numbers"] == {"send_bytes_per_rank": per * kk * 8, "recv_bytes_per_rank": per * nranks * kk * 8}
    assert line["exchange"]["all_to_all_topk"]["send_bytes_per_rank"] == nranks * ((per * k * 12 + 15) // 16 * 16)


def test_bench_refuses_more_ranks_than_gpus_on_rccl():
    """`python bench.py --gpus 2` on a one-GPU box with the RCCL backend: non-zero exit and a message, nothing started"""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "QUAKE_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "GPU(s)" in r.stderr
