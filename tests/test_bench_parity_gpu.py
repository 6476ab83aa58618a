"""HIP path vs the CPU oracle, BIT-EXACT, at BASELINE.json's sizes -- the bench batches themselves, not scaled-down stand-ins:

  configs[1]  10M x 128 L2, nlist 4096, 1024 queries, k 10   (the bench's own corpus / index / batch; nprobe 1, 8 and 32: k_scan,
              and the mixed work sequence of the row-per-lane scan with its bf16 prefilter) and the bench's second corpus
              (latent dimension 10, nprobe 16)
  configs[2]  10M x 768 IP, nlist 4096, 1024 queries, k 100  (nprobe 1 and 4)
  configs[0]  S-SIFT 1M x 128 integer-valued, nlist 1024, nprobe 10, k 10: batch = 1 and batch = 1000 against the SERIAL
              oracle (scan_list's direct-form distances) -- on integer data the direct and the expanded forms are both exact,
              so the reference's default path and the HIP path must agree bit for bit
  configs[3]  one rank's slice of the 8-GPU shape: 65536 replicated centroids, the 8192 lists p = 0 (mod 8) resident
              (12.5M vectors), 4096 queries, k 10, nprobe 8

The oracle's batched path (batched_serial_scan: expanded L2 / dot on k-ordered fmaf chains, (key, id) order) is the
canonical comparator: ids equal and float32 distances equal as uint32.  Mirrors the reference's own large checks
(test/cpp/list_scanning.cpp:432-562, query_coordinator.cpp:201-254) at the benchmark scale."""
import numpy as np
import pytest
import torch

import bench as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_form_feedback(False)  # the forms asserted below are the static rule's
    yield c
    c.close()


def _assert_same(gi, gd, oi, od, what):
    gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
    bad = np.argwhere(gi != oi)
    assert bad.size == 0, f"{what}: {len(bad)} ids differ, first at {bad[0]}: gpu {gi[tuple(bad[0])]} oracle {oi[tuple(bad[0])]}"
    assert (gd.view(np.uint32) == od.view(np.uint32)).all(), f"{what}: distance bits differ"


def _bench_case(ctx, n, d, nlist, k, metric, nprobes, sigma=0.3, manifold=0, forms=None):
    import oracle as O
    dev = torch.device("cuda", 0)
    if manifold:
        x, basis = B.gen_manifold(n, d, seed=1, device=dev, latent=manifold)
    else:
        x, cent_true = B.gen_mixture(n, d, nlist, seed=1, device=dev, sigma=sigma, unit=metric == "ip")
    idx = B.build_single(ctx, dev, x, nlist, metric, niter=5, keep_host=True)
    del x
    if manifold:
        q = B.gen_manifold(1024, d, seed=2, device=dev, latent=manifold, basis=basis)[0]  # the bench's batch 0
    else:
        q = B.gen_queries(1024, cent_true, seed=2, device=dev, sigma=sigma, unit=metric == "ip")  # the bench's batch 0
    hv, hi, ho, hc = idx["host"]
    qh = q.cpu().numpy()
    for nprobe in nprobes:
        gi, gd = ctx.search(idx["parent"], idx["store"], q, nprobe, k, metric)
        torch.cuda.synchronize()
        if forms:
            assert ctx.last_scan_kernel() == forms[nprobe], (nprobe, ctx.last_scan_kernel())
        oi, od = O.search(qh, hc, hv, hi, ho, nprobe, k, metric, batched_scan=True, num_threads=O.max_threads())
        _assert_same(gi, gd, oi, od, f"{n}x{d} {metric} nprobe={nprobe} k={k}")
    idx["store"].close()
    idx["parent"].close()
    torch.cuda.empty_cache()


def test_configs1_bench_batch_bit_exact(ctx):
    _bench_case(ctx, 10_000_000, 128, 4096, 10, "l2", (1, 4, 8, 12, 32),
                forms={1: "k_scan", 4: "k_scan_rl (mixed)", 8: "k_scan_rl (mixed)", 12: "k_scan_rl (mixed)", 32: "k_scan_rl (mixed)"})


def test_configs1_hard_corpus_bit_exact(ctx):
    """the bench's second workload (`workloads.hard`: x = zA + noise, latent dimension 10 -- no cluster structure, recall 0.9
    needs nprobe 16, every list is probed by several queries of the batch)"""
    _bench_case(ctx, 10_000_000, 128, 4096, 10, "l2", (16,), manifold=10, forms={16: "k_scan_rl (mixed)"})


def test_configs2_bench_batch_bit_exact(ctx):
    _bench_case(ctx, 10_000_000, 768, 4096, 100, "ip", (1, 4))


def test_configs0_ssift_batch1_and_batch1000_vs_serial_oracle(ctx):
    import oracle as O
    dev = torch.device("cuda", 0)
    n, nlist, nprobe, k = 1_000_000, 1024, 10, 10
    x, cent = B.gen_ssift(n, dev, seed=1234)
    assert float(x.min()) >= 0 and float(x.max()) <= 218 and bool((x == x.round()).all())
    q, _ = B.gen_ssift(1000, dev, seed=4321, cent=cent)
    idx = B.build_single(ctx, dev, x, nlist, "l2", niter=5, keep_host=True)
    del x
    hv, hi, ho, hc = idx["host"]
    qh = q.cpu().numpy()
    # the reference default: serial_scan (scan_list: direct-form sqrt(sum (x-y)^2) per row), one thread, batch = 1
    si, sd = O.search(qh, hc, hv, hi, ho, nprobe, k, "l2", batched_scan=False)
    bi, bd = O.search(qh, hc, hv, hi, ho, nprobe, k, "l2", batched_scan=True)
    assert (si == bi).all() and (sd.view(np.uint32) == bd.view(np.uint32)).all()  # integer data: both forms exact
    gi, gd = ctx.search(idx["parent"], idx["store"], q, nprobe, k, "l2")  # batch = 1000
    torch.cuda.synchronize()
    _assert_same(gi, gd, si, sd, "S-SIFT batch=1000")
    for i in range(0, 1000, 31):  # batch = 1
        g1, d1 = ctx.search(idx["parent"], idx["store"], q[i:i + 1].contiguous(), nprobe, k, "l2")
        torch.cuda.synchronize()
        _assert_same(g1, d1, si[i:i + 1], sd[i:i + 1], f"S-SIFT batch=1 query {i}")
    # host buffers (what the reference's CPU tensors are), batch = 1
    for i in (0, 499, 999):
        g1, d1 = ctx.search(idx["parent"], idx["store"], qh[i:i + 1], nprobe, k, "l2")
        assert (g1 == si[i:i + 1]).all() and (d1.view(np.uint32) == sd[i:i + 1].view(np.uint32)).all()
    idx["store"].close()
    idx["parent"].close()
    torch.cuda.empty_cache()


def test_configs3_one_rank_slice_bit_exact(ctx):
    """rank 0 of 8: all 65536 centroids replicated, only lists p % 8 == 0 hold vectors (12.5M), 4096 queries."""
    import oracle as O
    from quake_amd.capi import Store
    dev = torch.device("cuda", 0)
    d, nlist, world, n_local, Q, k, nprobe = 128, 65536, 8, 12_500_000, 4096, 10, 8
    g = torch.Generator(device=dev).manual_seed(31)
    cent = torch.randn(nlist, d, generator=g, device=dev)
    own = torch.arange(0, nlist, world, device=dev)  # the lists of rank 0
    comp = own[torch.randint(0, own.shape[0], (n_local,), generator=g, device=dev)]  # fixed assignment: the generating centre
    x = torch.empty(n_local, d, device=dev)
    for i0 in range(0, n_local, 1 << 20):
        m = min(1 << 20, n_local - i0)
        x[i0:i0 + m] = cent[comp[i0:i0 + m]] + 0.3 * torch.randn(m, d, generator=g, device=dev)
    order = torch.argsort(comp, stable=True)
    counts = torch.bincount(comp, minlength=nlist).cpu().numpy().astype(np.int64)
    offsets = np.zeros(nlist + 1, np.int64)
    offsets[1:] = np.cumsum(counts)
    xs, ids = x[order].contiguous(), (order * world).contiguous()  # ids = global numbers of this rank's vectors
    del x, comp, order
    store = Store(ctx, d)
    store.build_csr(offsets, ids, xs)
    parent = Store(ctx, d)
    parent.build_csr(np.array([0, nlist], np.int64), torch.arange(nlist, device=dev), cent.contiguous())
    q = B.gen_queries(Q, cent, seed=32, device=dev)
    gi, gd = ctx.search(parent, store, q, nprobe, k, "l2")
    torch.cuda.synchronize()
    oi, od = O.search(q.cpu().numpy(), cent.cpu().numpy(), xs.cpu().numpy(), ids.cpu().numpy(), offsets, nprobe, k, "l2",
                      batched_scan=True)
    _assert_same(gi, gd, oi, od, "configs[3] rank slice")
    # most probed lists live on other ranks: the local answer is partial for many queries, padded with -1
    assert (oi == -1).any() and (oi >= 0).any()
    store.close()
    parent.close()
    torch.cuda.empty_cache()
