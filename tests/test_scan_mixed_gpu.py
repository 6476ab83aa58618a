"""The mixed work sequence of the row-per-lane scan (qk_scan_rl.hip, HOT form): lists probed by >= 13 queries of the batch
are scanned as dense workgroup items -- a bf16 prefilter on v_mfma_f32_16x16x32_bf16 with a one-sided error bound, the exact
chains on v_mfma_f32_16x16x4_f32 for every row tile that could still hold a candidate -- the others by the per-wave walk: one
launch, one record format, the same bits as the oracle's batched path (query_coordinator.cpp:675-799,
list_scanning.h:313-366).  The host turns the form on from three probing queries per list (batch average) on, on indexes whose
lists average >= 1400 rows; a list is hot with >= 13 probing queries AND >= 512 rows."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    c.set_form_feedback(False)  # this module pins WHICH form answers: the static rule alone (feedback: test_scan_feedback_gpu.py)
    yield c
    c.close()


def make_sized_ivf(sizes, d, seed, metric="l2", integer=False):
    """lists of the given sizes: Gaussian blobs around random centres (or SIFT-like integers: exact ties)"""
    rng = np.random.default_rng(seed)
    nlist = len(sizes)
    if integer:
        cent = rng.integers(0, 40, size=(nlist, d)).astype(np.float32)
    else:
        cent = rng.standard_normal((nlist, d)).astype(np.float32)
    vecs, offsets = [], np.zeros(nlist + 1, np.int64)
    for p, n in enumerate(sizes):
        if integer:
            v = cent[p] + rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        else:
            v = (cent[p] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
        vecs.append(v)
        offsets[p + 1] = offsets[p] + n
    x = np.ascontiguousarray(np.concatenate(vecs, 0), np.float32)
    if metric == "ip":
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6)
    ids = rng.permutation(x.shape[0]).astype(np.int64) + 1000
    return dict(vecs=x, ids=ids, offsets=offsets, centroids=cent, d=d, nlist=nlist)


def build(ctx, ivf):
    from quake_amd.capi import Store
    s = Store(ctx, ivf["d"])
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    return s


def skewed_pids(Q, P, nlist, hot, rng):
    """[Q, P] list numbers without repeats inside a row: list h of `hot` = {list: cnt} is probed by exactly cnt queries"""
    pids = np.full((Q, P), -1, np.int64)
    fill = np.zeros(Q, np.int64)
    for p, cnt in hot.items():
        rows = rng.permutation(Q)[:cnt]
        for r in rows:
            if fill[r] < P:
                pids[r, fill[r]] = p
                fill[r] += 1
    cold = np.array([p for p in range(nlist) if p not in hot])
    for r in range(Q):
        need = P - fill[r]
        if need > 0:
            pids[r, fill[r]:] = rng.choice(cold, size=need, replace=False)
    return pids


def check(ctx, s, ivf, q, pids, k, metric, want_form="k_scan_rl (mixed)"):
    gi, gd = ctx.scan(s, q, pids, k, metric)
    assert ctx.last_scan_kernel() == want_form
    oi, od = O.batched_serial_scan(q, ivf["vecs"], ivf["ids"], ivf["offsets"], pids, k, metric)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("d,k", [(128, 10), (100, 32), (32, 1), (64, 17)])
def test_hot_and_cold_lists_one_launch(ctx, metric, d, k):
    rng = np.random.default_rng(11)
    nlist = 400
    sizes = rng.integers(1400, 1900, size=nlist)
    sizes[390:] = 0
    # hot lists of every shape: many row ranges, a ragged last tile, exactly / just under the row threshold; lists shared by
    # many queries but too short for an item (17 rows, 5 rows, 1 row, fewer than k): passes of the per-wave walk
    sizes[:8] = [6000, 1300, 17, 512, 5, 1, 2500, 511]
    ivf = make_sized_ivf(sizes, d, seed=12, metric=metric)
    s = build(ctx, ivf)
    Q, P = 700, 6
    q = ivf["vecs"][rng.integers(0, ivf["vecs"].shape[0], size=Q)] + 0.1 * rng.standard_normal((Q, d)).astype(np.float32)
    q = np.ascontiguousarray(q, np.float32)
    # 700 queries on list 0 (6 blocks), 130 on list 1 (80 + 50), 129, 257, 48, 64 on short lists; 13 / 12 around the threshold
    hot = {0: 700, 1: 130, 2: 129, 3: 13, 4: 257, 5: 48, 6: 12, 7: 64}
    pids = skewed_pids(Q, P, nlist, hot, rng)
    pids[::9, -1] = -1
    check(ctx, s, ivf, q, pids, k, metric)
    s.close()


def test_hot_lists_with_exact_ties(ctx):
    """SIFT-like integer data: exact distance ties across the waves of an item and across items -> the (key, id) order"""
    rng = np.random.default_rng(21)
    nlist = 300
    sizes = rng.integers(1400, 1700, size=nlist)
    sizes[:3] = [3000, 900, 64]
    ivf = make_sized_ivf(sizes, 128, seed=22, integer=True)
    s = build(ctx, ivf)
    Q, P = 512, 8
    q = ivf["vecs"][rng.integers(0, 30000, size=Q)] + rng.integers(-2, 3, size=(Q, 128)).astype(np.float32)
    q = np.ascontiguousarray(q, np.float32)
    pids = skewed_pids(Q, P, nlist, {0: 512, 1: 200, 2: 40}, rng)
    for k in (10, 24):  # (d = 128 with k = 32: the per-wave form's LDS does not fit, k_scan serves it)
        check(ctx, s, ivf, q, pids, k, "l2")
    s.close()


def test_every_list_hot_and_none(ctx):
    rng = np.random.default_rng(31)
    nlist = 300
    sizes = rng.integers(1400, 1600, size=nlist)
    sizes[:4] = [2000, 1000, 600, 520]
    ivf = make_sized_ivf(sizes, 128, seed=32)
    s = build(ctx, ivf)
    Q = 600
    q = np.ascontiguousarray(ivf["vecs"][rng.integers(0, 20000, size=Q)] + 0.1 * rng.standard_normal((Q, 128)).astype(np.float32))
    # every pair lands on a hot list: the per-wave sequence is empty
    pids = np.tile(np.array([0, 1, 2, 3], np.int64), (Q, 1))
    check(ctx, s, ivf, q, pids, 10, "l2")
    # nobody shares a list with more than a few others: no hot item at all, the same kernel
    pids = (4 + ((np.arange(Q)[:, None] * 4 + np.arange(4)[None, :]) * 7919) % (nlist - 4)).astype(np.int64)  # round robin
    assert np.bincount(pids.ravel()).max() < 13 and all(len(set(r)) == 4 for r in pids.tolist())
    check(ctx, s, ivf, q, pids, 10, "l2")
    s.close()


def test_repeated_calls_reuse_counters(ctx):
    """the hot queue's counter lives in the per-call zeroed state: a second and third call must start from item 0"""
    rng = np.random.default_rng(41)
    nlist = 300
    sizes = rng.integers(1400, 1600, size=nlist)
    sizes[:2] = [4000, 700]
    ivf = make_sized_ivf(sizes, 64, seed=42)
    s = build(ctx, ivf)
    Q = 520
    q = np.ascontiguousarray(ivf["vecs"][rng.integers(0, 20000, size=Q)] + 0.1 * rng.standard_normal((Q, 64)).astype(np.float32))
    for it in range(3):
        pids = skewed_pids(Q, 4, nlist, {0: 520 - 40 * it, 1: 100 + it}, rng)
        check(ctx, s, ivf, q, pids, 10, "l2")
    s.close()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_hot_lists_value_range(ctx, metric):
    """the prefilter's bound over the whole float range: rows and queries with elements of 1e5-1e6 and of 1e-6 next to ordinary
    ones, in hot lists (bf16 keeps fp32's exponent range: no guard needed; an fp16 variant needed two)"""
    rng = np.random.default_rng(51)
    nlist = 300
    sizes = rng.integers(1400, 1600, size=nlist)
    sizes[:3] = [3000, 1200, 700]
    ivf = make_sized_ivf(sizes, 96, seed=52)
    x = ivf["vecs"]
    x[100:400] *= 3.0e5       # elements ~ 1e5..1e6: beyond 65504
    x[400:900] *= 1.0e-6      # fp16 subnormals / zeros
    x[3000:3300] *= 2.0e3     # large but representable
    if metric == "ip":
        pass                   # (not normalised: the bound must hold for any norms)
    s = build(ctx, ivf)
    Q = 600
    q = np.ascontiguousarray(x[rng.integers(0, 4900, size=Q)] * (1 + 0.05 * rng.standard_normal((Q, 96))).astype(np.float32))
    q[:20] *= 1.0e5
    q[20:40] *= 1.0e-5
    pids = skewed_pids(Q, 4, nlist, {0: 600, 1: 300, 2: 64}, rng)
    for k in (1, 10, 32):
        check(ctx, s, ivf, q, pids, k, metric)
    s.close()


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_prefilter_bound_with_unequal_norms(ctx, metric):
    """rows that differ from each other by less than a bf16 rounding step, queries of 100x their norm: the approximate order
    inside a hot list is scrambled, so every row the bound cannot exclude must be recomputed -- and the bound has to use BOTH
    norms under either metric (an IP bound built from the row norms alone lost candidates here)"""
    rng = np.random.default_rng(61)
    nlist = 300
    sizes = rng.integers(1400, 1600, size=nlist)
    sizes[:2] = [2500, 1800]
    ivf = make_sized_ivf(sizes, 64, seed=62)
    x = ivf["vecs"]
    base = rng.standard_normal((2, 64)).astype(np.float32)
    x[:2500] = base[0] + 1.0e-3 * rng.standard_normal((2500, 64)).astype(np.float32)
    x[2500:4300] = base[1] + 1.0e-3 * rng.standard_normal((1800, 64)).astype(np.float32)
    s = build(ctx, ivf)
    Q = 512
    q = (100.0 * (base[rng.integers(0, 2, size=Q)] + 0.05 * rng.standard_normal((Q, 64)))).astype(np.float32)
    pids = skewed_pids(Q, 4, nlist, {0: 512, 1: 400}, rng)
    for k in (10, 32):
        check(ctx, s, ivf, np.ascontiguousarray(q), pids, k, metric)
    s.close()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_prefilter_adversarial_at_the_kth_key(ctx, metric):
    """The one place a silent recall loss could hide behind green random tests (round-3 review): rows whose keys sit within ONE
    bf16 rounding step of the query's k-th best -- the approximate products of hundreds of rows are equal or out of order, so
    the prefilter's one-sided bound alone decides which row tiles get their exact chain -- in lists whose norms span 2^-20 ..
    2^+20 (the slack of the bound scales with |x|^2 + |y|^2).  Every hot list holds ~2000 rows of the form
    scale * (base + eps * noise) with eps = 2^-12 (a sixteenth of a bf16 step) around one base vector, probed by queries
    scale * (base + small offset): the exact order among the rows is decided 4-5 decimal digits below what bf16 resolves.
    Ids and distance bits must equal the oracle's, as everywhere."""
    rng = np.random.default_rng(97)
    d, nlist = 64, 300
    scales = [2.0 ** -20, 2.0 ** -10, 1.0, 2.0 ** 10, 2.0 ** 20]
    sizes = rng.integers(1400, 1600, size=nlist)
    sizes[:len(scales)] = 2000
    ivf = make_sized_ivf(sizes, d, seed=98)
    x = ivf["vecs"]
    bases = rng.standard_normal((len(scales), d)).astype(np.float32)
    for j, sc in enumerate(scales):
        lo, hi = int(ivf["offsets"][j]), int(ivf["offsets"][j + 1])
        x[lo:hi] = (sc * (bases[j] + 2.0 ** -12 * rng.standard_normal((hi - lo, d)))).astype(np.float32)
        x[lo:lo + 40] = x[lo + 40:lo + 80]          # exact duplicates on top: ties on the k-th key, (key, id) decides
    if metric == "ip":
        pass  # (raw inner products: rows of the same direction and nearly equal length -- the hardest case for the IP bound)
    s = build(ctx, ivf)
    Q = 640
    which = rng.integers(0, len(scales), size=Q)
    q = np.stack([scales[w] * (bases[w] + 2.0 ** -9 * rng.standard_normal(d)) for w in which]).astype(np.float32)
    # every query probes "its" scale's list (hot: ~128 probing queries each) + three cold lists
    pids = np.full((Q, 4), -1, np.int64)
    cold = np.arange(len(scales), nlist)
    for r in range(Q):
        pids[r, 0] = which[r]
        pids[r, 1:] = rng.choice(cold, size=3, replace=False)
    for k in (1, 10, 32):
        check(ctx, s, ivf, np.ascontiguousarray(q), pids, k, metric)
    s.close()
