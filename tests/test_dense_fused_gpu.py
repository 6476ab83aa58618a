"""The coarse step over a few thousand centroids in one launch (qk_dense_fused.hip: exact keys on fp32 MFMA stay in LDS, the k-th
smallest key of a slice by bisection on the key bits, slices merged by k_merge_slices) against the oracle's parent search
(query_coordinator.cpp:628-644 -> batched_scan_list, list_scanning.h:313-366): ids and float32 distance bits."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    yield c
    c.close()


def _parent(ctx, cent, ids=None):
    from quake_amd.capi import Store
    n, d = cent.shape
    p = Store(ctx, d)
    p.build_csr(np.array([0, n], np.int64), np.arange(n, dtype=np.int64) if ids is None else ids, cent)
    return p


def _check(ctx, parent, cent, q, k, metric, ids=None, form="k_dense_fused"):
    gp, gd = ctx.coarse(parent, q, k, metric)
    if form:
        forms = (form,) if isinstance(form, str) else form
        assert ctx.last_scan_kernel() in forms, (ctx.last_scan_kernel(), cent.shape, q.shape, k)
    op, od = O.coarse(q, cent, ids, k, metric)
    np.testing.assert_array_equal(gp, op)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("d", [128, 100, 64, 32, 16])
def test_fused_coarse_matches_oracle(ctx, metric, d):
    rng = np.random.default_rng(70 + d)
    for n in (1024, 1500, 2048, 3000, 4096):
        cent = rng.standard_normal((n, d)).astype(np.float32)
        if metric == "ip":
            cent /= np.linalg.norm(cent, axis=1, keepdims=True)
        parent = _parent(ctx, cent)
        for nq in (300, 1000, 1024, 2500):
            q = (cent[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
            if metric == "ip":
                q /= np.linalg.norm(q, axis=1, keepdims=True)
            for k in (2, 10, 32, 33, 64):
                # the form serves up to 2048 rows, 3072 at k = 32, 4096 beyond (fused_plan); the prefiltered form the rest (the
                # key-matrix form where that one has too few row groups for its bound: small batches with a large k)
                fused = n <= (4096 if k > 32 else 3072 if k == 32 else 2048)
                _check(ctx, parent, cent, q, k, metric, form="k_dense_fused" if fused else ("k_dense_pf", "k_dense"))
        parent.close()


def test_fused_ties_duplicates_and_arbitrary_ids(ctx):
    """exact duplicates straddling the k-th place of a slice and of the whole list (the (key, id) order decides; ids shuffled, above
    2^32 and unrelated to the row order), and a list of one vector repeated: every key of every slice ties"""
    rng = np.random.default_rng(111)
    base = rng.integers(0, 6, size=(300, 64)).astype(np.float32)     # small integers: exact arithmetic, many ties
    cent = base[rng.integers(0, 300, 2000)]
    ids = rng.permutation(2000).astype(np.int64) * 3 + (1 << 33)
    parent = _parent(ctx, cent, ids)
    q = (base[rng.integers(0, 300, 400)] + rng.integers(-1, 2, size=(400, 64))).astype(np.float32)
    for k in (5, 40, 64):
        _check(ctx, parent, cent, q, k, "l2", ids)
        _check(ctx, parent, cent, q, k, "ip", ids)
    parent.close()
    same = np.tile(rng.standard_normal((1, 128)).astype(np.float32), (2000, 1))  # 2000 copies of one vector
    ids = rng.permutation(2000).astype(np.int64)
    parent = _parent(ctx, same, ids)
    q = rng.standard_normal((300, 128)).astype(np.float32)
    for k in (10, 64):
        _check(ctx, parent, same, q, k, "l2", ids)
        _check(ctx, parent, same, q, k, "ip", ids)
    parent.close()


def test_fused_short_last_slice_and_k_above_its_rows(ctx):
    """a last slice that holds fewer rows than k (its candidates are padded), a list barely longer than a slice"""
    rng = np.random.default_rng(112)
    for n in (1025, 1024 + 256 + 7, 2048 - 3):
        cent = rng.standard_normal((n, 48)).astype(np.float32)
        parent = _parent(ctx, cent)
        q = np.ascontiguousarray(cent[rng.integers(n - 40, n, 512)] + 0.01 * rng.standard_normal((512, 48)).astype(np.float32), np.float32)
        for k in (7, 64):
            _check(ctx, parent, cent, q, k, "l2")
        parent.close()


def test_fused_after_centroids_change(ctx):
    """the parent store changes between calls (rows appended and removed): the fused form reads the live arena"""
    from quake_amd.capi import Store
    rng = np.random.default_rng(113)
    n, d = 1200, 64
    cent = rng.standard_normal((n, d)).astype(np.float32)
    ids = np.arange(n, dtype=np.int64)
    parent = _parent(ctx, cent, ids)
    q = (cent[rng.integers(0, n, 640)] + 0.2 * rng.standard_normal((640, d))).astype(np.float32)
    _check(ctx, parent, cent, q, 16, "l2", ids)
    extra = rng.standard_normal((500, d)).astype(np.float32)
    eids = np.arange(10000, 10500, dtype=np.int64)
    parent.add_entries(0, eids, extra)
    cent2 = np.concatenate([cent, extra])
    ids2 = np.concatenate([ids, eids])
    _check(ctx, parent, cent2, q, 16, "l2", ids2)
    parent.close()


def test_shapes_outside_the_fused_form(ctx):
    """above 2048 rows (4096 for k > 32), k = 1, k > 64: other dense forms answer, same results"""
    rng = np.random.default_rng(114)
    cent = rng.standard_normal((5000, 32)).astype(np.float32)
    parent = _parent(ctx, cent)
    q = (cent[rng.integers(0, 5000, 500)] + 0.3 * rng.standard_normal((500, 32))).astype(np.float32)
    _check(ctx, parent, cent, q, 8, "l2", form="k_dense_pf")
    _check(ctx, parent, cent, q, 32, "l2", form=("k_dense_pf", "k_dense"))
    parent.close()
    cent = rng.standard_normal((2048, 32)).astype(np.float32)
    parent = _parent(ctx, cent)
    q = (cent[rng.integers(0, 2048, 500)] + 0.3 * rng.standard_normal((500, 32))).astype(np.float32)
    _check(ctx, parent, cent, q, 1, "l2", form="k_dense")
    _check(ctx, parent, cent, q, 100, "l2", form="k_dense")
    _check(ctx, parent, cent, q, 8, "l2", form="k_dense_fused")
    parent.close()


def _dense_case(rng):
    d = int(rng.choice([1, 3, 16, 17, 31, 32, 48, 64, 100, 128]))
    n = int(rng.choice([1024, 1025, 1100, 1279, 1280, 1536, 1793, 2047, 2048, 2049, 2600]))
    metric = str(rng.choice(["l2", "ip"]))
    integer = bool(rng.random() < 0.35)
    if integer:
        base = rng.integers(0, 4, size=(int(rng.choice([20, 200, n])), d)).astype(np.float32)
        cent = base[rng.integers(0, base.shape[0], n)]
    else:
        cent = rng.standard_normal((n, d)).astype(np.float32)
        if metric == "ip":
            cent /= np.maximum(np.linalg.norm(cent, axis=1, keepdims=True), 1e-6)
    ids = rng.permutation(n).astype(np.int64) * int(rng.choice([1, 7])) + int(rng.choice([0, 5, 1 << 34]))
    Q = int(rng.choice([64, 65, 100, 257, 512, 1030, 3000]))
    k = int(rng.choice([2, 3, 5, 10, 31, 32, 33, 50, 64]))
    q = (cent[rng.integers(0, n, size=Q)] + (rng.integers(-1, 2, size=(Q, d)) if integer else 0.1 * rng.standard_normal((Q, d)))).astype(np.float32)
    return cent, ids, q, k, metric


# (a one-off run with QK_RANDOM_DENSE=400 passes)
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("QK_RANDOM_DENSE", "24"))))
def test_random_dense_shapes_around_the_form_boundary(ctx, seed):
    """seeded shapes nobody picked by hand on both sides of the 2048-row boundary between the one-launch form and the prefiltered
    one: odd dimensions, ragged batches, integer data with dense ties (few distinct rows), ids unrelated to the row order"""
    rng = np.random.default_rng(5000 + seed)
    cent, ids, q, k, metric = _dense_case(rng)
    parent = _parent(ctx, cent, ids)
    try:
        gp, gd = ctx.coarse(parent, q, k, metric)
        form = ctx.last_scan_kernel()
        op, od = O.coarse(q, cent, ids, k, metric)
        tag = f"seed={seed} n={cent.shape[0]} d={cent.shape[1]} Q={q.shape[0]} k={k} {metric} form={form}"
        np.testing.assert_array_equal(gp, op, err_msg=tag)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32), err_msg=tag)
        if cent.shape[0] <= 2048 and q.shape[0] > 256:
            assert form == "k_dense_fused", tag
    finally:
        parent.close()
