"""The two summation orders of the oracle's k-means update (CPU): the reference's row-after-row accumulate of
kmeans_refine_partitions (clustering.cpp:162-176) and the blocked canonical order of the Lloyd driver's mean update
(qo_kmeans_accumulate_blocked: 32-row blocks, 32-block groups, groups in order), each against an independent numpy
restatement built from float32 cumulative sums (np.cumsum over float32 IS the sequential chain)."""
import numpy as np
import pytest

import oracle as O


def _seq(v):
    """sequential fp32 sum of the rows of v, from +0 (np.cumsum accumulates in the array's dtype, one add per row)"""
    if v.shape[0] == 0:
        return np.zeros(v.shape[1], np.float32)
    z = np.zeros((1, v.shape[1]), np.float32)
    return np.cumsum(np.concatenate([z, v], 0), axis=0, dtype=np.float32)[-1]


def _blocked(v, l1=32, l2=32):
    groups = []
    for g0 in range(0, v.shape[0], l1 * l2):
        g = v[g0:g0 + l1 * l2]
        blocks = np.stack([_seq(g[b0:b0 + l1]) for b0 in range(0, g.shape[0], l1)])
        groups.append(_seq(blocks))
    return _seq(np.stack(groups)) if groups else np.zeros(v.shape[1], np.float32)


@pytest.mark.parametrize("n,m,d", [(5000, 7, 12), (40000, 3, 5), (1024, 1, 4), (1025, 1, 3), (33, 2, 9)])
def test_both_orders_against_numpy(n, m, d):
    rng = np.random.default_rng(n + m)
    x = (rng.standard_normal((n, d)) * 10.0 ** rng.integers(-3, 4, size=(n, 1))).astype(np.float32)
    a = rng.integers(0, m, size=n).astype(np.int64)
    a[rng.integers(0, n, size=5)] = -1
    a[rng.integers(0, n, size=5)] = m
    ss, sc = O.kmeans_accumulate(x, a, m)
    bs, bc = O.kmeans_accumulate(x, a, m, blocked=True)
    assert (sc == bc).all()
    for c in range(m):
        v = x[a == c]  # ascending row order
        assert sc[c] == v.shape[0]
        np.testing.assert_array_equal(ss[c].view(np.uint32), _seq(v).view(np.uint32))
        np.testing.assert_array_equal(bs[c].view(np.uint32), _blocked(v).view(np.uint32))
        np.testing.assert_allclose(bs[c], v.astype(np.float64).sum(0), rtol=1e-4, atol=1e-2 * np.abs(v).max(initial=1.0))


def test_blocked_order_is_the_drivers():
    """qo_kmeans = subsample / init / Lloyd iterations with the BLOCKED update: restated from the oracle's pieces"""
    rng = np.random.default_rng(3)
    cent = rng.standard_normal((6, 10)).astype(np.float32)
    x = (cent[rng.integers(0, 6, 1500)] + 0.3 * rng.standard_normal((1500, 10))).astype(np.float32)
    c, a, _ = O.kmeans(x, 6, "l2", niter=3, seed=5)
    perm = O.rand_perm(1500, 6, 5)  # (n <= 256 m: no subsample)
    cc = np.ascontiguousarray(x[perm[:6]])
    for _ in range(3):
        ta, _ = O.kmeans_assign(x, cc, "l2")
        s, cnt = O.kmeans_accumulate(x, ta, 6, blocked=True)
        cc, _ = O.kmeans_update(s, cnt, cc)
    np.testing.assert_array_equal(c.view(np.uint32), cc.view(np.uint32))
    np.testing.assert_array_equal(a, O.kmeans_assign(x, cc, "l2")[0])
