"""GPU parity of the recall-target search (adaptive partition scanning, qk_search_aps) against the oracle's restatement of
serial_scan's use_aps branch, through the C ABI.  Bar: int64 ids bit-exact, float32 distances bit-exact (both sides use
the canonical expanded-form arithmetic), and the SAME number of partitions visited per query -- the stopping rule is a
float comparison, evaluated on identical float32 inputs on both sides (the double-precision cap-volume ratios are rounded
to float32 before they are used, geometry.h:375)."""
import numpy as np
import pytest

import oracle as O
from helpers import make_ivf, make_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from quake_amd.capi import Context
    c = Context(0)
    yield c
    c.close()


def build_stores(ctx, ivf):
    from quake_amd.capi import Store
    s = Store(ctx, ivf["d"])
    s.build_csr(ivf["offsets"], ivf["ids"], ivf["vecs"])
    parent = Store(ctx, ivf["d"])
    nlist = ivf["nlist"]
    parent.build_csr(np.array([0, nlist], np.int64), np.arange(nlist, dtype=np.int64), ivf["centroids"])
    return parent, s


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("use_precomputed", [True, False])
def test_aps_matches_oracle(ctx, metric, use_precomputed):
    ivf = make_ivf(30000, 48, 60, seed=11, metric=metric, empty=(7,))
    q = make_queries(130, 48, seed=12, like=ivf["x"], metric=metric)
    parent, s = build_stores(ctx, ivf)
    for rt, thr, frac, k in [(0.5, 0.0, 0.5, 10), (0.9, 0.001, 0.5, 10), (0.99, 0.05, 0.25, 1), (0.8, 0.0, 1.0, 100)]:
        gi, gd, gn = ctx.search_aps(parent, s, q, k, metric, rt, recompute_threshold=thr, use_precomputed=use_precomputed,
                                    initial_search_fraction=frac)
        oi, od, on = O.search_aps(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], k, metric, rt,
                                  recompute_threshold=thr, use_precomputed=use_precomputed, initial_search_fraction=frac,
                                  expanded=True, num_threads=8)
        np.testing.assert_array_equal(gn, on)
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
        assert gn.min() >= 2


def test_aps_device_tensors_and_errors(ctx):
    import torch
    from quake_amd._lib import QuakeHipError
    ivf = make_ivf(20000, 32, 50, seed=13)
    q = make_queries(64, 32, seed=14, like=ivf["x"])
    parent, s = build_stores(ctx, ivf)
    hi, hd, hn = ctx.search_aps(parent, s, q, 10, "l2", 0.9, initial_search_fraction=0.4)
    di, dd, dn = ctx.search_aps(parent, s, torch.from_numpy(q).cuda(), 10, "l2", 0.9, initial_search_fraction=0.4)
    ctx.synchronize()
    np.testing.assert_array_equal(hi, di.cpu().numpy())
    np.testing.assert_array_equal(hd, dd.cpu().numpy())
    np.testing.assert_array_equal(hn, dn.cpu().numpy())
    # recall actually reached vs exact search: the estimate is conservative on this data
    gi, _ = ctx.search(parent, s, q, 50, 10, "l2")
    rec = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(hi, gi)])
    assert rec >= 0.9
    with pytest.raises(QuakeHipError):  # fewer than 2 candidate partitions (geometry.h:350)
        ctx.search_aps(parent, s, q, 10, "l2", 0.9, initial_search_fraction=0.02)
    with pytest.raises(QuakeHipError):
        ctx.search_aps(None, s, q, 10, "l2", 0.9)


def test_aps_many_candidates(ctx):
    """M = nlist * initial_search_fraction beyond QK_MAX_K: candidates come from the large-k selection."""
    ivf = make_ivf(40000, 16, 1200, seed=15)
    q = make_queries(48, 16, seed=16, like=ivf["x"])
    parent, s = build_stores(ctx, ivf)
    gi, gd, gn = ctx.search_aps(parent, s, q, 10, "l2", 0.9, initial_search_fraction=0.5)  # M = 600
    oi, od, on = O.search_aps(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 10, "l2", 0.9,
                              initial_search_fraction=0.5, expanded=True, num_threads=8)
    np.testing.assert_array_equal(gn, on)
    np.testing.assert_array_equal(gi, oi)
    np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_aps_first_round_bound_edges(ctx, metric):
    """The first round's bound is learnt from a sample of every query's NEAREST list (qk_scan_args::seed_first): lists long enough
    for the row-per-lane / mixed forms of the per-pair scan, some lists cut below k rows with queries placed right at their
    centroids (no bound may come from a later list for them), an empty list; twice, so that the form feedback takes another form."""
    k = 10
    ivf = make_ivf(100000, 64, 40, seed=31, metric=metric, empty=(5,))
    offs, keep = ivf["offsets"], np.ones(len(ivf["ids"]), bool)
    short = [3, 17, 22]
    for p, cut in zip(short, [0, 4, 9]):
        keep[int(offs[p]) + cut:int(offs[p + 1])] = False
    sizes = np.array([keep[int(offs[p]):int(offs[p + 1])].sum() for p in range(40)], np.int64)
    offs2 = np.zeros(41, np.int64)
    offs2[1:] = np.cumsum(sizes)
    ids2, vecs2 = np.ascontiguousarray(ivf["ids"][keep]), np.ascontiguousarray(ivf["vecs"][keep])
    q = make_queries(600, 64, seed=32, like=ivf["x"], metric=metric)
    for t, p in enumerate(short):
        q[t] = ivf["centroids"][p]
    parent, s = build_stores(ctx, dict(ivf, offsets=offs2, ids=ids2, vecs=vecs2))
    for rt, frac in [(0.9, 0.5), (0.99, 1.0)]:
        oi, od, on = O.search_aps(q, ivf["centroids"], vecs2, ids2, offs2, k, metric, rt, initial_search_fraction=frac,
                                  expanded=True, num_threads=8)
        for _ in range(4):
            gi, gd, gn = ctx.search_aps(parent, s, q, k, metric, rt, initial_search_fraction=frac)
            np.testing.assert_array_equal(gn, on)
            np.testing.assert_array_equal(gi, oi)
            np.testing.assert_array_equal(gd.view(np.uint32), od.view(np.uint32))
