"""CPU checks of the oracle's adaptive-partition-scanning restatement (oracle/quake_oracle.c, APS section).

The reference holds no golden vector for APS (its tests print the recall reached, test/cpp/search_recall_tests.cpp:284-340),
so the pieces are pinned against independent implementations: scipy's regularised incomplete beta, float64 numpy geometry,
and the fixed-nprobe search the walk must degenerate to when it never stops."""
import numpy as np
import pytest
from scipy.special import betainc

import oracle as O
from helpers import make_ivf, make_queries


def test_incomplete_beta_matches_scipy():
    # geometry.h:115-161 (Lentz, STOP = 1e-8): relative agreement ~1e-8 over the arguments APS uses
    for d in (16, 128, 768):
        a = (d + 1.0) / 2.0
        for x in (1e-3, 0.05, 0.3, 0.5, 0.9, 0.99, 0.999):
            got = O.incomplete_beta(a, 0.5, x)
            want = betainc(a, 0.5, x)
            assert got == pytest.approx(want, rel=2e-7, abs=1e-300)
    assert O.incomplete_beta(3.0, 0.5, -0.1) == float("inf")  # :116
    assert O.incomplete_beta(3.0, 0.5, 1.5) == float("inf")


def test_table_and_lookup():
    t = O.incomplete_beta_table(128)
    assert t.shape == (1001,) and t[0] == 0.0 and t[-1] == 1.0
    assert np.all(np.diff(t) >= 0)
    np.testing.assert_allclose(t[1:-1], betainc(64.5, 0.5, np.arange(1, 1000) / 1000.0), rtol=2e-7, atol=1e-300)
    # exact at the nodes, linear in between, clamped outside (geometry.h:182-211)
    assert O.incomplete_beta_lookup(t, 0.5) == pytest.approx(t[500], rel=1e-12)
    mid = O.incomplete_beta_lookup(t, 0.9005)
    assert mid == pytest.approx(0.5 * (t[900] + t[901]), rel=1e-9)
    assert O.incomplete_beta_lookup(t, 7.0) == pytest.approx(t[-1], rel=1e-12) and O.incomplete_beta_lookup(t, -1.0) == t[0]
    assert O.incomplete_beta_lookup(t, float("nan")) == pytest.approx(t[-1], rel=1e-12)  # std::min(1.0, NaN) -> 1.0


def test_boundary_distances_geometry():
    rng = np.random.default_rng(0)
    d = 24
    c = rng.standard_normal((9, d)).astype(np.float32)
    q = (c[0] + 0.1 * rng.standard_normal(d)).astype(np.float32)
    bd = O.boundary_distances(q, c, euclidean=True)
    assert bd[0] == -1.0
    q64, c64 = q.astype(np.float64), c.astype(np.float64)
    for j in range(1, 9):  # distance from q to the bisecting hyperplane of (c0, cj)
        v = c64[j] - c64[0]
        want = abs(np.dot(q64 - 0.5 * (c64[0] + c64[j]), v)) / np.linalg.norm(v)
        assert bd[j] == pytest.approx(want, rel=1e-4)
    bi = O.boundary_distances(q, c, euclidean=False)
    for j in range(1, 9):  # angle between q and the normalised midpoint
        m = 0.5 * (c64[0] + c64[j])
        ang = np.dot(q64, m / np.linalg.norm(m))
        if abs(ang) <= 1:
            assert bi[j] == pytest.approx(np.arccos(ang), rel=1e-4)


def test_recall_profile_properties():
    bd = np.array([-1.0, 0.2, 0.5, 0.9, 3.0], np.float32)
    p = O.recall_profile(bd, 1.0, 64)
    assert p[4] == 0.0  # boundary beyond the radius (geometry.h:364-367)
    assert p.sum() == pytest.approx(1.0, abs=1e-6)
    assert p[0] == pytest.approx(2 * p[1], rel=1e-6)  # :385
    assert p[1] > p[2] > p[3] > 0
    # unnormalised ratio of partition 1 against the closed form 0.5 * I_x((d+1)/2, 1/2), x = sin^2 of the cap angle
    r, b, d = 1.0, 0.2, 64
    h = r - b
    x = np.sqrt((2 * r * h - h * h) / (r * r))
    assert np.exp(O.log_cap_volume(r, b, d, use_precomputed=False)) == pytest.approx(0.5 * betainc((d + 1) / 2, 0.5, x), rel=1e-6)
    assert np.exp(O.log_cap_volume(r, b, d, use_precomputed=True)) == pytest.approx(0.5 * betainc((d + 1) / 2, 0.5, x), rel=2e-3)
    # nothing inside the radius: uniform (:396-399)
    np.testing.assert_allclose(O.recall_profile(np.array([-1, 5, 6, 7], np.float32), 1.0, 64), 0.25)
    with pytest.raises(RuntimeError):
        O.recall_profile(np.array([-1.0], np.float32), 1.0, 64)


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_search_aps_walk(metric):
    ivf = make_ivf(12000, 24, 40, seed=7, metric=metric)
    q = make_queries(60, 24, seed=8, like=ivf["x"], metric=metric)
    args = (q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 10, metric)
    prev_n = None
    for rt in (0.3, 0.7, 0.95):
        i, dd, n = O.search_aps(*args, rt, recompute_threshold=0.0, initial_search_fraction=0.5, num_threads=4)
        assert n.min() >= 2 and n.max() <= 20  # never stops before the second partition; M = 40 * 0.5
        if prev_n is not None:
            assert np.all(n >= prev_n)  # a higher target never scans less
        prev_n = n
        # the answer is the fixed-nprobe answer over the partitions the walk visited
        pids, _ = O.coarse(q, ivf["centroids"], None, 20, metric)
        for qi in range(q.shape[0]):
            pp = pids[qi:qi + 1, :n[qi]]
            wi, wd = O.batched_serial_scan(q[qi:qi + 1], ivf["vecs"], ivf["ids"], ivf["offsets"], pp, 10, metric)
            np.testing.assert_array_equal(i[qi], wi[0])
            np.testing.assert_array_equal(dd[qi].view(np.uint32), wd[0].view(np.uint32))
    # a target that is never reached: every candidate is scanned = search with nprobe = M
    i, dd, n = O.search_aps(*args, 2.0, initial_search_fraction=0.5)
    assert np.all(n == 20)
    wi, wd = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 20, 10, metric, batched_scan=True)
    np.testing.assert_array_equal(i, wi)
    # direct-form arithmetic (the reference's scan_list) walks the same partitions on this data
    i2, d2, n2 = O.search_aps(*args, 0.7, recompute_threshold=0.0, initial_search_fraction=0.5, expanded=False)
    i1, d1, n1 = O.search_aps(*args, 0.7, recompute_threshold=0.0, initial_search_fraction=0.5, expanded=True)
    assert (n1 == n2).mean() >= 0.95
    np.testing.assert_allclose(d1[n1 == n2], d2[n1 == n2], atol=1e-4)
    with pytest.raises(RuntimeError):  # fewer than 2 candidates (geometry.h:350)
        O.search_aps(*args, 0.9, initial_search_fraction=0.01)
