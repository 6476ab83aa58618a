"""Search-level oracle checks mirroring the reference's coordinator tests
(test/cpp/query_coordinator.cpp:201-254 worker==serial ids equal / dist 1e-4; :309-371,459-497 padding;
:257-306 empty partition; search_recall_tests.cpp:160-254 flat recall >= 0.99)."""
import numpy as np
import pytest

import oracle as O
from helpers import brute_force, make_ivf, make_queries


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_serial_vs_batched_paths(metric):
    ivf = make_ivf(6000, 32, 16, seed=2, metric=metric, empty=(5,))
    q = make_queries(40, 32, seed=3, like=ivf["x"], metric=metric)
    for nprobe, k in [(1, 1), (4, 10), (16, 100)]:
        si, sd = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, metric,
                          batched_scan=False)
        bi, bd = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, metric,
                          batched_scan=True)
        si2, sd2 = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], nprobe, k, metric,
                            batched_scan=False, fast=False)
        np.testing.assert_array_equal(si, si2)  # SIMD-across-rows form is bit-identical to the scalar chain
        np.testing.assert_array_equal(sd, sd2)
        np.testing.assert_allclose(sd, bd, atol=1e-4)  # query_coordinator.cpp:251
        # ids may differ only where two candidates are closer than fp32 noise
        diff = si != bi
        if diff.any():
            assert np.abs(sd - bd)[diff].max() < 1e-4 and diff.mean() < 0.01


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_full_probe_equals_brute_force(metric):
    ivf = make_ivf(4000, 24, 8, seed=4, metric=metric)
    q = make_queries(25, 24, seed=5, metric=metric)
    k = 10
    gi, gd, gaps = brute_force(ivf["vecs"], ivf["ids"], q, k, metric)
    for batched in (False, True):
        oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 8, k, metric, batched_scan=batched)
        np.testing.assert_allclose(od, gd, atol=1e-4)
        ok = gaps.min(axis=1) > 1e-4
        np.testing.assert_array_equal(oi[ok], gi[ok])
        assert O.recall(oi, gi).mean() >= 0.99  # search_recall_tests.cpp:160-189,225-254


def test_flat_index_and_padding():
    # flat index: centroids=None -> every partition scanned (query_coordinator.cpp:624-626)
    ivf = make_ivf(7, 8, 2, seed=6)
    q = make_queries(3, 8, seed=7)
    for batched in (False, True):
        oi, od = O.search(q, None, ivf["vecs"], ivf["ids"], ivf["offsets"], 1, 10, "l2", batched_scan=batched)
        assert (oi[:, 7:] == -1).all() and np.isinf(od[:, 7:]).all() and (od[:, 7:] > 0).all()
        assert (oi[:, :7] >= 0).all()
        oi, od = O.search(q, None, ivf["vecs"], ivf["ids"], ivf["offsets"], 1, 10, "ip", batched_scan=batched)
        assert (oi[:, 7:] == -1).all() and np.isinf(od[:, 7:]).all() and (od[:, 7:] < 0).all()


def test_minus_one_pids_and_zero_partitions():
    ivf = make_ivf(300, 8, 4, seed=8)
    q = make_queries(5, 8, seed=9)
    pids = np.array([[0, -1], [-1, -1], [3, 2], [1, 1], [-1, 0]], np.int64)
    si, sd = O.serial_scan(q, ivf["vecs"], ivf["ids"], ivf["offsets"], pids, 3, "l2")
    assert (si[1] == -1).all() and np.isinf(sd[1]).all()
    bi, bd = O.batched_serial_scan(q, ivf["vecs"], ivf["ids"], ivf["offsets"], pids, 3, "l2")
    assert (bi[1] == -1).all()
    np.testing.assert_array_equal(si[[0, 2, 4]], bi[[0, 2, 4]])
    # zero partitions to scan: [nq, 0] pids -> all padding (query_coordinator.cpp:459-497)
    zi, zd = O.serial_scan(q, ivf["vecs"], ivf["ids"], ivf["offsets"], np.zeros((5, 0), np.int64), 4, "l2")
    assert (zi == -1).all() and np.isinf(zd).all()


def test_integer_ties_resolved_by_id():
    # SIFT-like integer data: exact fp32, many exact ties; canonical order is (d2, id)
    ivf = make_ivf(3000, 16, 4, seed=10, integer=True)
    q = make_queries(20, 16, seed=11, like=ivf["x"], integer=True)
    gi, gd, gaps = brute_force(ivf["vecs"], ivf["ids"], q, 20, "l2")
    assert (gaps == 0).any()  # the case really has ties
    for batched in (False, True):
        oi, od = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 4, 20, "l2", batched_scan=batched)
        np.testing.assert_array_equal(oi, gi)
        np.testing.assert_array_equal(od, gd)


def test_multithreaded_equals_single():
    ivf = make_ivf(5000, 16, 8, seed=12)
    q = make_queries(33, 16, seed=13, like=ivf["x"])
    for batched in (False, True):
        a = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 3, 5, "l2", batched_scan=batched, num_threads=1)
        b = O.search(q, ivf["centroids"], ivf["vecs"], ivf["ids"], ivf["offsets"], 3, 5, "l2", batched_scan=batched, num_threads=4)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
