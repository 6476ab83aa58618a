"""Pins the CPU oracle against the reference's own known-answer vectors:
test/cpp/topk_buffer.cpp, test/cpp/list_scanning.cpp (transcribed as data in tests/golden/hand_cases.json)
and fixtures recorded from the reference's src/python/utils.py (tests/golden/utils_knn.npz)."""
import json
import os

import numpy as np
import pytest

import oracle as O


@pytest.fixture(scope="module")
def hand(golden_dir):
    with open(os.path.join(golden_dir, "hand_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def knnfix(golden_dir):
    return np.load(os.path.join(golden_dir, "utils_knn.npz"))


def test_topk_buffer_cases(hand):
    for c in hand["topk_buffer"]:
        b = O.TopkBuffer(c["k"], c["desc"])
        if c["op"] == "batch_add":
            b.batch_add(c["vals"], c["ids"])
        else:
            for v, i in zip(c["vals"], c["ids"]):
                b.add(v, i)
        if c["op"] == "add_then_reset":
            b.reset()
        v, i = b.get()
        assert len(v) == len(c["exp_vals"]), c["src"]
        np.testing.assert_array_equal(v, np.float32(c["exp_vals"]), err_msg=c["src"])
        np.testing.assert_array_equal(i, np.int64(c["exp_ids"]), err_msg=c["src"])


def test_topk_buffer_flush_when_full():
    # capacity-triggered flush (list_scanning.h:117-122): more adds than capacity, result must equal a global sort
    rng = np.random.default_rng(3)
    v = rng.standard_normal(1000).astype(np.float32)
    for desc in (False, True):
        b = O.TopkBuffer(7, desc, capacity=16)
        for i, x in enumerate(v):
            b.add(x, i)
        got_v, got_i = b.get()
        order = np.argsort(-v if desc else v, kind="stable")[:7]
        np.testing.assert_array_equal(got_i, order)
        np.testing.assert_array_equal(got_v, v[order])


def _check_ids(got, alternatives, src):
    assert len(got) == len(alternatives), src
    for g, alt in zip(got, alternatives):
        assert int(g) in alt, (src, got, alternatives)


@pytest.mark.parametrize("fast", [False, True])
def test_scan_list_cases(hand, fast):
    for c in hand["scan_list"]:
        b = O.TopkBuffer(c["k"], c["metric"] == "ip")
        O.scan_list(np.float32(c["query"]), np.float32(c["list"]), c["list_ids"], b, c["metric"],
                    squared_domain=fast, fast=fast)
        v, i = b.get()
        if fast and c["metric"] == "l2":
            v = np.sqrt(v)
        np.testing.assert_allclose(v, np.float32(c["exp_dist"]), rtol=1e-6, err_msg=c["src"])
        _check_ids(i, c["exp_ids"], c["src"])


def test_batched_scan_list_cases(hand):
    for c in hand["batched_scan_list"]:
        q = np.float32(c["queries"])
        bufs = [O.TopkBuffer(c["k"], c["metric"] == "ip", 10 * c["k"]) for _ in range(len(q))]
        lst = np.float32(c["list"]).reshape(-1, c["d"])
        O.batched_scan_list(q, lst, c["list_ids"], bufs, c["metric"])
        for j, b in enumerate(bufs):
            v, i = b.get()
            np.testing.assert_allclose(v, np.float32(c["exp_dist"][j]), rtol=1e-6, err_msg=c["src"])
            _check_ids(i, c["exp_ids"][j], c["src"])


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_against_reference_utils_knn(knnfix, metric):
    """oracle scan_list / batched_scan_list vs the recorded output of the reference's utils.knn
    (same assertion as list_scanning.cpp:432-562: ids equal, |ddist| <= 0.01; we also demand 1e-4)."""
    q, x, k = knnfix["queries"], knnfix["vectors"], int(knnfix["k"])
    exp_i, exp_d, gap = knnfix[f"knn_{metric}_ids"], knnfix[f"knn_{metric}_dist"], knnfix[f"knn_{metric}_mingap"]
    ids = np.arange(x.shape[0], dtype=np.int64)
    sep = gap > 1e-4  # queries whose top-(k+1) ranks are separated by more than fp32 noise
    assert sep.sum() >= len(sep) // 2
    # serial scan_list, literal mode
    for qi in range(q.shape[0]):
        b = O.TopkBuffer(k, metric == "ip")
        O.scan_list(q[qi], x, ids, b, metric)
        v, i = b.get()
        np.testing.assert_allclose(v, exp_d[qi], atol=1e-4)
        if sep[qi]:
            np.testing.assert_array_equal(i, exp_i[qi])
    # batched
    bufs = [O.TopkBuffer(k, metric == "ip", 10 * k) for _ in range(q.shape[0])]
    O.batched_scan_list(q, x, ids, bufs, metric)
    for qi, b in enumerate(bufs):
        v, i = b.get()
        np.testing.assert_allclose(v, exp_d[qi], atol=1e-4)
        if sep[qi]:
            np.testing.assert_array_equal(i, exp_i[qi])


def test_recall_matches_reference_utils(knnfix):
    got = O.recall(knnfix["recall_ids"], knnfix["knn_l2_ids"])
    np.testing.assert_allclose(got, knnfix["recall_expected"], atol=1e-7)
    # the C++ calculate_recall (list_scanning.h:14-37) counts duplicated returned ids once per occurrence
    cpp = O.recall(knnfix["recall_ids"], knnfix["knn_l2_ids"], set_semantics=False)
    assert (cpp >= got).all() and (cpp[1::3] > got[1::3]).any()


def test_chain_is_sequential_fma():
    """The canonical arithmetic: a k-ordered fmaf chain, checked against exact rational arithmetic."""
    from fractions import Fraction
    rng = np.random.default_rng(5)
    x = rng.standard_normal(37).astype(np.float32)
    y = rng.standard_normal(37).astype(np.float32)
    acc = np.float32(0)
    for a, b in zip(x, y):
        exact = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(acc))
        acc = np.float32(float(exact))  # one rounding of the exact a*b+acc  == fmaf  (double->float of an exactly
        # representable-in-double value would double-round; Fraction->float->float32 can too, so verify via nextafter)
        lo, hi = np.nextafter(acc, np.float32(-np.inf)), np.nextafter(acc, np.float32(np.inf))
        best = min((abs(Fraction(float(c)) - exact), float(c)) for c in (lo, acc, hi))[1]
        acc = np.float32(best)
    assert O.ip(x, y) == acc
