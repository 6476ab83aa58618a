"""Host logic of the maintenance policy (quake_amd/maintenance.py) against the reference's own unit tests, restated:
test/cpp/hit_count_tracker.cpp, test/cpp/latency_estimator.cpp, test/cpp/maintenance_cost_estimator.cpp.  No GPU: the
latency model is injected."""
import math
import random

import numpy as np
import pytest

from quake_amd.maintenance import HitCountTracker, ListScanLatencyEstimator, MaintenanceCostEstimator


def test_hit_count_tracker_sliding_average():  # hit_count_tracker.cpp RandomQueriesTest / MultipleWindowCyclesTest
    for seed, nq, pmax in ((42, 20, 5), (123, 50, 4)):
        rng = random.Random(seed)
        window, total = 5, 1000
        tr = HitCountTracker(window, total)
        fr = []
        for q in range(nq):
            npart = rng.randint(1, pmax)
            ids = list(range(npart))
            sizes = [rng.randint(0, 300) for _ in range(npart)]
            tr.add_query_data(ids, sizes)
            fr.append(sum(sizes) / total)
            eff = min(q + 1, window)
            assert tr.get_current_scan_fraction() == pytest.approx(sum(fr[-eff:]) / eff, abs=1e-5)
        assert tr.get_num_queries_recorded() == window and tr.get_window_size() == window
        assert len(tr.get_per_query_hits()) == window


def test_hit_count_tracker_errors_and_reset():
    with pytest.raises(ValueError):
        HitCountTracker(0, 10)
    with pytest.raises(ValueError):
        HitCountTracker(5, 0)
    tr = HitCountTracker(3, 100)
    with pytest.raises(ValueError):
        tr.add_query_data([1, 2], [10])
    tr.add_query_data([1], [50])
    assert tr.get_current_scan_fraction() == pytest.approx(0.5)
    tr.reset()
    assert tr.get_num_queries_recorded() == 0 and tr.get_current_scan_fraction() == 1.0
    tr.set_total_vectors(200)
    tr.add_query_data([1], [50])
    assert tr.get_current_scan_fraction() == pytest.approx(0.25)
    with pytest.raises(ValueError):
        tr.set_total_vectors(0)


def test_hit_count_tracker_batch_path_equals_per_query_path():
    a, b = HitCountTracker(5, 1000), HitCountTracker(5, 1000)
    rng = np.random.default_rng(0)
    for _ in range(12):
        Q = int(rng.integers(1, 9))  # batches larger than the window included
        hp, sz = rng.integers(0, 10, (Q, 3)), rng.integers(0, 300, (Q, 3))
        for r, s_ in zip(hp, sz):
            a.add_query_data(r.tolist(), s_.tolist())
        b.add_batch(hp, sz)
        assert a.get_current_scan_fraction() == pytest.approx(b.get_current_scan_fraction(), abs=1e-5)
        # same window contents (the ring position may differ when one batch wraps the window more than once)
        assert sorted(list(map(int, x)) for x in a.get_per_query_hits()) == sorted(list(map(int, x)) for x in b.get_per_query_hits())
        ah = {}
        for qh in a.get_per_query_hits():
            for p_ in qh:
                ah[p_] = ah.get(p_, 0) + 1
        assert ah == b.aggregated_hits()


def plane(n, k):  # a latency surface that is exactly bilinear: interpolation and extrapolation must reproduce it
    return 100.0 + 3.0 * n + 7.0 * k + 0.5 * n * k


def test_latency_estimator_interpolation_extrapolation(tmp_path):  # latency_estimator.cpp BasicInterpolationExtrapolation
    fn = str(tmp_path / "test_latency_profile.csv")
    est = ListScanLatencyEstimator(8, [16, 32, 64], [1, 2, 4], 2, False, fn, profile_fn=plane)
    assert (tmp_path / "test_latency_profile.csv").exists()
    for n, k in [(16, 1), (24, 1), (32, 3), (64, 4), (48, 2), (128, 2), (40, 8), (256, 16)]:
        assert est.estimate_scan_latency(n, k) == pytest.approx(plane(n, k), rel=1e-6)
    assert est.estimate_scan_latency(0, 4) == 0.0 and est.estimate_scan_latency(16, 0) == 0.0
    with pytest.raises(IndexError):
        est.estimate_scan_latency(8, 1)
    # a second estimator with the same grid loads the file instead of profiling
    calls = []
    est2 = ListScanLatencyEstimator(8, [16, 32, 64], [1, 2, 4], 2, False, fn, profile_fn=lambda n, k: calls.append(1) or 0.0)
    assert not calls and est2.estimate_scan_latency(24, 1) == pytest.approx(plane(24, 1), rel=1e-6)
    # a different grid does not match the file -> re-profiles
    est3 = ListScanLatencyEstimator(8, [16, 64], [1, 4], 2, False, fn, profile_fn=lambda n, k: calls.append(1) or 5.0)
    assert len(calls) == 4 and est3.estimate_scan_latency(16, 1) == 5.0
    with pytest.raises(RuntimeError):
        ListScanLatencyEstimator(8, [32, 16], [1, 2], profile_fn=plane)


def make_estimator(alpha=0.9, k=10):
    lat = ListScanLatencyEstimator(128, [1, 2, 4, 16, 64, 256, 1024, 4096, 16384, 65536], [1, 4, 16, 64, 256], 1,
                                   profile_fn=lambda n, kk: 50.0 + 12.0 * n + 2.0 * kk)
    return MaintenanceCostEstimator(128, alpha, k, latency_estimator=lat), lat


def test_cost_estimator_split_and_delete_delta():  # maintenance_cost_estimator.cpp ComputeSplitDelta / ComputeDeleteDelta
    est, lat = make_estimator()
    L, k, alpha = lat.estimate_scan_latency, 10, 0.9
    size, hr, T = 1000, 0.3, 100
    want = (L(T + 1, k) - L(T, k)) + L(size // 2, k) * hr * (2 * alpha) - L(size, k) * hr
    assert est.compute_split_delta(size, hr, T) == pytest.approx(want, abs=1.0)
    avg_hr, avg_size = 0.25, size
    cost_old = (T - 1) * avg_hr * L(avg_size, k) + hr * L(size, k)
    merged_size = avg_size + size / (T - 1)
    merged_hr = avg_hr + hr / (T - 1)
    cost_new = (T - 1) * merged_hr * L(math.ceil(merged_size), k)
    want = (L(T - 1, k) - L(T, k)) + cost_new - cost_old
    assert est.compute_delete_delta(size, hr, T, avg_hr, avg_size) == pytest.approx(want, abs=1.0)
    # fewer vectors than partitions: only `size` partitions grow by one (maintenance_cost_estimator.cpp:439-442)
    small = 40
    cost_old = (T - 1) * avg_hr * L(avg_size, k) + hr * L(small, k)
    merged_hr = avg_hr + hr / (T - 1)
    cost_new = small * merged_hr * L(avg_size + 1, k) + (T - small - 1) * merged_hr * L(avg_size, k)
    want = (L(T - 1, k) - L(T, k)) + cost_new - cost_old
    assert est.compute_delete_delta(small, hr, T, avg_hr, avg_size) == pytest.approx(want, abs=1.0)
    assert est.compute_delete_delta(size, hr, 1, avg_hr, avg_size) == 0.0
    # reassignment-aware form (:456-490)
    rc, rs, rh = [600, 400], [900, 1100], [0.2, 0.1]
    want = (L(T - 1, k) - L(T, k)) + hr * L(size, k) + sum((h + hr) * L(s + size, k) - h * L(s, k) for s, h in zip(rs, rh))
    assert est.compute_delete_delta_w_reassign(size, hr, T, rc, rs, rh) == pytest.approx(want, abs=1.0)
    assert est.get_k() == 10 and est.get_latency_estimator() is lat


def test_cost_estimator_invalid_parameters():  # InvalidParametersThrow
    with pytest.raises(ValueError):
        MaintenanceCostEstimator(128, -0.5, 10, latency_estimator=object())
    with pytest.raises(ValueError):
        MaintenanceCostEstimator(128, 0.9, 0, latency_estimator=object())


def test_vectorised_cost_deltas_have_the_bits_of_the_scalar_calls():
    """the policy evaluates every partition at once (estimate_many / compute_deltas_many); each entry must be the scalar
    function's result bit for bit -- inside the grid, beyond it in n and in k, at the grid nodes, n = 0"""
    import math
    from quake_amd.maintenance import (DEFAULT_LATENCY_ESTIMATOR_RANGE_K as KV, DEFAULT_LATENCY_ESTIMATOR_RANGE_N as NV,
                                       ListScanLatencyEstimator, MaintenanceCostEstimator)
    rng = np.random.default_rng(0)
    for trial in range(40):
        a, b, c = rng.uniform(10, 500), rng.uniform(0.1, 5), rng.uniform(0, 3)
        lat = ListScanLatencyEstimator(16, NV, KV, 1, profile_fn=lambda n, k: a + b * n + c * k * math.sqrt(n) + rng.uniform(0, 20))
        ce = MaintenanceCostEstimator(16, float(rng.uniform(0.5, 1.0)), int(rng.choice([1, 10, 300])), latency_estimator=lat)
        n = np.concatenate([rng.integers(0, 200000, 300), np.array([0, 1, 2, 65535, 65536, 65537, 4096])])
        for k in (1, 3, 10, 256, 300):
            v = lat.estimate_many(n, k)
            s = np.array([lat.estimate_scan_latency(int(x), k) for x in n])
            assert (v.view(np.uint64) == s.view(np.uint64)).all()
        tp, sf, avg = int(rng.integers(2, 20000)), float(np.float32(rng.uniform(0, 0.2))), int(rng.integers(1, 5000))
        sizes = rng.integers(1, 30000, 300)
        hr = (rng.integers(0, 200, 300).astype(np.float32) / np.float32(200)).astype(np.float64)
        dd, sd = ce.compute_deltas_many(sizes, hr, tp, sf, avg)
        d1 = np.array([ce.compute_delete_delta(int(z), float(h), tp, sf, avg) for z, h in zip(sizes, hr)])
        s1 = np.array([ce.compute_split_delta(int(z), float(h), tp) for z, h in zip(sizes, hr)])
        assert (dd.view(np.uint64) == d1.view(np.uint64)).all() and (sd.view(np.uint64) == s1.view(np.uint64)).all()


def test_batched_rejection_deltas_are_the_scalar_function():
    """decide() asks compute_delete_delta_w_reassign for every delete candidate at once (_delete_deltas_w_reassign_many): the
    values must be the scalar function's bit for bit -- the two mirrors and the reference's rule (maintenance_cost_estimator.cpp:
    456-493) decide on them."""
    from quake_amd.maintenance import _delete_deltas_w_reassign_many
    ce, _ = make_estimator(0.9, 10)
    rng = np.random.default_rng(5)
    for trial in range(20):
        C = int(rng.integers(1, 40))
        sizes = rng.integers(1, 90000, C)
        hr = (rng.integers(0, 200, C).astype(np.float32) / np.float32(2048)).astype(np.float64)
        counts = rng.integers(0, 12, C)
        offsets = np.concatenate([[0], np.cumsum(counts)])
        t_sizes = rng.integers(0, 70000, int(offsets[-1]))
        t_hr = (rng.integers(0, 300, int(offsets[-1])).astype(np.float32) / np.float32(2048)).astype(np.float64)
        total = int(rng.integers(2, 30000))
        got = _delete_deltas_w_reassign_many(ce, sizes, hr, total, t_sizes, t_hr, offsets)
        for c in range(C):
            lo, hi = int(offsets[c]), int(offsets[c + 1])
            want = ce.compute_delete_delta_w_reassign(int(sizes[c]), float(hr[c]), total, [1] * (hi - lo), t_sizes[lo:hi], t_hr[lo:hi])
            assert np.float64(got[c]).tobytes() == np.float64(want).tobytes(), (trial, c, got[c], want)
    assert _delete_deltas_w_reassign_many(ce, [5], [0.1], 1, [], [], [0, 0])[0] == 0.0


class _StubIndex:
    """what MaintenancePolicy.decide() asks of an index: partition numbers and sizes, and where a delete candidate's rows would go"""

    def __init__(self, sizes, targets):
        self.sizes, self.targets, self.asked = dict(sizes), targets, []

    def _list_ids(self):
        return sorted(self.sizes)

    def nlist(self):
        return len(self.sizes)

    def ntotal(self):
        return int(sum(self.sizes.values()))

    def d(self):
        return 128

    def _partition_sizes(self, pids):
        return [self.sizes[int(p)] for p in pids]

    def _reassign_targets_many(self, pids):
        self.asked.extend(int(p) for p in pids)
        return {int(p): self.targets[int(p)] for p in pids}


def test_candidates_kept_without_the_nearest_two_search_get_the_same_verdict():
    """A delete candidate whose (overhead + hit_rate L(size)) is already above -threshold is kept whatever its rows' second-nearest
    centroids are (every reassignment term is >= 0 where the grid is nondecreasing): decide() skips the search for it.  The verdicts
    must be those of the full rule (maintenance_policies.cpp:68-131) -- compared here on random indexes, with and without the
    extension -- and the search must be asked for fewer candidates."""
    from quake_amd.index import MaintenancePolicyParams
    from quake_amd.maintenance import MaintenancePolicy
    rng = np.random.default_rng(11)
    saved = 0
    for trial in range(30):
        T = int(rng.integers(20, 400))
        sizes = {p: int(rng.choice([0, int(rng.integers(40, 400)), int(rng.integers(400, 30000))], p=[0.02, 0.3, 0.68])) for p in range(T)}
        targets = {}
        for p in range(T):
            others = [q for q in range(T) if q != p]
            t = sorted(int(v) for v in rng.choice(others, size=int(rng.integers(1, min(12, T - 1))), replace=False))
            targets[p] = (t, [int(rng.integers(1, 500)) for _ in t])
        window = 256
        mp = MaintenancePolicyParams()
        mp.window_size, mp.min_partition_size = window, 32
        mp.delete_threshold_ns = mp.split_threshold_ns = float(rng.choice([0.02, 0.5, 10.0]))
        mp.split_after_delete_rejection = bool(trial % 2)
        ce, lat = make_estimator(0.9, 10)
        hot = rng.choice(T, size=max(2, T // 6), replace=False)
        hits = np.where(rng.random((window, 4)) < 0.7, rng.choice(hot, (window, 4)), rng.integers(0, T, (window, 4)))
        out = []
        for shortcut in (True, False):
            idx = _StubIndex(sizes, targets)
            pol = MaintenancePolicy(idx, mp, cost_estimator=ce)
            pol.hit_count_tracker_.add_batch(hits, np.vectorize(sizes.get)(hits))
            if not shortcut:
                lat.monotone_from = lambda k: None
            try:
                out.append((pol.decide(), len(idx.asked)))
            finally:
                if not shortcut:
                    del lat.monotone_from
        assert out[0][0] == out[1][0], (trial, out[0][0], out[1][0])
        assert out[0][1] <= out[1][1]
        saved += out[1][1] - out[0][1]
    assert saved > 0
    # a grid that wobbles where partitions live: no shortcut (every candidate is examined)
    lat2 = ListScanLatencyEstimator(128, [1, 64, 1024, 65536], [1, 16], 1, profile_fn=lambda n, k: 100.0 - 0.5 * n if n <= 64 else 70.0 + 0.01 * n)
    assert lat2.monotone_from(10) == 64 and make_estimator()[1].monotone_from(10) == 1
    assert ListScanLatencyEstimator(128, [1, 64], [1, 16], 1, profile_fn=lambda n, k: 100.0 - n).monotone_from(10) is None
