"""Maintenance policy of the dynamic index: hit tracking, cost model, split / delete / local refinement.

Host-side bookkeeping over the device store -- the counterpart of
  HitCountTracker            src/cpp/src/hit_count_tracker.cpp:3-98
  ListScanLatencyEstimator   src/cpp/src/maintenance_cost_estimator.cpp:21-365 (grid + bilinear inter/extrapolation)
  MaintenanceCostEstimator   src/cpp/src/maintenance_cost_estimator.cpp:368-498
  MaintenancePolicy          src/cpp/src/maintenance_policies.cpp:18-202
Every data-parallel step it triggers runs on the GPU through the C ABI: the parent searches (qk_coarse), the 2-means
split (qk_kmeans), the reassignment of a deleted partition's vectors (qk_coarse + qk_store_add_batch) and the local
refinement (qk_store_refine_lists).

Two deliberate differences from the reference snapshot, both named in SURVEY.md section 8f-4:
  * QuakeIndex.search records the partitions each query scanned (the reference declares record_query_hits but never calls
    it, so its maintenance() can never act);
  * the latency model is profiled on the DEVICE scan (amortised per-query cost of scanning an n-row partition inside a
    batch), not on the CPU scan_list; any other model can be injected with `profile_fn`.
"""
import bisect
import math
import time

import numpy as np

DEFAULT_LATENCY_ESTIMATOR_RANGE_N = [1, 2, 4, 16, 64, 256, 1024, 4096, 16384, 65536]  # common.h:97
DEFAULT_LATENCY_ESTIMATOR_RANGE_K = [1, 4, 16, 64, 256]  # common.h:98
DEFAULT_LATENCY_ESTIMATOR_NTRIALS = 5  # common.h:99


class HitCountTracker:
    """Sliding window over the last `window_size` queries: which partitions each scanned and how many vectors that was."""

    def __init__(self, window_size, total_vectors):
        if window_size <= 0:
            raise ValueError("Window size must be positive")
        if total_vectors <= 0:
            raise ValueError("Total vectors must be positive")
        self.window_size_ = int(window_size)
        self.total_vectors_ = int(total_vectors)
        self.reset()

    def reset(self):
        self.curr_query_index_ = 0
        self.num_queries_recorded_ = 0
        self.running_sum_scan_fraction_ = np.float32(0.0)
        self.current_scan_fraction_ = np.float32(1.0)
        self.per_query_hits_ = [[] for _ in range(self.window_size_)]
        self.per_query_scanned_sizes_ = [[] for _ in range(self.window_size_)]

    def set_total_vectors(self, total_vectors):
        if total_vectors <= 0:
            raise ValueError("Total vectors must be positive")
        self.total_vectors_ = int(total_vectors)

    def _fraction(self, scanned_sizes):
        return np.float32(np.float32(int(sum(scanned_sizes))) / np.float32(self.total_vectors_))

    def add_query_data(self, hit_partition_ids, scanned_sizes):
        if len(hit_partition_ids) != len(scanned_sizes):
            raise ValueError("hit_partition_ids and scanned_sizes must be of equal length")
        frac = self._fraction(scanned_sizes)
        if self.num_queries_recorded_ < self.window_size_:
            self.per_query_hits_[self.num_queries_recorded_] = list(hit_partition_ids)
            self.per_query_scanned_sizes_[self.num_queries_recorded_] = list(scanned_sizes)
            self.running_sum_scan_fraction_ = np.float32(self.running_sum_scan_fraction_ + frac)
            self.num_queries_recorded_ += 1
        else:
            old = self._fraction(self.per_query_scanned_sizes_[self.curr_query_index_])
            self.running_sum_scan_fraction_ = np.float32(self.running_sum_scan_fraction_ - old)
            self.per_query_hits_[self.curr_query_index_] = list(hit_partition_ids)
            self.per_query_scanned_sizes_[self.curr_query_index_] = list(scanned_sizes)
            self.running_sum_scan_fraction_ = np.float32(self.running_sum_scan_fraction_ + frac)
            self.curr_query_index_ = (self.curr_query_index_ + 1) % self.window_size_
        eff = min(self.num_queries_recorded_, self.window_size_)
        self.current_scan_fraction_ = np.float32(self.running_sum_scan_fraction_ / np.float32(eff))

    def add_batch(self, hit_partition_ids, scanned_sizes):
        """A whole query batch at once ([Q, P] arrays, -1 = no partition): same window contents as Q add_query_data calls;
        the running sum is recomputed from the window (no per-query float32 drift)."""
        hp = np.asarray(hit_partition_ids).reshape(len(hit_partition_ids), -1)[-self.window_size_:]  # older rows cannot survive
        sz = np.where(hp >= 0, np.asarray(scanned_sizes).reshape(-1, hp.shape[1])[-self.window_size_:], 0)
        n = hp.shape[0]
        fill = min(n, self.window_size_ - self.num_queries_recorded_)  # rows that go into still-empty slots
        slots = np.empty(n, np.int64)
        slots[:fill] = np.arange(self.num_queries_recorded_, self.num_queries_recorded_ + fill)
        slots[fill:] = (self.curr_query_index_ + np.arange(n - fill)) % self.window_size_
        self.num_queries_recorded_ += fill
        self.curr_query_index_ = int((self.curr_query_index_ + (n - fill)) % self.window_size_)
        clean = bool((hp >= 0).all())
        for i, slot in enumerate(slots.tolist()):
            if clean:
                self.per_query_hits_[slot] = hp[i]
                self.per_query_scanned_sizes_[slot] = sz[i]
            else:
                keep = hp[i] >= 0
                self.per_query_hits_[slot] = hp[i][keep]
                self.per_query_scanned_sizes_[slot] = sz[i][keep]
        eff = min(self.num_queries_recorded_, self.window_size_)
        tot = float(sum(int(np.sum(v)) for v in self.per_query_scanned_sizes_[:eff])) if eff <= 64 else float(
            np.concatenate([np.asarray(v).reshape(-1) for v in self.per_query_scanned_sizes_[:eff]]).sum())
        self.running_sum_scan_fraction_ = np.float32(tot / self.total_vectors_)
        self.current_scan_fraction_ = np.float32(self.running_sum_scan_fraction_ / np.float32(max(eff, 1)))

    def aggregated_hits(self):
        """partition id -> number of window queries that scanned it (maintenance_policies.cpp:45-51)"""
        eff = min(self.num_queries_recorded_, self.window_size_)
        rows = [np.asarray(v, dtype=np.int64).reshape(-1) for v in self.per_query_hits_[:eff]]
        if not rows:
            return {}
        u, c = np.unique(np.concatenate(rows), return_counts=True)
        return dict(zip(u.tolist(), c.tolist()))

    def get_current_scan_fraction(self):
        return float(self.current_scan_fraction_)

    def get_per_query_hits(self):
        return self.per_query_hits_

    def get_per_query_scanned_sizes(self):
        return self.per_query_scanned_sizes_

    def get_window_size(self):
        return self.window_size_

    def get_num_queries_recorded(self):
        return self.num_queries_recorded_


def _linear_extrapolate(f1, f2, t):
    return f2 + t * (f2 - f1)  # maintenance_cost_estimator.h linear_extrapolate: slope of the last interval


class ListScanLatencyEstimator:
    """Latency grid L[n][k] in ns with bilinear interpolation inside the grid and linear extrapolation beyond it.
    profile_fn(n, k) -> ns fills the grid (default: profile the device scan, see device_profile_fn)."""

    def __init__(self, d, n_values, k_values, n_trials=DEFAULT_LATENCY_ESTIMATOR_NTRIALS, adaptive_nprobe=False,
                 profile_filename="", profile_fn=None):
        self.d_ = int(d)
        self.n_values_ = [int(v) for v in n_values]
        self.k_values_ = [int(v) for v in k_values]
        self.n_trials_ = int(n_trials)
        self.profile_filename_ = profile_filename
        if self.n_values_ != sorted(self.n_values_):
            raise RuntimeError("n_values must be sorted in ascending order.")
        if self.k_values_ != sorted(self.k_values_):
            raise RuntimeError("k_values must be sorted in ascending order.")
        self.scan_latency_model_ = [[0.0] * len(self.k_values_) for _ in self.n_values_]
        loaded = bool(profile_filename) and self.load_latency_profile(profile_filename)
        if not loaded:
            self.profile_scan_latency(profile_fn)
            if profile_filename:
                self.save_latency_profile(profile_filename)

    def profile_scan_latency(self, profile_fn=None):
        fn = profile_fn if profile_fn is not None else device_profile_fn(self.d_, self.n_trials_)
        for i, n in enumerate(self.n_values_):
            for j, k in enumerate(self.k_values_):
                self.scan_latency_model_[i][j] = float(fn(n, k))

    def set_scan_latency(self, n, k, latency_ns):
        self.scan_latency_model_[self.n_values_.index(n)][self.k_values_.index(k)] = float(latency_ns)

    @staticmethod
    def _axis(values, v):
        """(lower index, upper index, fraction, inside) along one axis; beyond the grid the fraction is measured from the
        LAST node in units of the last interval (maintenance_cost_estimator.cpp:143-190)."""
        if v <= values[-1]:
            it = bisect.bisect_right(values, v)
            if it == len(values):
                return len(values) - 2, len(values) - 1, 1.0, True
            lo, hi = it - 1, it
            return lo, hi, (v - values[lo]) / float(values[hi] - values[lo]), True
        lo, hi = len(values) - 2, len(values) - 1
        return lo, hi, (v - values[hi]) / float(values[hi] - values[lo]), False

    def estimate_scan_latency(self, n, k):
        n, k = int(n), int(k)
        if n == 0 or k == 0:
            return 0.0
        if n < self.n_values_[0] or k < self.k_values_[0]:
            raise IndexError("n or k is below the minimum supported values.")
        il, iu, t, n_in = self._axis(self.n_values_, n)
        jl, ju, u, k_in = self._axis(self.k_values_, k)
        m = self.scan_latency_model_
        f11, f12, f21, f22 = m[il][jl], m[il][ju], m[iu][jl], m[iu][ju]
        if n_in and k_in:
            return (1 - t) * (1 - u) * f11 + t * (1 - u) * f21 + (1 - t) * u * f12 + t * u * f22
        if not n_in and k_in:
            lo, up = _linear_extrapolate(f11, f21, t), _linear_extrapolate(f12, f22, t)
            return (1 - u) * lo + u * up
        if n_in and not k_in:
            lo, up = _linear_extrapolate(f11, f12, u), _linear_extrapolate(f21, f22, u)
            return (1 - t) * lo + t * up
        lo, up = _linear_extrapolate(f11, f21, t), _linear_extrapolate(f12, f22, t)
        return _linear_extrapolate(lo, up, u)

    def monotone_from(self, k):
        """the smallest grid value n0 from which the modelled latency is nondecreasing in n at this k -- along the grid in both k
        columns that bracket k, hence for every interpolated / extrapolated n >= n0 -- or None when it never is (a grid whose last
        segment falls).  (A profiled grid wobbles at n <= 4, where a scan is all launch latency.)"""
        if k < self.k_values_[0]:
            return None
        jl, ju, _, k_in = self._axis(self.k_values_, int(k))
        if not k_in:  # (beyond the grid in k the two columns enter with weights of both signs: nothing follows from their monotony)
            return None
        m = self.scan_latency_model_
        i0 = len(self.n_values_) - 1
        if len(self.n_values_) < 2 or m[i0][jl] < m[i0 - 1][jl] or m[i0][ju] < m[i0 - 1][ju]:
            return None
        while i0 > 0 and m[i0][jl] >= m[i0 - 1][jl] and m[i0][ju] >= m[i0 - 1][ju]:
            i0 -= 1
        return self.n_values_[i0]

    def estimate_many(self, n, k):
        """estimate_scan_latency for an int array n (same k): the same IEEE operations in the same order, element by element,
        so every entry has the bits of the scalar call (the policy's decisions must not depend on which one ran)."""
        n = np.asarray(n, dtype=np.int64)
        k = int(k)
        out = np.zeros(n.shape, np.float64)
        if k == 0 or n.size == 0:
            return out
        if k < self.k_values_[0] or (n[n != 0] < self.n_values_[0]).any():
            raise IndexError("n or k is below the minimum supported values.")
        nv = np.asarray(self.n_values_, np.int64)
        m = np.asarray(self.scan_latency_model_, np.float64)
        jl, ju, u, k_in = self._axis(self.k_values_, k)
        inside = n <= nv[-1]
        it = np.searchsorted(nv, n, side="right")
        il = np.where(inside, np.where(it == len(nv), len(nv) - 2, it - 1), len(nv) - 2)
        iu = np.where(inside, np.where(it == len(nv), len(nv) - 1, it), len(nv) - 1)
        il = np.clip(il, 0, len(nv) - 2)
        iu = np.clip(iu, 1, len(nv) - 1)
        span = (nv[iu] - nv[il]).astype(np.float64)
        t = np.where(inside, np.where(it == len(nv), 1.0, (n - nv[il]) / span), (n - nv[iu]) / span)
        f11, f12, f21, f22 = m[il, jl], m[il, ju], m[iu, jl], m[iu, ju]
        if k_in:
            a = (1 - t) * (1 - u) * f11 + t * (1 - u) * f21 + (1 - t) * u * f12 + t * u * f22
            lo, up = f21 + t * (f21 - f11), f22 + t * (f22 - f12)
            b = (1 - u) * lo + u * up
        else:
            lo, up = f12 + u * (f12 - f11), f22 + u * (f22 - f21)
            a = (1 - t) * lo + t * up
            lo2, up2 = f21 + t * (f21 - f11), f22 + t * (f22 - f12)
            b = up2 + u * (up2 - lo2)
        out = np.where(inside, a, b)
        out[n == 0] = 0.0
        return out

    # the reference's CSV layout (maintenance_cost_estimator.cpp:259-365): header, "n_size,k_size", n values, k values, rows
    def save_latency_profile(self, filename):
        try:
            with open(filename, "w") as f:
                f.write("n_size,k_size\n")
                f.write(f"{len(self.n_values_)},{len(self.k_values_)}\n")
                f.write(",".join(str(v) for v in self.n_values_) + "\n")
                f.write(",".join(str(v) for v in self.k_values_) + "\n")
                for row in self.scan_latency_model_:
                    f.write(",".join(repr(float(v)) for v in row) + "\n")
            return True
        except OSError:
            return False

    def load_latency_profile(self, filename):
        try:
            with open(filename) as f:
                lines = [ln.strip() for ln in f.read().splitlines()]
        except OSError:
            return False
        try:
            ns, ks = (int(v) for v in lines[1].split(","))
            nv = [int(v) for v in lines[2].split(",")]
            kv = [int(v) for v in lines[3].split(",")]
            if len(nv) != ns or len(kv) != ks or nv != self.n_values_ or kv != self.k_values_:
                return False
            model = []
            for i in range(ns):
                row = [float(v) for v in lines[4 + i].split(",")]
                if len(row) != ks:
                    return False
                model.append(row)
        except (IndexError, ValueError):
            return False
        self.scan_latency_model_ = model
        return True


def device_profile_fn(d, n_trials=DEFAULT_LATENCY_ESTIMATOR_NTRIALS, device=0, max_rows=1 << 22, elapsed=None):
    """profile_fn for a GPU index, THROUGHPUT regime: many partitions of n rows are scanned in one qk_scan call, one query
    each (up to 1024 pairs, at most max_rows rows in total), and the cost of one (query, partition) pair is the call's time
    divided by the number of pairs -- what one more probed partition of that size costs inside a busy serving batch.
    (A single-query latency, what the reference profiles on the CPU, is flat in n on a GPU: launch-bound.)
    elapsed: optional `elapsed(n, k, npart) -> seconds per call`, read in place of the clock (the scans still run): the grid becomes a
    function of its arguments, which is what a test of everything downstream of the grid needs."""
    import torch
    from . import capi

    state = {}

    def fn(n, k):
        dev = torch.device("cuda", device)
        if "ctx" not in state:
            state["ctx"] = capi.Context(device)
        ctx = state["ctx"]
        if state.get("n") != n:
            npart = int(max(16, min(1024, max_rows // max(n, 1))))
            s = capi.Store(ctx, d)
            s.build_csr(np.arange(npart + 1, dtype=np.int64) * n, torch.arange(npart * n, device=dev),
                        torch.rand(npart * n, d, device=dev))
            state.update(store=s, n=n, npart=npart, q=torch.rand(npart, d, device=dev),
                         pids=torch.arange(npart, device=dev, dtype=torch.int64)[:, None].contiguous())
        s, q, pids = state["store"], state["q"], state["pids"]
        kk = min(int(k), 448)
        ctx.scan(s, q, pids, kk, "l2")
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_trials):
            ctx.scan(s, q, pids, kk, "l2")
        ctx.synchronize()
        per_call = (time.perf_counter() - t0) / n_trials if elapsed is None else float(elapsed(n, kk, state["npart"]))
        return per_call / state["npart"] * 1e9

    return fn


class MaintenanceCostEstimator:
    def __init__(self, d, alpha, k, latency_estimator=None, profile_fn=None):
        if k <= 0:
            raise ValueError("k must be positive")
        if alpha <= 0.0:
            raise ValueError("alpha must be positive")
        self.d_, self.alpha_, self.k_ = int(d), float(alpha), int(k)
        self.latency_estimator_ = latency_estimator or ListScanLatencyEstimator(
            d, DEFAULT_LATENCY_ESTIMATOR_RANGE_N, DEFAULT_LATENCY_ESTIMATOR_RANGE_K, DEFAULT_LATENCY_ESTIMATOR_NTRIALS,
            profile_fn=profile_fn)

    def get_latency_estimator(self):
        return self.latency_estimator_

    def get_k(self):
        return self.k_

    def compute_split_delta(self, partition_size, hit_rate, total_partitions):  # :384-394
        L = self.latency_estimator_.estimate_scan_latency
        delta_overhead = L(total_partitions + 1, self.k_) - L(total_partitions, self.k_)
        old_cost = L(partition_size, self.k_) * hit_rate
        new_cost = L(partition_size // 2, self.k_) * hit_rate * (2.0 * self.alpha_)
        return delta_overhead + new_cost - old_cost

    def compute_delete_delta(self, partition_size, hit_rate, total_partitions, avg_partition_hit_rate, avg_partition_size):
        if total_partitions <= 1:
            return 0.0
        L = self.latency_estimator_.estimate_scan_latency
        k = self.k_
        delta_overhead = L(total_partitions - 1, k) - L(total_partitions, k)
        cost_old = (total_partitions - 1) * avg_partition_hit_rate * L(int(avg_partition_size), k) + hit_rate * L(partition_size, k)
        merged_size = avg_partition_size + float(partition_size) / (total_partitions - 1)
        merged_hit_rate = avg_partition_hit_rate + hit_rate / float(total_partitions - 1)
        if partition_size < total_partitions:
            cost_new = (partition_size * merged_hit_rate * L(int(avg_partition_size + 1), k)
                        + (total_partitions - partition_size - 1) * merged_hit_rate * L(int(avg_partition_size), k))
        else:
            cost_new = (total_partitions - 1) * merged_hit_rate * L(int(math.ceil(merged_size)), k)
        return delta_overhead + (cost_new - cost_old)

    def compute_deltas_many(self, sizes, hit_rates, total_partitions, avg_partition_hit_rate, avg_partition_size):
        """(delete deltas, split deltas) of every partition at once: compute_delete_delta / compute_split_delta element by
        element, same operations in the same order (10000 partitions: 240 ms of scalar calls -> a few ms)."""
        sizes = np.asarray(sizes, np.int64)
        hr = np.asarray(hit_rates, np.float64)
        Ls, Lm = self.latency_estimator_.estimate_scan_latency, self.latency_estimator_.estimate_many
        k, tp = self.k_, int(total_partitions)
        L_size = Lm(sizes, k)
        split = (Ls(tp + 1, k) - Ls(tp, k)) + Lm(sizes // 2, k) * hr * (2.0 * self.alpha_) - L_size * hr
        if tp <= 1:
            return np.zeros(sizes.shape, np.float64), split
        delta_overhead = Ls(tp - 1, k) - Ls(tp, k)
        L_avg, L_avg1 = Ls(int(avg_partition_size), k), Ls(int(avg_partition_size + 1), k)
        cost_old = (tp - 1) * avg_partition_hit_rate * L_avg + hr * L_size
        merged_size = avg_partition_size + sizes.astype(np.float64) / (tp - 1)
        merged_hit_rate = avg_partition_hit_rate + hr / float(tp - 1)
        small = sizes < tp
        cost_small = sizes * merged_hit_rate * L_avg1 + (tp - sizes - 1) * merged_hit_rate * L_avg
        cost_big = (tp - 1) * merged_hit_rate * Lm(np.ceil(merged_size).astype(np.int64), k)
        cost_new = np.where(small, cost_small, cost_big)
        return delta_overhead + (cost_new - cost_old), split

    def compute_delete_delta_w_reassign(self, partition_size, hit_rate, total_partitions, reassign_counts, reassign_sizes,
                                        reassign_hit_rates):
        if total_partitions <= 1:
            return 0.0
        assert len(reassign_sizes) == len(reassign_counts) == len(reassign_hit_rates)
        Ls, Lm = self.latency_estimator_.estimate_scan_latency, self.latency_estimator_.estimate_many
        k = self.k_
        delta_overhead = Ls(total_partitions - 1, k) - Ls(total_partitions, k)
        removal_delta = hit_rate * Ls(partition_size, k)
        reassign_delta = 0.0
        if len(reassign_sizes):
            # the latencies of all targets at once (estimate_many == the scalar call, bit for bit); the sum in the targets' order
            sz = np.asarray(reassign_sizes, np.int64)
            hr = np.asarray(reassign_hit_rates, np.float64)
            terms = (hr + hit_rate) * Lm(sz + int(partition_size), k) - hr * Lm(sz, k)
            for t in terms.tolist():
                reassign_delta += t
        return delta_overhead + removal_delta + reassign_delta


def _delete_deltas_w_reassign_many(ce, cand_sizes, cand_hr, total_partitions, t_sizes, t_hr, offsets):
    """compute_delete_delta_w_reassign for C candidates at once (candidate c's targets = t_*[offsets[c]:offsets[c+1]]): the latency
    terms of ALL targets in two estimate_many calls (== the scalar call, bit for bit), every candidate's sum in its targets' order --
    the values of the scalar function.  (20000 partitions: the scalar function per candidate was 90 us of numpy call overhead each.)"""
    C_ = len(cand_sizes)
    out = np.zeros(C_, np.float64)
    if total_partitions <= 1 or C_ == 0:
        return out
    Ls, Lm = ce.latency_estimator_.estimate_scan_latency, ce.latency_estimator_.estimate_many
    k = ce.k_
    delta_overhead = Ls(total_partitions - 1, k) - Ls(total_partitions, k)
    cand_sizes = np.asarray(cand_sizes, np.int64)
    cand_hr = np.asarray(cand_hr, np.float64)
    removal = cand_hr * Lm(cand_sizes, k)
    counts = np.diff(np.asarray(offsets, np.int64))
    t_sizes = np.asarray(t_sizes, np.int64)
    t_hr = np.asarray(t_hr, np.float64)
    rep_size, rep_hr = np.repeat(cand_sizes, counts), np.repeat(cand_hr, counts)
    terms = ((t_hr + rep_hr) * Lm(t_sizes + rep_size, k) - t_hr * Lm(t_sizes, k)).tolist() if t_sizes.size else []
    offs = [int(v) for v in offsets]
    for c in range(C_):
        acc = 0.0
        for t in terms[offs[c]:offs[c + 1]]:
            acc += t
        out[c] = delta_overhead + float(removal[c]) + acc
    return out


class MaintenancePolicy:
    """perform_maintenance(): delete the partitions whose removal lowers the modelled query cost, split the ones whose
    split does, then refine around the new partitions (maintenance_policies.cpp:33-177)."""

    def __init__(self, index, params, cost_estimator=None):
        self.index_ = index
        self.params_ = params
        self.cost_estimator_ = cost_estimator or MaintenanceCostEstimator(index.d(), params.alpha, 10)
        self.hit_count_tracker_ = HitCountTracker(params.window_size, max(index.ntotal(), 1))

    def record_query_hits(self, partition_ids, scanned_sizes=None):
        ids = [int(p) for p in partition_ids]
        if scanned_sizes is None:
            scanned_sizes = self.index_._partition_sizes(ids)
        self.hit_count_tracker_.add_query_data(ids, [int(s) for s in scanned_sizes])

    def reset(self):
        self.hit_count_tracker_.reset()

    def decide(self):
        """STEPS 1-2 of perform_maintenance (maintenance_policies.cpp:43-137) without acting: (partitions to delete, partitions to
        split) under the recorded window and the cost model -- ([], []) while the window is not full.  A second call right after
        perform_maintenance() says what the policy still wants; an index at the policy's fixed point gets two empty lists."""
        idx, p, tr = self.index_, self.params_, self.hit_count_tracker_
        if tr.get_num_queries_recorded() < p.window_size:  # :36-41
            return [], []
        hits = tr.aggregated_hits()
        all_pids = [int(v) for v in idx._list_ids()]
        total_partitions = idx.nlist()
        scan_fraction = tr.get_current_scan_fraction()
        avg_size = idx.ntotal() // max(total_partitions, 1)
        ce = self.cost_estimator_
        # the walk over the partitions (maintenance_policies.cpp:63-131) as array operations in partition order -- the same decisions:
        # every delta is the scalar function's value bit for bit (compute_deltas_many / estimate_many), only the Python loop is gone
        # (a 50M index has 20000 partitions; the loop, a size lookup per partition and per reassignment target through ctypes, and the
        # scalar cost calls were most of a maintenance call that decided to do nothing)
        pid_v = np.asarray(all_pids, np.int64)
        size_v = np.asarray(idx._partition_sizes(all_pids), np.int64)
        sizes = dict(zip(all_pids, size_v.tolist()))
        hit_v = np.zeros(pid_v.shape[0], np.float32)
        if hits:
            pos = {pid: i for i, pid in enumerate(all_pids)}
            for pid, h in hits.items():
                i = pos.get(int(pid))
                if i is not None:
                    hit_v[i] = h
        else:
            pos = None
        hr_v = (hit_v / np.float32(p.window_size)).astype(np.float64)
        dd_v, sd_v = ce.compute_deltas_many(size_v, hr_v, total_partitions, scan_fraction, avg_size)
        in_delete = dd_v < -p.delete_threshold_ns
        big = size_v > p.min_partition_size
        split_ok = sd_v < -p.split_threshold_ns
        examined = in_delete & big if p.enable_delete_rejection else np.zeros_like(in_delete)
        delete_m = in_delete & ~examined           # the delete branch without the rejection rule (:128-130)
        split_m = ~in_delete & big & split_ok      # the else branch (:131-139)
        # the delete candidates that the rejection rule examines: where their vectors would go is asked for all of them at once
        # Which of them need the nearest-two search of their rows at all?  The rule deletes when
        #     delta = (overhead + hit_rate L(size)) + sum over targets of ((hr_t + hit_rate) L(size_t + size) - hr_t L(size_t)) < -threshold
        # and every term of the sum is >= 0 where L is nondecreasing -- so a candidate whose FIRST bracket is already >= -threshold is
        # kept whatever its targets are (rounding is monotone: adding a non-negative sum cannot take the bracket below itself).  Only
        # the others are examined: the unhit ones, in practice -- a third of the candidates of a 50M index.  Exactness: the shortcut is
        # taken only when every partition's size lies where the grid is nondecreasing (monotone_from), and only beyond a guard band
        # of 1e-9 relative that covers the last-bit wobble of the interpolation; everything else goes through the full rule.
        kept_early = np.zeros_like(examined)
        n0 = ce.get_latency_estimator().monotone_from(ce.get_k()) if hasattr(ce.get_latency_estimator(), "monotone_from") else None
        if n0 is not None and examined.any() and total_partitions > 1 and not ((size_v > 0) & (size_v < n0)).any():
            Ls, Lm = ce.get_latency_estimator().estimate_scan_latency, ce.get_latency_estimator().estimate_many
            k_ = ce.get_k()
            bracket = (Ls(total_partitions - 1, k_) - Ls(total_partitions, k_)) + hr_v * Lm(size_v, k_)
            guard = 1e-9 * (np.abs(bracket) + abs(p.delete_threshold_ns) + 1.0)
            kept_early = examined & (bracket >= -p.delete_threshold_ns + guard)
        cand_ix = np.nonzero(examined & ~kept_early)[0]
        cand = [all_pids[i] for i in cand_ix.tolist()]
        if cand and hasattr(idx, "_reassign_targets_many"):
            targets = idx._reassign_targets_many(cand)
        else:
            targets = {}
        if cand and pos is None:
            pos = {pid: i for i, pid in enumerate(all_pids)}
        split_rejected = bool(getattr(p, "split_after_delete_rejection", False))
        if cand:
            # where would a candidate's vectors go?  second-nearest centroid of every vector (:79-101)
            t_ix, offsets = [], [0]
            for pid in cand:
                uniq, counts = targets[pid] if pid in targets else idx._reassign_targets(pid)
                t_ix.extend(pos[int(u)] for u in uniq)
                offsets.append(len(t_ix))
            t_ix = np.asarray(t_ix, np.int64)
            deltas = _delete_deltas_w_reassign_many(ce, size_v[cand_ix], hr_v[cand_ix], total_partitions, size_v[t_ix], hr_v[t_ix], offsets)
            for i, delta in zip(cand_ix.tolist(), deltas.tolist()):
                if delta < -p.delete_threshold_ns:
                    delete_m[i] = True
                elif split_rejected and split_ok[i]:
                    split_m[i] = True  # (extension, MaintenancePolicyParams: kept by the rejection -> split test)
        if split_rejected:
            split_m |= kept_early & split_ok  # (kept by the rejection rule without the search: the same split test)
        to_delete = [all_pids[i] for i in np.nonzero(delete_m)[0].tolist()]
        to_split = [all_pids[i] for i in np.nonzero(split_m)[0].tolist()]
        if len(to_delete) >= total_partitions:
            # (safety, not in the reference: a model that wants every partition gone would leave the vectors nowhere to
            #  go -- the largest partition survives)
            to_delete.remove(max(to_delete, key=lambda q_: sizes[q_]))
        return to_delete, to_split

    def perform_maintenance(self):
        import torch
        from .index import MaintenanceTimingInfo
        info = MaintenanceTimingInfo()
        idx, p, tr = self.index_, self.params_, self.hit_count_tracker_
        if tr.get_num_queries_recorded() < p.window_size:  # :36-41
            return info
        t_total = time.perf_counter()
        to_delete, to_split = self.decide()
        t0 = time.perf_counter()
        if to_delete:
            idx._delete_partitions(to_delete, reassign=True)
        info.delete_time_us = int((time.perf_counter() - t0) * 1e6)
        t0 = time.perf_counter()
        new_pids = []
        if to_split:
            new_pids = idx._split_partitions_in_place(to_split) if hasattr(idx, "_split_partitions_in_place") else None
            if new_pids is None:
                split = idx._split_partitions(to_split)
                idx._delete_partitions(to_split, reassign=False)
                new_pids = idx._add_partitions(split)
        info.split_time_us = int((time.perf_counter() - t0) * 1e6)
        t0 = time.perf_counter()
        if new_pids:
            self.local_refinement(new_pids)
        info.split_refine_time_us = int((time.perf_counter() - t0) * 1e6)
        info.n_splits = len(to_split)
        info.n_deletes = len(to_delete)
        info.total_time_us = int((time.perf_counter() - t_total) * 1e6)
        tr.set_total_vectors(max(idx.ntotal(), 1))
        return info

    def local_refinement(self, partition_ids):  # :187-202
        import torch
        idx, p = self.index_, self.params_
        if p.refinement_radius == 0:
            return
        refine = torch.tensor(idx._neighbour_partitions(list(partition_ids), int(p.refinement_radius)), dtype=torch.int64)
        idx.refine_partitions(refine, int(p.refinement_iterations))
