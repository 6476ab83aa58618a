"""ctypes + numpy front end of the C oracle (oracle/quake_oracle.c).

TEST INFRASTRUCTURE ONLY -- never imported by quake_amd/.  Every function here is a thin
marshalling layer; the arithmetic lives in the C file, which cites the reference file:line
each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libquake_oracle.so")

METRIC_IP = 0  # faiss::METRIC_INNER_PRODUCT
METRIC_L2 = 1  # faiss::METRIC_L2

__all__ = [
    "METRIC_IP", "METRIC_L2", "build", "lib", "metric_code", "ip", "l2sqr_direct", "row_norms", "TopkBuffer",
    "scan_list", "batched_scan_list", "serial_scan", "batched_serial_scan", "coarse", "search", "rand_perm",
    "kmeans_assign", "kmeans_accumulate", "kmeans", "kmeans_update", "normalize_rows", "kmeans_refine_partitions", "recall", "csr_from_partitions",
    "max_threads", "effective_cores", "incomplete_beta", "incomplete_beta_table", "incomplete_beta_lookup", "log_cap_volume", "recall_profile",
    "boundary_distances", "search_aps",
]


def build(force=False):
    """Compile libquake_oracle.so with the committed Makefile (gcc, a few seconds)."""
    src = os.path.join(_HERE, "quake_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libquake_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.qo_ip.restype = C.c_float
        L.qo_ip.argtypes = [_f32p, _f32p, C.c_int]
        L.qo_l2sqr_direct.restype = C.c_float
        L.qo_l2sqr_direct.argtypes = [_f32p, _f32p, C.c_int]
        L.qo_l2sqr_expanded.restype = C.c_float
        L.qo_l2sqr_expanded.argtypes = [C.c_float, C.c_float, C.c_float]
        L.qo_row_norms.argtypes = [_f32p, C.c_int64, C.c_int, _f32p]
        L.qo_topk_create.restype = C.c_void_p
        L.qo_topk_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.qo_topk_destroy.argtypes = [C.c_void_p]
        L.qo_topk_add.argtypes = [C.c_void_p, C.c_float, C.c_int64]
        L.qo_topk_batch_add.argtypes = [C.c_void_p, _f32p, _i64p, C.c_int]
        L.qo_topk_reset.argtypes = [C.c_void_p]
        L.qo_topk_get.restype = C.c_int
        L.qo_topk_get.argtypes = [C.c_void_p, _f32p, _i64p]
        L.qo_scan_list.argtypes = [_f32p, _f32p, _i64p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.qo_scan_list_fast.argtypes = [_f32p, _f32p, _i64p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.qo_batched_scan_list.argtypes = [_f32p, _f32p, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                           C.c_int, C.c_int]
        L.qo_serial_scan.argtypes = [_f32p, C.c_int64, _f32p, _i64p, _i64p, C.c_int64, C.c_int, _i64p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, _i64p, _f32p]
        L.qo_batched_serial_scan.argtypes = [_f32p, C.c_int64, _f32p, _i64p, _i64p, C.c_int64, C.c_int, _i64p,
                                             C.c_int, C.c_int, C.c_int, C.c_int, _i64p, _f32p]
        L.qo_coarse.restype = C.c_int
        L.qo_coarse.argtypes = [_f32p, C.c_int64, _f32p, _i64p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _i64p,
                                _f32p]
        L.qo_search.argtypes = [_f32p, C.c_int64, _f32p, _i64p, _f32p, _i64p, _i64p, C.c_int64, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i64p, _f32p]
        L.qo_rand_perm.argtypes = [C.c_int64, C.c_int64, C.c_uint64, _i64p]
        L.qo_kmeans_assign.argtypes = [_f32p, C.c_int64, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, _i64p, _f32p]
        L.qo_kmeans_accumulate.argtypes = [_f32p, C.c_int64, C.c_int, _i64p, C.c_int64, _f32p, _i64p]
        L.qo_kmeans_accumulate_blocked.argtypes = [_f32p, C.c_int64, C.c_int, _i64p, C.c_int64, _f32p, _i64p]
        L.qo_kmeans_finalize.argtypes = [_f32p, _i64p, C.c_int64, C.c_int, C.c_int, _f32p]
        L.qo_normalize_rows.argtypes = [_f32p, C.c_int64, C.c_int]
        L.qo_kmeans_update.restype = C.c_int
        L.qo_kmeans_update.argtypes = [_f32p, _i64p, C.c_int64, C.c_int, _f32p]
        L.qo_kmeans.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_int, _f32p,
                                _i64p]
        L.qo_kmeans_refine_partitions.argtypes = [_f32p, C.c_int64, C.c_int, _f32p, _i64p, _i64p, C.c_int, C.c_int,
                                                  C.c_int, _f32p, _i64p, _i64p]
        L.qo_recall.argtypes = [_i64p, _i64p, C.c_int64, C.c_int, _f32p]
        L.qo_recall_set.argtypes = [_i64p, _i64p, C.c_int64, C.c_int, _f32p]
        L.qo_max_threads.restype = C.c_int
        _f64p = C.POINTER(C.c_double)
        L.qo_incomplete_beta.restype = C.c_double
        L.qo_incomplete_beta.argtypes = [C.c_double, C.c_double, C.c_double]
        L.qo_incomplete_beta_table.argtypes = [C.c_int, _f64p]
        L.qo_incomplete_beta_lookup.restype = C.c_double
        L.qo_incomplete_beta_lookup.argtypes = [_f64p, C.c_double]
        L.qo_log_cap_volume.restype = C.c_double
        L.qo_log_cap_volume.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _f64p]
        L.qo_recall_profile.restype = C.c_int
        L.qo_recall_profile.argtypes = [_f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _f64p, _f32p]
        L.qo_boundary_distances.argtypes = [_f32p, C.POINTER(_f32p), C.c_int, C.c_int, C.c_int, _f32p]
        L.qo_search_aps.restype = C.c_int
        L.qo_search_aps.argtypes = [_f32p, C.c_int64, _f32p, _i64p, C.c_int64, _f32p, _i64p, _i64p, C.c_int64, C.c_int,
                                    C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, _i64p,
                                    _f32p, C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def metric_code(metric):
    if isinstance(metric, str):
        m = metric.lower()
        if m == "l2":
            return METRIC_L2
        if m == "ip":
            return METRIC_IP
        raise ValueError("Invalid metric type: " + metric)  # common.h:145-156
    return int(metric)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _pf(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _pi(a):
    return a.ctypes.data_as(_i64p) if a is not None else None


def max_threads():
    return lib().qo_max_threads()


def ip(x, y):
    x, y = _f32(x), _f32(y)
    return lib().qo_ip(_pf(x), _pf(y), x.shape[0])


def l2sqr_direct(x, y):
    x, y = _f32(x), _f32(y)
    return lib().qo_l2sqr_direct(_pf(x), _pf(y), x.shape[0])


def row_norms(x):
    x = _f32(x)
    out = np.empty(x.shape[0], np.float32)
    lib().qo_row_norms(_pf(x), x.shape[0], x.shape[1], _pf(out))
    return out


class TopkBuffer:
    """TypedTopKBuffer<float,int64_t> (list_scanning.h:41-204)."""

    def __init__(self, k, is_descending, capacity=8192):
        self.k = k
        self.h = lib().qo_topk_create(k, int(bool(is_descending)), capacity)

    def __del__(self):
        if getattr(self, "h", None):
            lib().qo_topk_destroy(self.h)
            self.h = None

    def add(self, v, i):
        lib().qo_topk_add(self.h, float(v), int(i))

    def batch_add(self, v, i):
        v, i = _f32(v), _i64(i)
        lib().qo_topk_batch_add(self.h, _pf(v), _pi(i), v.shape[0])

    def reset(self):
        lib().qo_topk_reset(self.h)

    def get(self):
        """(get_topk(), get_topk_indices()): min(curr,k) sorted entries."""
        v = np.empty(max(self.k, 1), np.float32)
        i = np.empty(max(self.k, 1), np.int64)
        n = lib().qo_topk_get(self.h, _pf(v), _pi(i))
        return v[:n].copy(), i[:n].copy()


def scan_list(q, vecs, ids, buf, metric, squared_domain=False, fast=False):
    q, vecs = _f32(q), _f32(vecs).reshape(-1, q.shape[0]) if len(vecs) else np.zeros((0, len(q)), np.float32)
    ids = _i64(ids) if ids is not None else None
    n, d = vecs.shape
    if fast:
        assert squared_domain
        lib().qo_scan_list_fast(_pf(q), _pf(vecs), _pi(ids), n, d, buf.h, metric_code(metric))
    else:
        lib().qo_scan_list(_pf(q), _pf(vecs), _pi(ids), n, d, buf.h, metric_code(metric), int(squared_domain))


def batched_scan_list(queries, vecs, ids, bufs, metric, squared_domain=False):
    queries = _f32(queries)
    nq, d = queries.shape
    vecs = _f32(vecs).reshape(-1, d)
    ids = _i64(ids) if ids is not None else None
    arr = (C.c_void_p * nq)(*[b.h for b in bufs])
    lib().qo_batched_scan_list(_pf(queries), _pf(vecs) if vecs.shape[0] else None, _pi(ids), nq, vecs.shape[0], d, arr,
                               metric_code(metric), int(squared_domain))


def csr_from_partitions(part_vecs, part_ids, d):
    """list of [n_p,d] arrays + list of [n_p] id arrays -> (vecs, ids, offsets) CSR arena."""
    sizes = [len(i) for i in part_ids]
    offsets = np.zeros(len(sizes) + 1, np.int64)
    offsets[1:] = np.cumsum(sizes)
    vecs = np.concatenate([_f32(v).reshape(-1, d) for v in part_vecs], 0) if sizes else np.zeros((0, d), np.float32)
    ids = np.concatenate([_i64(i) for i in part_ids]) if sizes else np.zeros(0, np.int64)
    return _f32(vecs), _i64(ids), offsets


def _prep_pids(pids, nq):
    pids = _i64(pids)
    if pids.ndim == 1:  # same set for every query (query_coordinator.cpp:506-508)
        pids = np.ascontiguousarray(np.broadcast_to(pids[None, :], (nq, pids.shape[0])))
    return pids


def serial_scan(x, vecs, ids, offsets, pids, k, metric, num_threads=1, fast=True):
    x, vecs, ids, offsets = _f32(x), _f32(vecs), _i64(ids), _i64(offsets)
    nq, d = x.shape
    pids = _prep_pids(pids, nq)
    k = k if k > 0 else 1
    out_i = np.empty((nq, k), np.int64)
    out_d = np.empty((nq, k), np.float32)
    lib().qo_serial_scan(_pf(x), nq, _pf(vecs), _pi(ids), _pi(offsets), offsets.shape[0] - 1, d, _pi(pids),
                         pids.shape[1], k, metric_code(metric), num_threads, int(fast), _pi(out_i), _pf(out_d))
    return out_i, out_d


def batched_serial_scan(x, vecs, ids, offsets, pids, k, metric, num_threads=1):
    x, vecs, ids, offsets = _f32(x), _f32(vecs), _i64(ids), _i64(offsets)
    nq, d = x.shape
    pids = _prep_pids(pids, nq)
    k = k if k > 0 else 1
    out_i = np.empty((nq, k), np.int64)
    out_d = np.empty((nq, k), np.float32)
    lib().qo_batched_serial_scan(_pf(x), nq, _pf(vecs), _pi(ids), _pi(offsets), offsets.shape[0] - 1, d, _pi(pids),
                                 pids.shape[1], k, metric_code(metric), num_threads, _pi(out_i), _pf(out_d))
    return out_i, out_d


def coarse(x, centroids, centroid_ids, nprobe, metric, num_threads=1):
    x, centroids = _f32(x), _f32(centroids)
    nq, d = x.shape
    nlist = centroids.shape[0]
    kk = min(nprobe, nlist)
    cids = _i64(centroid_ids) if centroid_ids is not None else None
    out_p = np.empty((nq, max(kk, 0)), np.int64)
    out_d = np.empty((nq, max(kk, 0)), np.float32)
    if kk > 0:
        lib().qo_coarse(_pf(x), nq, _pf(centroids), _pi(cids), nlist, d, nprobe, metric_code(metric), num_threads,
                        _pi(out_p), _pf(out_d))
    return out_p, out_d


def search(x, centroids, vecs, ids, offsets, nprobe, k, metric, batched_scan=False, num_threads=1, fast=True,
           centroid_ids=None):
    x, vecs, ids, offsets = _f32(x), _f32(vecs), _i64(ids), _i64(offsets)
    nq, d = x.shape
    centroids = _f32(centroids) if centroids is not None else None
    cids = _i64(centroid_ids) if centroid_ids is not None else None
    k = k if k > 0 else 1
    out_i = np.empty((nq, k), np.int64)
    out_d = np.empty((nq, k), np.float32)
    lib().qo_search(_pf(x), nq, _pf(centroids), _pi(cids), _pf(vecs), _pi(ids), _pi(offsets), offsets.shape[0] - 1, d,
                    nprobe, k, metric_code(metric), int(batched_scan), num_threads, int(fast), _pi(out_i), _pf(out_d))
    return out_i, out_d


def effective_cores(nthreads=0, iters=40_000_000):
    """aggregate FMA rate on `nthreads` threads / rate on one thread (what the host really gives this process)"""
    L = lib()
    L.qo_effective_cores.restype = C.c_double
    L.qo_effective_cores.argtypes = [C.c_int, C.c_long]
    return float(L.qo_effective_cores(int(nthreads), int(iters)))


def rand_perm(n, m, seed):
    out = np.empty(min(n, m), np.int64)
    lib().qo_rand_perm(n, m, seed, _pi(out))
    return out


def kmeans_assign(x, c, metric, num_threads=0):
    x, c = _f32(x), _f32(c)
    n, d = x.shape
    a = np.empty(n, np.int64)
    v = np.empty(n, np.float32)
    lib().qo_kmeans_assign(_pf(x), n, _pf(c), c.shape[0], d, metric_code(metric), num_threads, _pi(a), _pf(v))
    return a, v


def kmeans_accumulate(x, assign, m, blocked=False):
    """per-centroid sums / counts.  blocked=False: rows added one after the other (the reference's refine loop,
    clustering.cpp:162-176); True: the blocked canonical order of the k-means driver's mean update (32-row blocks, 32-block
    groups, groups in order)."""
    x, assign = _f32(x), _i64(assign)
    n, d = x.shape
    sums = np.empty((m, d), np.float32)
    counts = np.empty(m, np.int64)
    (lib().qo_kmeans_accumulate_blocked if blocked else lib().qo_kmeans_accumulate)(_pf(x), n, d, _pi(assign), m, _pf(sums), _pi(counts))
    return sums, counts


def kmeans_update(sums, counts, centroids):
    """centroids = sums / counts (empty clusters keep theirs), then the deterministic empty-cluster split; returns
    (centroids, counts) -- new arrays."""
    sums, counts, c = _f32(sums), _i64(counts).copy(), _f32(centroids).copy()
    m, d = c.shape
    lib().qo_kmeans_update(_pf(sums), _pi(counts), m, d, _pf(c))
    return c, counts


def normalize_rows(x):
    x = _f32(x).copy()
    lib().qo_normalize_rows(_pf(x), x.shape[0], x.shape[1])
    return x


def kmeans(x, m, metric, niter=5, seed=1234, num_threads=0):
    """Returns (centroids, assign, x_used) -- x_used is the normalised copy for IP (clustering.cpp:25-26)."""
    x = _f32(x).copy()
    n, d = x.shape
    c = np.empty((m, d), np.float32)
    a = np.empty(n, np.int64)
    lib().qo_kmeans(_pf(x), n, d, m, metric_code(metric), niter, seed, num_threads, _pf(c), _pi(a))
    return c, a, x


def kmeans_refine_partitions(centroids, vecs, ids, offsets, metric, refinement_iterations=0, num_threads=0):
    c = _f32(centroids).copy()
    vecs, ids, offsets = _f32(vecs), _i64(ids), _i64(offsets)
    m, d = c.shape
    ov = np.empty_like(vecs)
    oi = np.empty_like(ids)
    oo = np.empty(m + 1, np.int64)
    lib().qo_kmeans_refine_partitions(_pf(c), m, d, _pf(vecs), _pi(ids), _pi(offsets), metric_code(metric),
                                      refinement_iterations, num_threads, _pf(ov), _pi(oi), _pi(oo))
    return c, ov, oi, oo


def recall(ids, gt, set_semantics=True):
    """set_semantics=True: src/python/utils.py compute_recall; False: C++ calculate_recall (list_scanning.h:14-37),
    which counts a duplicated returned id once per occurrence."""
    ids, gt = _i64(ids), _i64(gt)
    nq, k = ids.shape
    out = np.empty(nq, np.float32)
    gtk = np.ascontiguousarray(gt[:, :k])
    (lib().qo_recall_set if set_semantics else lib().qo_recall)(_pi(ids), _pi(gtk), nq, k, _pf(out))
    return out


# ---- adaptive partition scanning (geometry.h + serial_scan's use_aps branch) -----------------------------------------
def incomplete_beta(a, b, x):
    return float(lib().qo_incomplete_beta(float(a), float(b), float(x)))


def incomplete_beta_table(d):
    t = np.empty(1001, np.float64)
    lib().qo_incomplete_beta_table(int(d), t.ctypes.data_as(C.POINTER(C.c_double)))
    return t


def incomplete_beta_lookup(table, x):
    table = np.ascontiguousarray(table, np.float64)
    return float(lib().qo_incomplete_beta_lookup(table.ctypes.data_as(C.POINTER(C.c_double)), float(x)))


def log_cap_volume(radius, boundary_distance, d, use_precomputed=True, euclidean=True, table=None):
    if table is None:
        table = incomplete_beta_table(d)
    table = np.ascontiguousarray(table, np.float64)
    return float(lib().qo_log_cap_volume(float(radius), float(boundary_distance), int(d), int(use_precomputed), int(euclidean),
                                         table.ctypes.data_as(C.POINTER(C.c_double))))


def recall_profile(boundary_distances, query_radius, d, use_precomputed=True, euclidean=True):
    bd = _f32(boundary_distances)
    table = incomplete_beta_table(d)
    out = np.empty(bd.shape[0], np.float32)
    rc = lib().qo_recall_profile(_pf(bd), bd.shape[0], float(query_radius), int(d), int(use_precomputed), int(euclidean),
                                 table.ctypes.data_as(C.POINTER(C.c_double)), _pf(out))
    if rc != 0:
        raise RuntimeError("Boundary distances must have at least 2 partitions to create an estimate.")  # geometry.h:350
    return out


def boundary_distances(q, centroids, euclidean=True):
    """centroids [M][d] in rank order (row 0 = nearest)."""
    q = _f32(q)
    c = _f32(centroids)
    M, d = c.shape
    ptrs = (_f32p * M)(*[c[j].ctypes.data_as(_f32p) for j in range(M)])
    out = np.empty(M, np.float32)
    lib().qo_boundary_distances(_pf(q), ptrs, M, d, int(euclidean), _pf(out))
    return out


def search_aps(x, centroids, vecs, ids, offsets, k, metric, recall_target, recompute_threshold=0.001, use_precomputed=True,
               initial_search_fraction=0.02, centroid_ids=None, expanded=True, num_threads=1):
    """QuakeIndex::search with SearchParams.recall_target > 0 (serial_scan APS).  Returns (ids, dist, nscanned)."""
    x = _f32(x)
    centroids = _f32(centroids)
    vecs = _f32(vecs)
    ids = _i64(ids)
    offsets = _i64(offsets)
    nq, d = x.shape
    nlist = offsets.shape[0] - 1
    k = max(int(k), 1)
    cid = _i64(centroid_ids) if centroid_ids is not None else None
    out_i = np.empty((nq, k), np.int64)
    out_d = np.empty((nq, k), np.float32)
    out_n = np.zeros(nq, np.int32)
    rc = lib().qo_search_aps(_pf(x), nq, _pf(centroids), _pi(cid), centroids.shape[0], _pf(vecs), _pi(ids), _pi(offsets), nlist, d,
                             k, metric_code(metric), float(recall_target), float(recompute_threshold), int(use_precomputed),
                             float(initial_search_fraction), int(expanded), int(num_threads), _pi(out_i), _pf(out_d),
                             out_n.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise RuntimeError("Boundary distances must have at least 2 partitions to create an estimate.")
    return out_i, out_d, out_n
