"""Thin object wrappers over the C ABI (include/quake_hip.h) for Python callers and the test-suite.

Host data is passed as numpy arrays (QK_MEM_HOST); device data as torch CUDA tensors (QK_MEM_DEVICE).
Every call goes through libquake_hip.so -- there is no alternative code path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import QK_MEM_DEVICE, QK_MEM_HOST, QK_METRIC_IP, QK_METRIC_L2, QkTiming, check


def metric_code(metric):
    """str_to_metric_type (src/cpp/include/common.h:145-156)."""
    if isinstance(metric, str):
        m = metric.lower()
        if m == "l2":
            return QK_METRIC_L2
        if m == "ip":
            return QK_METRIC_IP
        raise ValueError("Invalid metric type: " + metric)
    return int(metric)


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


def _mem_of(*arrs):
    kinds = set()
    for a in arrs:
        if a is None:
            continue
        if _is_torch(a):
            kinds.add(QK_MEM_DEVICE if a.is_cuda else QK_MEM_HOST)
        else:
            kinds.add(QK_MEM_HOST)
    if len(kinds) > 1:
        raise ValueError("mixed host/device arguments")
    return kinds.pop() if kinds else QK_MEM_HOST


def _f32(a):
    if _is_torch(a):
        import torch
        return a.contiguous().to(torch.float32)
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    if _is_torch(a):
        import torch
        return a.contiguous().to(torch.int64)
    return np.ascontiguousarray(a, dtype=np.int64)


def _empty_like_mem(shape, dtype, ref):
    if _is_torch(ref) and ref.is_cuda:
        import torch
        tdt = torch.int64 if dtype == np.int64 else torch.int32 if dtype == np.int32 else torch.float32
        return torch.empty(shape, dtype=tdt, device=ref.device)
    return np.empty(shape, dtype)


def timing_dict(t):
    return {f: getattr(t, f) for f, _ in QkTiming._fields_}


class Context:
    def __init__(self, device=0):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        check(self.lib.qk_ctx_create(int(device), C.byref(self.h)))
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.lib.qk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.qk_ctx_synchronize(self.h))

    def set_form_feedback(self, enabled):
        """measured choice between the scan forms per batch shape (default on); off = the static rule alone"""
        check(self.lib.qk_ctx_set_form_feedback(self.h, int(bool(enabled))))

    def set_form_times(self, ms3):
        """the feedback rule on injected figures: ms3 = (tile form, per-wave walk, mixed) in ms, or None for measured times"""
        if ms3 is None:
            check(self.lib.qk_ctx_set_form_times(self.h, None))
        else:
            check(self.lib.qk_ctx_set_form_times(self.h, (C.c_float * 3)(*[float(v) for v in ms3])))

    def set_stream(self, hip_stream):
        """hip_stream: a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream); 0 = the device's NULL stream
        (torch's default stream); None = back to the context's private stream.
        Calls on DEVICE buffers (torch CUDA tensors in, torch CUDA tensors out) are ordered on the context's stream only: the private
        stream is non-blocking, so a caller that mixes them with torch operations binds the context to torch's current stream first
        (GpuEngine and the bindings do) -- otherwise torch may read an output before the library's kernel wrote it."""
        if hip_stream is None:
            check(self.lib.qk_ctx_set_stream(self.h, None))
        elif int(hip_stream) == 0:
            check(self.lib.qk_ctx_set_null_stream(self.h))
        else:
            check(self.lib.qk_ctx_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def set_timing(self, mode=1):
        """0 off, 1 per-call (synchronising), 2 deferred (read with read_timing()), 3 deferred with one event pair around
        the scan kernel only (what bench.py keeps on inside its timed region)."""
        check(self.lib.qk_ctx_set_timing(self.h, int(mode)))

    def get_timing(self):
        """the timing mode set last (qk_ctx_get_timing)"""
        m = C.c_int(0)
        check(self.lib.qk_ctx_get_timing(self.h, C.byref(m)))
        return int(m.value)

    def read_timing(self):
        t, n = QkTiming(), C.c_int64()
        check(self.lib.qk_ctx_read_timing(self.h, C.byref(t), C.byref(n)))
        out = timing_dict(t)
        out["calls"] = n.value
        return out

    def set_squared_l2(self, on=True):
        check(self.lib.qk_ctx_set_squared_l2(self.h, int(bool(on))))

    def last_scan_kernel(self):
        """name of the partition-scan kernel the last scan / search on this context launched"""
        buf = C.create_string_buffer(64)
        check(self.lib.qk_ctx_last_scan_kernel(self.h, buf, 64))
        return buf.value.decode()

    def device_info(self):
        cus, clk, hbm = C.c_int(), C.c_int(), C.c_int64()
        arch = C.create_string_buffer(64)
        check(self.lib.qk_ctx_device_info(self.h, C.byref(cus), C.byref(clk), C.byref(hbm), arch, 64))
        return dict(num_cus=cus.value, clock_khz=clk.value, hbm_bytes=hbm.value, arch=arch.value.decode())

    # ---- search ---------------------------------------------------------------------------------------
    def coarse(self, parent, x, nprobe, metric, values=True):
        """the nprobe nearest lists of every row (and, with values, their distances; values=False passes out_dist = NULL)"""
        x = _f32(x)
        Q = x.shape[0]
        kk = int(min(nprobe, parent.ntotal()))
        mem = _mem_of(x)
        out_p = _empty_like_mem((Q, kk), np.int64, x)
        out_d = _empty_like_mem((Q, kk), np.float32, x) if values else None
        check(self.lib.qk_coarse(self.h, parent.h, _ptr(x), Q, int(nprobe), metric_code(metric), _ptr(out_p),
                                 _ptr(out_d) if values else None, mem))
        return out_p, out_d

    def scan(self, store, x, pids, k, metric, timing=False):
        x, pids = _f32(x), _i64(pids)
        Q = x.shape[0]
        if pids.ndim == 1:  # same set for every query (query_coordinator.cpp:506-508)
            pids = (pids[None, :].expand(Q, -1).contiguous() if _is_torch(pids)
                    else np.ascontiguousarray(np.broadcast_to(pids[None, :], (Q, pids.shape[0]))))
        mem = _mem_of(x, pids)
        out_i = _empty_like_mem((Q, k), np.int64, x)
        out_d = _empty_like_mem((Q, k), np.float32, x)
        t = QkTiming()
        check(self.lib.qk_scan(self.h, store.h, _ptr(x), Q, _ptr(pids) if pids.shape[1] > 0 else None, int(pids.shape[1]),
                               int(k), metric_code(metric), _ptr(out_i), _ptr(out_d), mem, C.byref(t) if timing else None))
        return (out_i, out_d, timing_dict(t)) if timing else (out_i, out_d)

    def scan_into(self, store, x, pids, k, metric, out):
        """scan() writing into caller-provided (ids, dist) tensors (no allocation in the timed loop)."""
        x, pids = _f32(x), _i64(pids)
        Q = x.shape[0]
        out_i, out_d = out
        check(self.lib.qk_scan(self.h, store.h, _ptr(x), Q, _ptr(pids), int(pids.shape[1]), int(k), metric_code(metric),
                               _ptr(out_i), _ptr(out_d), _mem_of(x, pids), None))
        return out_i, out_d

    def search(self, parent, store, x, nprobe, k, metric, timing=False, out=None):
        x = _f32(x)
        Q = x.shape[0]
        mem = _mem_of(x)
        if out is None:
            out_i = _empty_like_mem((Q, k), np.int64, x)
            out_d = _empty_like_mem((Q, k), np.float32, x)
        else:
            out_i, out_d = out
        t = QkTiming()
        check(self.lib.qk_search(self.h, parent.h if parent is not None else None, store.h, _ptr(x), Q, int(nprobe), int(k),
                                 metric_code(metric), _ptr(out_i), _ptr(out_d), mem, C.byref(t) if timing else None))
        return (out_i, out_d, timing_dict(t)) if timing else (out_i, out_d)

    def search_tracked(self, parent, store, x, nprobe, k, metric, timing=False):
        """qk_search_tracked: search + the [Q, min(nprobe, parent lists)] list numbers every query scanned, one enqueue.
        Returns (ids, dist, probed[, timing])."""
        x = _f32(x)
        Q = x.shape[0]
        mem = _mem_of(x)
        width = max(min(int(nprobe), int(parent.ntotal())), 0)
        out_i = _empty_like_mem((Q, k), np.int64, x)
        out_d = _empty_like_mem((Q, k), np.float32, x)
        out_p = _empty_like_mem((Q, max(width, 1)), np.int64, x)
        t = QkTiming()
        check(self.lib.qk_search_tracked(self.h, parent.h, store.h, _ptr(x), Q, int(nprobe), int(k), metric_code(metric),
                                         _ptr(out_i), _ptr(out_d), _ptr(out_p), mem, C.byref(t) if timing else None))
        out_p = out_p[:, :width]
        return (out_i, out_d, out_p, timing_dict(t)) if timing else (out_i, out_d, out_p)

    def search_aps(self, parent, store, x, k, metric, recall_target, recompute_threshold=0.001, use_precomputed=True,
                   initial_search_fraction=0.02, timing=False):
        """recall-target search (adaptive partition scanning).  Returns (ids, dist, nscanned[, timing])."""
        x = _f32(x)
        Q = x.shape[0]
        k = max(int(k), 1)
        mem = _mem_of(x)
        out_i = _empty_like_mem((Q, k), np.int64, x)
        out_d = _empty_like_mem((Q, k), np.float32, x)
        out_n = _empty_like_mem((Q,), np.int32, x)
        t = QkTiming()
        check(self.lib.qk_search_aps(self.h, parent.h if parent is not None else None, store.h, _ptr(x), Q, k, metric_code(metric),
                                     float(recall_target), float(recompute_threshold), int(bool(use_precomputed)),
                                     float(initial_search_fraction), _ptr(out_i), _ptr(out_d), _ptr(out_n), mem,
                                     C.byref(t) if timing else None))
        return (out_i, out_d, out_n, timing_dict(t)) if timing else (out_i, out_d, out_n)

    def merge_topk(self, ids, keys, metric):
        """ids/keys: CUDA tensors [G][Q][k] (keys = squared L2 / dot); returns ([Q][k] ids, [Q][k] distances)."""
        import torch
        ids, keys = _i64(ids), _f32(keys)
        G, Q, k = ids.shape
        out_i = torch.empty((Q, k), dtype=torch.int64, device=ids.device)
        out_d = torch.empty((Q, k), dtype=torch.float32, device=ids.device)
        check(self.lib.qk_merge_topk(self.h, _ptr(ids), _ptr(keys), G, Q, k, metric_code(metric), _ptr(out_i), _ptr(out_d)))
        return out_i, out_d

    def topk_block_bytes(self, per, k):
        return int(self.lib.qk_topk_block_bytes(int(per), int(k)))

    def pack_topk(self, ids, keys, G, out=None):
        """ids/keys: CUDA tensors [G*per][k] -> uint8 [G][block] send buffer of the one-collective exchange (block j = the
        entries of queries [j*per, (j+1)*per): ids then keys)."""
        import torch
        ids, keys = _i64(ids), _f32(keys)
        Q, k = ids.shape
        per = Q // int(G)
        blk = self.topk_block_bytes(per, k)
        if out is None or tuple(out.shape) != (int(G), blk) or out.device != ids.device:
            out = torch.empty((int(G), blk), dtype=torch.uint8, device=ids.device)
        check(self.lib.qk_pack_topk(self.h, _ptr(ids), _ptr(keys), int(G), per, k, _ptr(out)))
        return out

    def merge_topk_packed(self, packed, per, k, metric):
        """packed: uint8 [G][block] receive buffer (block r = rank r's entries for this rank's `per` queries)."""
        import torch
        G = packed.shape[0]
        out_i = torch.empty((per, k), dtype=torch.int64, device=packed.device)
        out_d = torch.empty((per, k), dtype=torch.float32, device=packed.device)
        check(self.lib.qk_merge_topk_packed(self.h, _ptr(packed), G, int(per), int(k), metric_code(metric), _ptr(out_i), _ptr(out_d)))
        return out_i, out_d

    # ---- k-means ----------------------------------------------------------------------------------------
    def kmeans_assign(self, x, c, metric, values=True):
        """nearest centroid of every row (and, with values, its distance / dot product); values=False passes val = NULL, the
        form the Lloyd driver uses (rows with a single candidate skip the exact key: qk_assign_pf.hip)."""
        x, c = _f32(x), _f32(c)
        n, d = x.shape
        mem = _mem_of(x, c)
        a = _empty_like_mem((n,), np.int64, x)
        v = _empty_like_mem((n,), np.float32, x) if values else None
        check(self.lib.qk_kmeans_assign(self.h, _ptr(x), n, _ptr(c), c.shape[0], d, metric_code(metric), _ptr(a),
                                        _ptr(v) if values else None, mem))
        return a, v

    def kmeans_assign_only(self, x, c, metric):
        """the assignments alone (val = NULL): what a Lloyd iteration needs"""
        return self.kmeans_assign(x, c, metric, values=False)[0]

    def kmeans_accumulate(self, x, assign, m, blocked=False):
        """per-centroid sums and counts.  blocked=False: rows added one after the other (the reference's refine loop,
        clustering.cpp:162-176); True: the blocked canonical order of the Lloyd driver (qk_kmeans_accumulate_blocked)."""
        x, assign = _f32(x), _i64(assign)
        n, d = x.shape
        mem = _mem_of(x, assign)
        sums = _empty_like_mem((m, d), np.float32, x)
        counts = _empty_like_mem((m,), np.int64, x)
        fn = self.lib.qk_kmeans_accumulate_blocked if blocked else self.lib.qk_kmeans_accumulate
        check(fn(self.h, _ptr(x), n, d, _ptr(assign), m, _ptr(sums), _ptr(counts), mem))
        return sums, counts

    def normalize_rows(self, x):
        """x /= ||x|| row by row, in place (canonical norm); returns x."""
        check(self.lib.qk_normalize_rows(self.h, _ptr(x), x.shape[0], x.shape[1], _mem_of(x)))
        return x

    def kmeans_update(self, sums, counts, centroids):
        """mean update + empty-cluster split of one Lloyd iteration; `centroids` and `counts` are updated in place."""
        m, d = centroids.shape
        check(self.lib.qk_kmeans_update(self.h, _ptr(sums), _ptr(counts), m, d, _ptr(centroids), _mem_of(sums, counts, centroids)))
        return centroids, counts

    def rand_perm(self, n, m, seed):
        out = np.empty(min(int(n), int(m)), np.int64)
        check(self.lib.qk_rand_perm(int(n), int(m), int(seed), _ptr(out)))
        return out

    def kmeans(self, x, m, metric, niter=5, seed=1234):
        """Returns (centroids, assign, x_used); x_used is the (IP-normalised) copy the caller should store."""
        x = _f32(x)
        x = x.clone() if _is_torch(x) else x.copy()
        n, d = x.shape
        mem = _mem_of(x)
        c = _empty_like_mem((m, d), np.float32, x)
        a = _empty_like_mem((n,), np.int64, x)
        check(self.lib.qk_kmeans(self.h, _ptr(x), n, d, m, metric_code(metric), int(niter), int(seed), _ptr(c), _ptr(a), mem))
        return c, a, x

    def kmeans_inplace(self, x, m, metric, centroids, assign, niter=5, seed=1234):
        """qk_kmeans on CUDA tensors the caller owns: x [n, d] (IP: normalised IN PLACE), centroids [m, d] and assign [n] written"""
        n, d = x.shape
        check(self.lib.qk_kmeans(self.h, _ptr(x), n, d, int(m), metric_code(metric), int(niter), int(seed), _ptr(centroids), _ptr(assign),
                                 QK_MEM_DEVICE))

    def kmeans_last_timing(self):
        """kernel-side ms of the assign / update step of the last Lloyd iteration of the last kmeans() (qk_kmeans_last_timing)"""
        am, um = C.c_float(0), C.c_float(0)
        rows, m = C.c_int64(0), C.c_int64(0)
        check(self.lib.qk_kmeans_last_timing(self.h, C.byref(am), C.byref(um), C.byref(rows), C.byref(m)))
        return dict(assign_ms=am.value, update_ms=um.value, rows=rows.value, m=m.value)


class Store:
    """Device mirror of faiss::DynamicInvertedLists (dynamic_inverted_list.h:25-33)."""

    def __init__(self, ctx, d):
        self.ctx = ctx
        self.lib = ctx.lib
        self.h = C.c_void_p()
        check(self.lib.qk_store_create(ctx.h, int(d), C.byref(self.h)))
        self.d = int(d)

    def close(self):
        if getattr(self, "h", None) and self.h and self.ctx.h:
            self.lib.qk_store_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self.lib.qk_store_reset(self.h))

    def add_list(self, list_no):
        check(self.lib.qk_store_add_list(self.h, int(list_no)))

    def remove_list(self, list_no):
        check(self.lib.qk_store_remove_list(self.h, int(list_no)))

    def add_entries(self, list_no, ids, vecs):
        ids, vecs = _i64(ids), _f32(vecs)
        n = ids.shape[0]
        check(self.lib.qk_store_add_entries(self.h, int(list_no), n, _ptr(ids), _ptr(vecs), _mem_of(ids, vecs)))

    def add_batch(self, ids, vecs, assign):
        ids, vecs, assign = _i64(ids), _f32(vecs), _i64(assign)
        check(self.lib.qk_store_add_batch(self.h, ids.shape[0], _ptr(ids), _ptr(vecs), _ptr(assign), _mem_of(ids, vecs, assign)))

    def build_csr(self, offsets, ids, vecs):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        ids, vecs = _i64(ids), _f32(vecs)
        check(self.lib.qk_store_build_csr(self.h, offsets.shape[0] - 1, _ptr(offsets), _ptr(ids), _ptr(vecs),
                                          _mem_of(ids, vecs)))

    def remove_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        n = C.c_int64()
        check(self.lib.qk_store_remove_ids(self.h, ids.shape[0], _ptr(ids), C.byref(n)))
        return n.value

    def list_size(self, list_no):
        out = C.c_int64()
        check(self.lib.qk_store_list_size(self.h, int(list_no), C.byref(out)))
        return out.value

    def list_sizes(self, list_nos):
        """sizes of many lists in one call -> int64 array"""
        nos = np.ascontiguousarray(list_nos, dtype=np.int64).reshape(-1)
        out = np.empty(nos.shape[0], np.int64)
        if nos.shape[0]:
            check(self.lib.qk_store_list_sizes(self.h, _ptr(nos), nos.shape[0], _ptr(out)))
        return out

    def ntotal(self):
        return self.lib.qk_store_ntotal(self.h)

    def nlist(self):
        return self.lib.qk_store_nlist(self.h)

    def list_ids(self):
        n = C.c_int64()
        check(self.lib.qk_store_list_ids(self.h, None, C.byref(n)))
        out = np.empty(n.value, np.int64)
        if n.value:
            check(self.lib.qk_store_list_ids(self.h, _ptr(out), C.byref(n)))
        return out

    def get_list(self, list_no):
        n = self.list_size(list_no)
        vecs = np.empty((n, self.d), np.float32)
        ids = np.empty(n, np.int64)
        check(self.lib.qk_store_get_list(self.h, int(list_no), _ptr(vecs), _ptr(ids), QK_MEM_HOST))
        return vecs, ids

    def get_list_ids(self, list_no):
        """ids [n] of a list in row order, from the store's host mirror (qk_store_get_list with vecs = NULL: nothing is extracted)"""
        n = self.list_size(list_no)
        ids = np.empty(n, np.int64)
        if n:
            check(self.lib.qk_store_get_list(self.h, int(list_no), None, _ptr(ids), QK_MEM_HOST))
        return ids

    def get_list_device(self, list_no):
        """(vectors [n, d] as a CUDA tensor on the store's device -- extracted from the tile-major arena on the context's stream,
        no host hop -- , ids [n] as a host array: the store's own id mirror)"""
        import torch
        n = self.list_size(list_no)
        dev = torch.device("cuda", self.ctx.device)
        vecs = torch.empty((n, self.d), dtype=torch.float32, device=dev)
        ids = np.empty(n, np.int64)
        if n:
            check(self.lib.qk_store_get_list(self.h, int(list_no), _ptr(vecs), None, QK_MEM_DEVICE))
            check(self.lib.qk_store_get_list(self.h, int(list_no), None, _ptr(ids), QK_MEM_HOST))
        return vecs, ids

    def get_vectors(self, ids):
        """vectors of many ids in one call -> (float32 [n, d], bool [n] found)"""
        ids = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        out = np.empty((ids.shape[0], self.d), np.float32)
        found = np.zeros(ids.shape[0], np.int32)
        if ids.shape[0]:
            check(self.lib.qk_store_get_vectors(self.h, _ptr(ids), ids.shape[0], _ptr(out), _ptr(found)))
        return out, found.astype(bool)

    def publish(self):
        """pending modifications become visible to searches now (list table upload, row-major copy) instead of inside the next query"""
        check(self.lib.qk_store_publish(self.h))

    def get_lists_device(self, list_nos, with_ids=False):
        """rows of many lists, one list after the other, as ONE CUDA tensor [sum of sizes, d] (qk_store_get_lists) + the sizes
        (+ with_ids: the rows' ids as a CUDA tensor)"""
        import torch
        nos = np.ascontiguousarray(list_nos, dtype=np.int64).reshape(-1)
        sizes = self.list_sizes(nos)
        dev = torch.device("cuda", self.ctx.device)
        vecs = torch.empty((int(sizes.sum()), self.d), dtype=torch.float32, device=dev)
        ids = torch.empty((vecs.shape[0],), dtype=torch.int64, device=dev) if with_ids else None
        if vecs.shape[0]:
            check(self.lib.qk_store_get_lists(self.h, _ptr(nos), nos.shape[0], _ptr(vecs), _ptr(ids) if with_ids else None, QK_MEM_DEVICE))
        return (vecs, sizes, ids) if with_ids else (vecs, sizes)

    def get_vector(self, vid):
        out = np.empty(self.d, np.float32)
        found = C.c_int()
        check(self.lib.qk_store_get_vector(self.h, int(vid), _ptr(out), C.byref(found)))
        return out if found.value else None

    def refine_lists(self, list_nos, centroids, metric, refinement_iterations=0):
        """kmeans_refine_partitions on the device store; returns the centroids used for the last assignment."""
        list_nos = np.ascontiguousarray(list_nos, dtype=np.int64)
        c = _f32(centroids)
        c = c.clone() if _is_torch(c) else c.copy()
        check(self.lib.qk_store_refine_lists(self.h, _ptr(list_nos), list_nos.shape[0], _ptr(c), metric_code(metric),
                                             int(refinement_iterations), _mem_of(c)))
        return c

    def device_bytes(self):
        return self.lib.qk_store_device_bytes(self.h)

    COUNTER_NAMES = ("arena_reallocations", "arena_compactions", "list_relocations", "rows_copied", "rowmajor_rebuilds",
                     "table_uploads", "id_index_rebuilds", "context_scratch_reallocations")

    def counters(self):
        """what mutations cost beyond the rows they wrote (qk_store_counters), as a dict"""
        out = np.zeros(8, np.int64)
        check(self.lib.qk_store_counters(self.h, _ptr(out), 8))
        return dict(zip(self.COUNTER_NAMES, (int(v) for v in out)))


class Group:
    """Device group (qk_group_*): IndexBuildParams.num_workers as GPUs.  One process drives G members -- a context and a shard
    store each, list p in member p % G -- behind the Store surface (same method names; every call is routed to the member that
    holds the list) plus search() / scan() over the members.  `devices`: one HIP ordinal per member (repeats allowed)."""

    def __init__(self, devices, d):
        self.lib = _lib.load()
        devs = [int(v) for v in devices]
        arr = (C.c_int * len(devs))(*devs)
        self.h = C.c_void_p()
        check(self.lib.qk_group_create(arr, len(devs), int(d), C.byref(self.h)))
        self.devices = devs
        self.device = devs[0]
        self.d = int(d)

    def close(self):
        if getattr(self, "h", None) and self.h:
            self.lib.qk_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self.lib.qk_group_size(self.h))

    def owner(self, list_no):
        return int(self.lib.qk_group_owner(self.h, int(list_no)))

    def member_handles(self, i):
        """(qk_ctx*, qk_store*) of member i (borrowed)"""
        c, s = C.c_void_p(), C.c_void_p()
        check(self.lib.qk_group_member(self.h, int(i), C.byref(c), C.byref(s)))
        return c, s

    def member_list_ids(self, i):
        """list numbers member i holds"""
        _, s = self.member_handles(i)
        n = C.c_int64()
        check(self.lib.qk_store_list_ids(s, None, C.byref(n)))
        out = np.empty(n.value, np.int64)
        if n.value:
            check(self.lib.qk_store_list_ids(s, _ptr(out), C.byref(n)))
        return out

    def set_stream(self, hip_stream):
        """the lead's stream (see Context.set_stream)"""
        if hip_stream is None:
            check(self.lib.qk_group_set_stream(self.h, None))
        elif int(hip_stream) == 0:
            check(self.lib.qk_group_set_null_stream(self.h))
        else:
            check(self.lib.qk_group_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        check(self.lib.qk_group_synchronize(self.h))

    def set_form_feedback(self, enabled):
        check(self.lib.qk_group_set_form_feedback(self.h, int(bool(enabled))))

    def set_submit_threads(self, enabled):
        """one persistent enqueue thread per member (default) / everything on the caller's thread"""
        check(self.lib.qk_group_set_submit_threads(self.h, int(bool(enabled))))

    # ---- the Store surface ------------------------------------------------------------------------------------------------
    def reset(self):
        check(self.lib.qk_group_reset(self.h))

    def add_list(self, list_no):
        check(self.lib.qk_group_add_list(self.h, int(list_no)))

    def remove_list(self, list_no):
        check(self.lib.qk_group_remove_list(self.h, int(list_no)))

    def add_entries(self, list_no, ids, vecs):
        ids, vecs = _i64(ids), _f32(vecs)
        check(self.lib.qk_group_add_entries(self.h, int(list_no), ids.shape[0], _ptr(ids), _ptr(vecs), _mem_of(ids, vecs)))

    def add_batch(self, ids, vecs, assign):
        ids, vecs, assign = _i64(ids), _f32(vecs), _i64(assign)
        check(self.lib.qk_group_add_batch(self.h, ids.shape[0], _ptr(ids), _ptr(vecs), _ptr(assign), _mem_of(ids, vecs, assign)))

    def build_csr(self, offsets, ids, vecs):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        ids, vecs = _i64(ids), _f32(vecs)
        check(self.lib.qk_group_build_csr(self.h, offsets.shape[0] - 1, _ptr(offsets), _ptr(ids), _ptr(vecs), _mem_of(ids, vecs)))

    def remove_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        n = C.c_int64()
        check(self.lib.qk_group_remove_ids(self.h, ids.shape[0], _ptr(ids), C.byref(n)))
        return n.value

    def list_size(self, list_no):
        out = C.c_int64()
        check(self.lib.qk_group_list_size(self.h, int(list_no), C.byref(out)))
        return out.value

    def list_sizes(self, list_nos):
        nos = np.ascontiguousarray(list_nos, dtype=np.int64).reshape(-1)
        out = np.empty(nos.shape[0], np.int64)
        if nos.shape[0]:
            check(self.lib.qk_group_list_sizes(self.h, _ptr(nos), nos.shape[0], _ptr(out)))
        return out

    def ntotal(self):
        return self.lib.qk_group_ntotal(self.h)

    def nlist(self):
        return self.lib.qk_group_nlist(self.h)

    def list_ids(self):
        n = C.c_int64()
        check(self.lib.qk_group_list_ids(self.h, None, C.byref(n)))
        out = np.empty(n.value, np.int64)
        if n.value:
            check(self.lib.qk_group_list_ids(self.h, _ptr(out), C.byref(n)))
        return out

    def get_list(self, list_no):
        n = self.list_size(list_no)
        vecs = np.empty((n, self.d), np.float32)
        ids = np.empty(n, np.int64)
        check(self.lib.qk_group_get_list(self.h, int(list_no), _ptr(vecs), _ptr(ids), QK_MEM_HOST))
        return vecs, ids

    def get_list_ids(self, list_no):
        n = self.list_size(list_no)
        ids = np.empty(n, np.int64)
        if n:
            check(self.lib.qk_group_get_list(self.h, int(list_no), None, _ptr(ids), QK_MEM_HOST))
        return ids

    def get_list_device(self, list_no):
        """(vectors [n, d] as a CUDA tensor on the LEAD's device, written there by the owner; ids [n] as a host array)"""
        import torch
        n = self.list_size(list_no)
        vecs = torch.empty((n, self.d), dtype=torch.float32, device=torch.device("cuda", self.device))
        ids = np.empty(n, np.int64)
        if n:
            check(self.lib.qk_group_get_list(self.h, int(list_no), _ptr(vecs), None, QK_MEM_DEVICE))
            check(self.lib.qk_group_get_list(self.h, int(list_no), None, _ptr(ids), QK_MEM_HOST))
        return vecs, ids

    def get_lists_device(self, list_nos):
        import torch
        nos = np.ascontiguousarray(list_nos, dtype=np.int64).reshape(-1)
        sizes = self.list_sizes(nos)
        vecs = torch.empty((int(sizes.sum()), self.d), dtype=torch.float32, device=torch.device("cuda", self.device))
        if vecs.shape[0]:
            check(self.lib.qk_group_get_lists(self.h, _ptr(nos), nos.shape[0], _ptr(vecs), None, QK_MEM_DEVICE))
        return vecs, sizes

    def get_vector(self, vid):
        out = np.empty(self.d, np.float32)
        found = C.c_int()
        check(self.lib.qk_group_get_vector(self.h, int(vid), _ptr(out), C.byref(found)))
        return out if found.value else None

    def refine_lists(self, list_nos, centroids, metric, refinement_iterations=0):
        list_nos = np.ascontiguousarray(list_nos, dtype=np.int64)
        c = _f32(centroids)
        c = c.clone() if _is_torch(c) else c.copy()
        check(self.lib.qk_group_refine_lists(self.h, _ptr(list_nos), list_nos.shape[0], _ptr(c), metric_code(metric),
                                             int(refinement_iterations), _mem_of(c)))
        return c

    def device_bytes(self):
        return self.lib.qk_group_device_bytes(self.h)

    def counters(self):
        """Store.counters() summed over the members"""
        tot = np.zeros(8, np.int64)
        for i in range(self.size()):
            out = np.zeros(8, np.int64)
            check(self.lib.qk_store_counters(self.member_handles(i)[1], _ptr(out), 8))
            tot += out
        return dict(zip(Store.COUNTER_NAMES, (int(v) for v in tot)))

    # ---- search over the members ------------------------------------------------------------------------------------------
    def search(self, parent, x, nprobe, k, metric, timing=False, out=None):
        """QueryCoordinator::search with workers.  parent: the parent's Store (replicated per member by the library)."""
        x = _f32(x)
        Q = x.shape[0]
        if out is None:
            out_i = _empty_like_mem((Q, k), np.int64, x)
            out_d = _empty_like_mem((Q, k), np.float32, x)
        else:
            out_i, out_d = out
        t = QkTiming()
        check(self.lib.qk_group_search(self.h, parent.h, _ptr(x), Q, int(nprobe), int(k), metric_code(metric), _ptr(out_i),
                                       _ptr(out_d), _mem_of(x), C.byref(t) if timing else None))
        return (out_i, out_d, timing_dict(t)) if timing else (out_i, out_d)

    def search_aps(self, parent, x, k, metric, recall_target, recompute_threshold=0.001, use_precomputed=True,
                   initial_search_fraction=0.02, timing=False):
        """recall-target search over the members (qk_group_search_aps).  Returns (ids, dist, nscanned[, timing])."""
        x = _f32(x)
        Q = x.shape[0]
        k = max(int(k), 1)
        out_i = _empty_like_mem((Q, k), np.int64, x)
        out_d = _empty_like_mem((Q, k), np.float32, x)
        out_n = _empty_like_mem((Q,), np.int32, x)
        t = QkTiming()
        check(self.lib.qk_group_search_aps(self.h, parent.h, _ptr(x), Q, k, metric_code(metric), float(recall_target),
                                           float(recompute_threshold), int(bool(use_precomputed)), float(initial_search_fraction),
                                           _ptr(out_i), _ptr(out_d), _ptr(out_n), _mem_of(x), C.byref(t) if timing else None))
        return (out_i, out_d, out_n, timing_dict(t)) if timing else (out_i, out_d, out_n)

    def scan(self, x, pids, k, metric, timing=False):
        """scan_partitions with workers (worker_scan): pids [Q, P] (or [P]: the same set for every query), -1 = skip"""
        x, pids = _f32(x), _i64(pids)
        Q = x.shape[0]
        if pids.ndim == 1:
            pids = (pids[None, :].expand(Q, -1).contiguous() if _is_torch(pids)
                    else np.ascontiguousarray(np.broadcast_to(pids[None, :], (Q, pids.shape[0]))))
        out_i = _empty_like_mem((Q, k), np.int64, x)
        out_d = _empty_like_mem((Q, k), np.float32, x)
        t = QkTiming()
        check(self.lib.qk_group_scan(self.h, _ptr(x), Q, _ptr(pids) if pids.shape[1] > 0 else None, int(pids.shape[1]), int(k),
                                     metric_code(metric), _ptr(out_i), _ptr(out_d), _mem_of(x, pids), C.byref(t) if timing else None))
        return (out_i, out_d, timing_dict(t)) if timing else (out_i, out_d)
