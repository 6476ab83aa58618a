"""Host-side mirror of the reference's public surface for the hot path, on top of the C ABI.

Names, attributes, argument meaning and error behaviour follow `quake._bindings`
(src/cpp/bindings/wrap.cpp:48-368) and the C++ classes behind it (src/cpp/include/quake_index.h:18-142,
src/cpp/include/common.h:104-247), so that the reference's tests can be restated one for one
(tests/test_index_gpu.py).  All arithmetic happens in libquake_hip.so; this file only marshals tensors and keeps
the bookkeeping the reference keeps in PartitionManager (resident-id set, next partition id).

Tensors: torch tensors in, torch tensors out.  CPU tensors behave like the reference (results are CPU tensors);
CUDA tensors stay on the device.
"""
import os
import struct
import time

import numpy as np
import torch

from . import capi
from ._lib import QuakeHipError

# defaults: src/cpp/include/common.h:66-99
DEFAULT_NLIST = 0
DEFAULT_NITER = 5
DEFAULT_METRIC = "l2"
DEFAULT_NUM_WORKERS = 0
DEFAULT_K = 1
DEFAULT_NPROBE = 1
DEFAULT_RECALL_TARGET = -1.0
DEFAULT_BATCHED_SCAN = False
DEFAULT_PRECOMPUTED = True
DEFAULT_INITIAL_SEARCH_FRACTION = 0.02
DEFAULT_RECOMPUTE_THRESHOLD = 0.001
DEFAULT_APS_FLUSH_PERIOD_US = 100
SERIALIZATION_MAGIC = 0x44494E4C  # common.h:66
SERIALIZATION_VERSION = 3        # common.h:67
INT32_MAX = 2 ** 31 - 1


def _js(v):
    """one value of the JSON-style summaries wrap.cpp's __repr__s print (C++ stream formatting: true / false, %g floats)"""
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, str):
        return '"%s"' % v
    if isinstance(v, float):
        return "%g" % v
    return str(v)


class _Summary:
    """__repr__ = the reference's one-line summary: _repr_fields names (summary key, attribute) pairs, _repr_trailing the
    ", }" two of the reference's summaries end in (wrap.cpp:122-128, 211-226)"""
    _repr_fields = ()
    _repr_trailing = False

    def _repr_pairs(self):
        return [(k, getattr(self, a)) for k, a in self._repr_fields]

    def __repr__(self):
        return "{" + ", ".join('"%s": %s' % (k, _js(v)) for k, v in self._repr_pairs()) + (", }" if self._repr_trailing else "}")


class MaintenancePolicyParams(_Summary):  # common.h:104-118, wrap.cpp:189-226
    _repr_fields = tuple((k, k) for k in ("maintenance_policy", "window_size", "refinement_radius", "refinement_iterations",
                                          "min_partition_size", "alpha", "enable_split_rejection", "enable_delete_rejection",
                                          "delete_threshold_ns", "split_threshold_ns"))
    _repr_trailing = True

    def __init__(self):
        self.maintenance_policy = "query_cost"
        self.window_size = 1000
        self.refinement_radius = 25
        self.refinement_iterations = 3
        self.min_partition_size = 32
        self.alpha = 0.9
        self.enable_split_rejection = True
        self.enable_delete_rejection = True
        self.delete_threshold_ns = 10.0
        self.split_threshold_ns = 10.0
        # EXTENSION (not a reference field; not in __repr__, which stays the reference's): a partition that entered the delete
        # branch and was KEPT by the rejection rule is then examined for a split like every other kept partition.  The reference
        # (maintenance_policies.cpp:68-131) never does: its delete model spreads a partition's vectors and hits evenly over all
        # others, which is strongly negative for exactly the partitions that are both larger and hotter than average -- they enter
        # the delete branch, the rejection (which looks at where the vectors would really go) keeps them, and the `else` with the
        # split test is never reached: the lists a skewed insert stream grows are the ones that are never split.  False = the
        # reference's decisions, bit for bit.
        self.split_after_delete_rejection = False


class IndexBuildParams(_Summary):  # common.h:123-143, wrap.cpp:131-150
    _repr_fields = tuple((k, k) for k in ("nlist", "niter", "metric", "num_workers"))

    def __init__(self):
        self.dimension = 0
        self.nlist = DEFAULT_NLIST
        self.num_workers = DEFAULT_NUM_WORKERS
        self.metric = DEFAULT_METRIC
        self.niter = DEFAULT_NITER
        self.use_gpu = True  # the reference's switch for GPU k-means (common.h:135); always on here
        self.seed = 1234     # faiss ClusteringParameters::seed default


class SearchParams(_Summary):  # common.h:171-184, wrap.cpp:153-186
    _repr_fields = tuple((k, k) for k in ("k", "nprobe", "recall_target", "batched_scan", "use_precomputed",
                                          "initial_search_fraction", "recompute_threshold", "aps_flush_period_us"))

    def __init__(self):
        self.nprobe = DEFAULT_NPROBE
        self.k = DEFAULT_K
        self.recall_target = DEFAULT_RECALL_TARGET
        self.num_threads = 1
        self.k_factor = 1.0
        self.use_precomputed = DEFAULT_PRECOMPUTED
        self.batched_scan = DEFAULT_BATCHED_SCAN
        self.recompute_threshold = DEFAULT_RECOMPUTE_THRESHOLD
        self.initial_search_fraction = DEFAULT_INITIAL_SEARCH_FRACTION
        self.aps_flush_period_us = DEFAULT_APS_FLUSH_PERIOD_US


class SearchTimingInfo(_Summary):  # common.h:214-228, wrap.cpp:279-320; device phases come from HIP events (qk_timing)
    def _repr_pairs(self):
        p = [(k, getattr(self, k)) for k in ("total_time_ns", "buffer_init_time_ns", "job_enqueue_time_ns",
                                             "boundary_distance_time_ns", "job_wait_time_ns", "result_aggregate_time_ns")]
        if self.parent_info is not None:
            p.append(("parent_scan_time_ns", self.parent_info.total_time_ns))
        return p + [(k, getattr(self, k)) for k in ("n_queries", "n_clusters", "partitions_scanned")]

    def __init__(self):
        self.n_queries = 0
        self.n_clusters = 0
        self.partitions_scanned = 0
        self.search_params = None
        self.parent_info = None
        self.buffer_init_time_ns = 0
        self.job_enqueue_time_ns = 0
        self.boundary_distance_time_ns = 0
        self.job_wait_time_ns = 0
        self.result_aggregate_time_ns = 0
        self.total_time_ns = 0


class BuildTimingInfo(_Summary):  # common.h:189-198, wrap.cpp:323-350
    _repr_fields = (("total_time_us", "total_time_us"), ("assign_time_us", "assign_time_us"), ("train_time_us", "train_time_us"),
                    ("d", "d"), ("code_size", "code_size"), ("n_codebooks", "num_codebooks"), ("n_vectors", "n_vectors"))

    @property
    def n_codebooks(self):  # the bound name of num_codebooks (wrap.cpp:334)
        return self.num_codebooks

    def __init__(self):
        self.n_vectors = 0
        self.n_clusters = 0
        self.d = 0
        self.num_codebooks = -1
        self.code_size = -1
        self.train_time_us = 0
        self.assign_time_us = 0
        self.total_time_us = 0


class ModifyTimingInfo(_Summary):  # common.h:203-209; wrap.cpp:264 aliases modify_count to n_vectors
    _repr_fields = (("modify_count", "n_vectors"), ("input_validation_time_us", "input_validation_time_us"),
                    ("modify_time_us", "modify_time_us"), ("find_partition_time_us", "find_partition_time_us"))

    def __init__(self):
        self.n_vectors = 0
        self.input_validation_time_us = 0
        self.find_partition_time_us = 0
        self.modify_time_us = 0
        self.maintenance_time_us = 0

    @property
    def modify_count(self):
        return self.n_vectors


class MaintenanceTimingInfo(_Summary):  # common.h:233-241, wrap.cpp:244-256
    _repr_fields = tuple((k, k) for k in ("total_time_us", "split_time_us", "delete_time_us", "split_refine_time_us",
                                          "delete_refine_time_us", "n_splits", "n_deletes"))

    def __init__(self):
        self.n_splits = 0
        self.n_deletes = 0
        self.delete_time_us = 0
        self.delete_refine_time_us = 0
        self.split_time_us = 0
        self.split_refine_time_us = 0
        self.total_time_us = 0


class SearchResult:  # common.h:243-247
    def __init__(self):
        self.ids = None
        self.distances = None
        self.timing_info = None


_CONTEXTS = {}


class _ResidentIds:
    """PartitionManager::resident_ids_ (partition_manager.h): which vector ids are in the index.  A byte map indexed by id
    (ids are < INT_MAX, partition_manager.cpp:150-153) so that batches are checked with numpy instead of per-id lookups;
    ids beyond 2^28 (a 256 MB map) switch to a plain set."""
    LIMIT = 1 << 28

    def __init__(self):
        self._map = np.zeros(0, np.bool_)
        self._set = None

    def _to_set(self):
        self._set = set(np.nonzero(self._map)[0].tolist())
        self._map = np.zeros(0, np.bool_)

    def any_present(self, ids):
        if self._set is not None:
            return any(v in self._set for v in ids.tolist())
        ids = ids[(ids >= 0) & (ids < self._map.shape[0])]
        return bool(self._map[ids].any()) if ids.size else False

    def all_present(self, ids):
        if ids.size == 0:
            return True
        if self._set is not None:
            return all(v in self._set for v in ids.tolist())
        if int(ids.max()) >= self._map.shape[0] or int(ids.min()) < 0:
            return False
        return bool(self._map[ids].all())

    def update(self, ids):
        ids = np.asarray(list(ids) if not isinstance(ids, np.ndarray) else ids, dtype=np.int64).reshape(-1)
        if not ids.size:
            return
        if self._set is None and int(ids.max()) >= self.LIMIT:
            self._to_set()
        if self._set is not None:
            self._set.update(ids.tolist())
            return
        m = int(ids.max())
        if m >= self._map.shape[0]:
            new = np.zeros(min(self.LIMIT, max(m + 1, 2 * self._map.shape[0], 1024)), np.bool_)
            new[:self._map.shape[0]] = self._map
            self._map = new
        self._map[ids] = True

    def discard_all(self, ids):
        if self._set is not None:
            self._set.difference_update(ids.tolist())
        else:
            self._map[ids] = False

    def discard_present(self, ids):
        """forget the ids that are here, ignore the others (a shard is handed the whole remove batch)"""
        if self._set is not None:
            self._set.difference_update(ids.tolist())
        else:
            ids = ids[(ids >= 0) & (ids < self._map.shape[0])]
            self._map[ids] = False

    def __contains__(self, v):
        if self._set is not None:
            return v in self._set
        return 0 <= v < self._map.shape[0] and bool(self._map[v])

    def __len__(self):
        return len(self._set) if self._set is not None else int(self._map.sum())


def _context(device=0):
    """the per-device context, bound to torch's CURRENT stream of that device: device tensors handed in were produced on
    it and the tensors handed back are consumed on it, so the library's work has to be ordered with it"""
    if device not in _CONTEXTS:
        _CONTEXTS[device] = capi.Context(device)
    ctx = _CONTEXTS[device]
    ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
    return ctx


_SPLIT_WORKERS = {}  # device -> (contexts with a private stream each, thread pool)


def _split_workers(device, n):
    """`n` contexts of their own (private streams, own k-means scratch) and a thread pool over them: the 2-means of a maintenance
    call's splits are independent problems of ~0.5 ms of launch latency each (5 Lloyd iterations on a 512-row sample, a
    synchronisation per iteration), hundreds of them when the window first fills -- several in flight at once, the same library call
    and the same bits each"""
    import concurrent.futures
    ent = _SPLIT_WORKERS.get(device)
    if ent is None or len(ent[0]) < n:
        ctxs = list(ent[0]) if ent else []
        while len(ctxs) < n:
            ctxs.append(capi.Context(device))
        if ent:
            ent[1].shutdown(wait=True)
        ent = (ctxs, concurrent.futures.ThreadPoolExecutor(max_workers=n, thread_name_prefix="quake-split"))
        _SPLIT_WORKERS[device] = ent
    return ent


def _us(t0):
    return int((time.perf_counter() - t0) * 1e6)


class QuakeIndex:
    """quake_index.h:18-142.  One level of the (recursive) index; `parent` is a flat QuakeIndex over the centroids."""

    def __init__(self, current_level=0, device=0):
        self.parent = None
        self.current_level = int(current_level)
        self.metric_ = None
        self.build_params_ = None
        self.maintenance_policy_params_ = None
        self.maintenance_policy_ = None
        self._policy_cost_estimator = None
        self.track_hits = False  # record the partitions each query scans (maintenance_policies.cpp:179-182)
        self.debug_ = False
        self._device = int(device)
        self._has_ctx = False
        self._store = None
        self._resident = _ResidentIds()   # PartitionManager::resident_ids_
        self._next_pid = 0       # PartitionManager::curr_partition_id_
        self._d = 0
        # [Q, nprobe] list numbers of tracked searches not yet handed to the policy (device tensors): a search with hit tracking on
        # does not synchronise the device or touch the host arrays of the tracker; the lists cross in ONE transfer before anything
        # that changes a list's size or reads the policy (add / remove / refine / maintenance / a new policy) -- so every hit is
        # credited with the size its list had when it was scanned, as if it had been recorded inside search
        self._pending_hits = []
        # counts everything that changes a list's contents or a centroid (add / remove / refine / split / delete): what the policy
        # derived from the lists (where a delete candidate's vectors would go) stays valid while it does not move
        self._mutations = 0
        self._reassign_cache = {}

    # -- helpers -------------------------------------------------------------------------------------------------
    @property
    def parent_(self):
        return self.parent

    @property
    def _ctx(self):
        return _context(self._device) if self._has_ctx else None

    def _to_dev(self, t, dtype):
        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        return t.to(device=torch.device("cuda", self._device), dtype=dtype).contiguous()

    def _require_built(self, who):
        if self._store is None:
            raise RuntimeError(who)

    def _new_lists(self, d, num_workers, partitioned):
        """the device lists of this level: ONE store, or -- num_workers > 0 and there are partitions to distribute (a flat index is
        one partition) -- a device group of num_workers members, member j on GPU j % #GPUs, partition p in member p % num_workers:
        the reference's partition -> worker map (PartitionManager::distribute_partitions, partition_manager.cpp:557-603) with a
        GPU as the worker.  Both answer to the same method names (capi.Store / capi.Group)."""
        nw = int(num_workers or 0)
        if nw > 0 and partitioned:
            ndev = max(1, torch.cuda.device_count())
            return capi.Group([(self._device + j) % ndev for j in range(nw)], d)
        return capi.Store(self._ctx, d)

    @property
    def num_workers_(self):
        """members of the device group the partitions are distributed over (0: one store)"""
        return self._store.size() if isinstance(self._store, capi.Group) else 0

    # -- build (quake_index.cpp:29-90) -------------------------------------------------------------------------------
    def build(self, x, ids, build_params):
        t_total = time.perf_counter()
        self.build_params_ = build_params
        self.metric_ = capi.metric_code(build_params.metric)  # raises ValueError("Invalid metric type: ...")
        if x.dim() != 2:
            raise RuntimeError("[QuakeIndex::build] x must be 2-D [num_vectors, dimension]")
        if x.shape[0] != ids.shape[0]:
            raise RuntimeError("[QuakeIndex::build] x.size(0) != ids.size(0)")
        self._has_ctx = True
        n, d = int(x.shape[0]), int(x.shape[1])
        self._d = d
        info = BuildTimingInfo()
        info.n_vectors, info.d = n, d
        xd = self._to_dev(x, torch.float32)
        idd = self._to_dev(ids, torch.int64)
        nlist = int(build_params.nlist)
        self._store = self._new_lists(d, getattr(build_params, "num_workers", 0), nlist > 1)
        if nlist > 1:
            t0 = time.perf_counter()
            # kmeans(): clustering.cpp:13-97 (IP: the normalised copy is what gets stored, :25-26,71)
            centroids, assign, xd = self._ctx.kmeans(xd, nlist, self.metric_, niter=int(build_params.niter),
                                                     seed=int(getattr(build_params, "seed", 1234)))
            info.train_time_us = _us(t0)
            t0 = time.perf_counter()
            order = torch.argsort(assign, stable=True)  # torch::sort + index_select, clustering.cpp:69-72
            counts = torch.bincount(assign, minlength=nlist).cpu().numpy().astype(np.int64)
            offsets = np.zeros(nlist + 1, np.int64)
            offsets[1:] = np.cumsum(counts)
            self._store.build_csr(offsets, idd[order].contiguous(), xd[order].contiguous())
            self.parent = QuakeIndex(self.current_level + 1, self._device)
            pparams = IndexBuildParams()
            pparams.metric = build_params.metric
            pparams.num_workers = build_params.num_workers
            self.parent.build(centroids, torch.arange(nlist, dtype=torch.int64), pparams)
            info.assign_time_us = _us(t0)
            info.n_clusters = nlist
            self._next_pid = nlist
        else:
            # flat index: one partition holding x (quake_index.cpp:68-79)
            self._store.build_csr(np.array([0, n], np.int64), idd, xd)
            self.parent = None
            info.n_clusters = 1
            self._next_pid = 1
        self._resident = _ResidentIds()
        self._resident.update(ids.reshape(-1).cpu().numpy().astype(np.int64))
        self.initialize_maintenance_policy(MaintenancePolicyParams())
        info.total_time_us = _us(t_total)
        return info

    @classmethod
    def from_partitions(cls, centroids, offsets, ids, vecs, metric, device=0):
        """An index over a clustering made elsewhere (a cross-shard k-means, a file): list p = rows [offsets[p], offsets[p+1])
        of ids / vecs, centroid p = centroids[p].  The state after build(): parent over the centroids, partition ids
        0..nlist-1, default maintenance policy (quake_index.cpp:29-90 minus the clustering)."""
        self = cls(0, device)
        bp = IndexBuildParams()
        bp.metric = metric
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        bp.nlist = int(offsets.shape[0] - 1)
        self.build_params_ = bp
        self.metric_ = capi.metric_code(metric)
        self._has_ctx = True
        cd = self._to_dev(centroids, torch.float32)
        self._d = int(cd.shape[1])
        if bp.nlist != cd.shape[0]:
            raise RuntimeError("[QuakeIndex::from_partitions] one centroid per list is required")
        self._store = capi.Store(self._ctx, self._d)
        idd = self._to_dev(ids, torch.int64)
        self._store.build_csr(offsets, idd, self._to_dev(vecs, torch.float32))
        self.parent = cls(1, device)
        pp = IndexBuildParams()
        pp.metric = metric
        self.parent.build(cd, torch.arange(bp.nlist, dtype=torch.int64), pp)
        self._next_pid = bp.nlist
        self._resident = _ResidentIds()
        self._resident.update(idd.cpu().numpy().astype(np.int64))
        self.initialize_maintenance_policy(MaintenancePolicyParams())
        return self

    # -- search (quake_index.cpp:93-99 -> query_coordinator.cpp:612-657) --------------------------------------------------
    def search(self, x, search_params):
        self._require_built("[QuakeIndex::search()] No query coordinator. Did you build the index?")
        res = SearchResult()
        ti = SearchTimingInfo()
        ti.search_params = search_params
        ti.n_clusters = self.nlist()
        res.timing_info = ti
        if x is None or x.shape[0] == 0:  # query_coordinator.cpp:476-482
            res.ids = torch.empty((0,), dtype=torch.int64)
            res.distances = torch.empty((0,), dtype=torch.float32)
            return res
        t0 = time.perf_counter()
        on_dev = x.is_cuda
        xd = self._to_dev(x, torch.float32)
        k = search_params.k if search_params.k and search_params.k > 0 else 1  # query_coordinator.cpp:490
        use_aps = (search_params.recall_target is not None and search_params.recall_target > 0.0 and self.parent is not None
                   and not search_params.batched_scan)  # query_coordinator.cpp:502,637-641,659-673
        grp = self._store if isinstance(self._store, capi.Group) else None
        if grp is not None:
            grp.set_stream(torch.cuda.current_stream(grp.device).cuda_stream)  # the lead's stream (see _context)
        if use_aps:
            # adaptive partition scanning: candidates = nlist * initial_search_fraction, per-query early stop (with workers the
            # rounds run on the group's lead and every member scans the pairs whose partitions it holds)
            aps_args = dict(recompute_threshold=float(search_params.recompute_threshold),
                            use_precomputed=bool(search_params.use_precomputed),
                            initial_search_fraction=float(search_params.initial_search_fraction), timing=True)
            if grp is not None:
                ids, dist, nscan, tm = grp.search_aps(self.parent._store, xd, int(k), self.metric_,
                                                      float(search_params.recall_target), **aps_args)
            else:
                ids, dist, nscan, tm = self._ctx.search_aps(self.parent._store, self._store, xd, int(k), self.metric_,
                                                            float(search_params.recall_target), **aps_args)
            ti.n_queries = int(x.shape[0])
            ti.partitions_scanned = int(nscan.sum().item())
            ti.job_wait_time_ns = int(tm["total_ms"] * 1e6)
            pi = SearchTimingInfo()
            pi.n_queries = ti.n_queries
            pi.n_clusters = 1
            ti.parent_info = pi
            ti.total_time_ns = int((time.perf_counter() - t0) * 1e9)
            res.ids = ids if on_dev else ids.cpu()
            res.distances = dist if on_dev else dist.cpu()
            return res
        nprobe = max(int(search_params.nprobe), 1)
        self._ctx.set_timing(1)
        try:
            if self.track_hits and self.parent is not None:
                # hit tracking for maintenance(): the probed partitions are needed by the policy (later, on the host)
                if grp is not None:
                    pids, _ = self._ctx.coarse(self.parent._store, xd, nprobe, self.metric_, values=False)
                    ids, dist, tm = grp.scan(xd, pids, int(k), self.metric_, timing=True)
                else:  # one enqueue: the nearest-centroid step writes the list numbers where the policy will read them
                    ids, dist, pids, tm = self._ctx.search_tracked(self.parent._store, self._store, xd, nprobe, int(k), self.metric_,
                                                                   timing=True)
                self._pending_hits.append(pids)
                if len(self._pending_hits) >= 64:
                    self._flush_hits()
            elif grp is not None:  # workers: every member scans the partitions it holds, the lead merges (worker_scan)
                ids, dist, tm = grp.search(self.parent._store, xd, nprobe, int(k), self.metric_, timing=True)
            else:
                ids, dist, tm = self._ctx.search(self.parent._store if self.parent is not None else None, self._store, xd,
                                                 nprobe, int(k), self.metric_, timing=True)
        finally:
            self._ctx.set_timing(0)
        ti.n_queries = int(x.shape[0])
        ti.partitions_scanned = int(tm["n_items"])
        ti.job_wait_time_ns = int(tm["scan_ms"] * 1e6)
        ti.result_aggregate_time_ns = int(tm["merge_ms"] * 1e6)
        ti.job_enqueue_time_ns = int(tm["group_ms"] * 1e6)
        if self.parent is not None:
            pi = SearchTimingInfo()
            pi.n_queries = ti.n_queries
            pi.n_clusters = 1
            pi.total_time_ns = int(tm["coarse_ms"] * 1e6)
            ti.parent_info = pi
        ti.total_time_ns = int((time.perf_counter() - t0) * 1e9)
        res.ids = ids if on_dev else ids.cpu()
        res.distances = dist if on_dev else dist.cpu()
        return res

    # -- get / get_ids (partition_manager.cpp:322-343) --------------------------------------------------------------------
    def get(self, ids):
        self._require_built("[QuakeIndex::get()] No partition manager. Index not built?")
        if hasattr(self._store, "get_vectors") and ids.shape[0] > 1:  # one call, one gather (qk_store_get_vectors)
            vecs, found = self._store.get_vectors(ids.reshape(-1).cpu().numpy())
            if not found.all():
                raise RuntimeError("ID not found in any partition")
            return torch.from_numpy(vecs)
        out = torch.empty((ids.shape[0], self._d), dtype=torch.float32)
        for i, v in enumerate(ids.reshape(-1).tolist()):
            vec = self._store.get_vector(int(v))
            if vec is None:
                raise RuntimeError("ID not found in any partition")
            out[i] = torch.from_numpy(vec)
        return out

    def get_ids(self):
        self._require_built("[QuakeIndex::get_ids()] No partition manager. Index not built?")
        # (ids only, from the store's host mirror: get_list would extract every vector of the index on the way)
        parts = [torch.from_numpy(self._store.get_list_ids(int(p))) for p in self._store.list_ids()]
        return torch.cat(parts) if parts else torch.empty((0,), dtype=torch.int64)

    # -- add (partition_manager.cpp:123-262) ---------------------------------------------------------------------------------
    def add(self, x, ids):
        self._require_built("[QuakeIndex::add()] No partition manager. Build the index first.")
        self._flush_hits()
        info = ModifyTimingInfo()
        t0 = time.perf_counter()
        if x.shape[0] != ids.shape[0]:
            raise RuntimeError("[PartitionManager] add: mismatch in vectors.size(0) and vector_ids.size(0).")
        n = int(x.shape[0])
        info.n_vectors = n
        if n == 0:
            return info
        if x.dim() != 2:
            raise RuntimeError("[PartitionManager] add: 'vectors' must be 2D [N, dim].")
        idn = ids.reshape(-1).cpu().numpy().astype(np.int64)
        if int(idn.max()) > INT32_MAX:
            raise RuntimeError("[PartitionManager] add: vector_ids must be less than INT_MAX.")
        if int(idn.min()) < 0:
            raise RuntimeError("[PartitionManager] add: vector_ids must be non-negative (-1 marks an empty result slot).")
        # (uniqueness of a large batch is checked by a device sort: np.unique of 1M ids was 60 of the 87 ms this validation took)
        if n >= 65536:
            n_unique = int(torch.unique(self._to_dev(ids.reshape(-1), torch.int64)).shape[0])
        else:
            n_unique = int(np.unique(idn).shape[0])
        if n_unique != n:
            raise RuntimeError("[PartitionManager] add: vector_ids must be unique.")
        if self._resident.any_present(idn):
            raise RuntimeError("[PartitionManager] init_partitions: vector ID already exists in the index.")
        info.input_validation_time_us = _us(t0)
        t0 = time.perf_counter()
        xd = self._to_dev(x, torch.float32)
        idd = self._to_dev(ids.reshape(-1), torch.int64)
        if self.parent is None:
            assign = torch.zeros(n, dtype=torch.int64, device=xd.device)
        else:
            # parent_->search(x, {k = 1, nprobe = parent nlist}) (:219-230) == coarse step with nprobe 1
            pids, _ = self._ctx.coarse(self.parent._store, xd, 1, self.metric_, values=False)
            assign = pids.reshape(-1)
        info.find_partition_time_us = _us(t0)
        t0 = time.perf_counter()
        self._store.add_batch(idd, xd, assign.contiguous())  # per-list append order = input order (:245-258)
        self._mutations += 1
        self._resident.update(idn)  # only once the device step succeeded: a failed add leaves no phantom ids behind
        self._publish()
        info.modify_time_us = _us(t0)
        return info

    # -- remove (partition_manager.cpp:264-320) -------------------------------------------------------------------------------
    def remove(self, ids):
        self._require_built("[QuakeIndex::remove()] No partition manager. Build the index first.")
        self._flush_hits()
        info = ModifyTimingInfo()
        info.n_vectors = int(ids.shape[0])
        if ids.shape[0] == 0:
            return info
        t0 = time.perf_counter()
        idn = ids.reshape(-1).cpu().numpy().astype(np.int64)
        if not self._resident.all_present(idn):
            raise RuntimeError("[PartitionManager] remove: vector ID does not exist in the index.")
        self._resident.discard_all(idn)
        info.input_validation_time_us = _us(t0)
        t0 = time.perf_counter()
        self._store.remove_ids(idn)
        self._mutations += 1
        self._publish()
        info.modify_time_us = _us(t0)
        return info

    def modify(self, ids, x):  # quake_index.cpp:147-150
        self.remove(ids)
        return self.add(x, ids)

    # -- local refinement (partition_manager.cpp:446-487 -> kmeans_refine_partitions, clustering.cpp:99-182) ----------------
    def refine_partitions(self, partition_ids=None, iterations=0):
        """PartitionManager::refine_partitions: re-assign the vectors of the given partitions among their centroids on the
        GPU (assign = MFMA kernel, update = ordered sums), replace the partitions, write the centroids back to the parent
        with parent_->modify (:478)."""
        self._require_built("[PartitionManager] refine_partitions: index not built")
        self._flush_hits()
        if self.parent is None:
            return
        if partition_ids is None:
            partition_ids = self.parent.get_ids()
        if partition_ids.shape[0] == 0:
            return
        self._mutations += 1
        cent = self.parent.get(partition_ids)
        new_c = self._store.refine_lists(partition_ids.reshape(-1).cpu().numpy(), cent.numpy(), self.metric_, int(iterations))
        self.parent.modify(partition_ids, torch.from_numpy(np.ascontiguousarray(new_c)))

    # -- maintenance (quake_index.cpp:152-163; maintenance_policies.cpp; partition_manager.cpp:344-554) ---------------------
    def initialize_maintenance_policy(self, maintenance_policy_params, cost_estimator=None):
        """quake_index.cpp:165-168.  The policy object (cost model included) is created lazily: profiling the device scan
        for the latency grid costs a few hundred launches, which a search-only user never needs."""
        if self.maintenance_policy_ is not None:
            self._flush_hits()  # hits recorded under the old policy reach it before it goes
        self._pending_hits = []
        self.maintenance_policy_params_ = maintenance_policy_params
        self.maintenance_policy_ = None
        self._policy_cost_estimator = cost_estimator

    def _policy(self):
        if self._pending_hits:  # whoever reads the policy sees every hit recorded so far
            self._flush_hits()
        if self.maintenance_policy_ is None:
            from .maintenance import MaintenancePolicy
            self.maintenance_policy_ = MaintenancePolicy(self, self.maintenance_policy_params_, self._policy_cost_estimator)
        return self.maintenance_policy_

    def _flush_hits(self):
        pend, self._pending_hits = self._pending_hits, []
        if not pend:
            return
        if len({tuple(p.shape[1:]) for p in pend}) == 1:
            self.record_query_hits(torch.cat(pend, 0).cpu().numpy())  # one transfer, one pass over the window
        else:
            for p in pend:
                self.record_query_hits(p.cpu().numpy())

    def record_query_hits(self, pids):
        """pids [Q, nprobe] (host): partitions every query scanned.  maintenance_policies.cpp:179-182."""
        pol = self._policy()
        arr = np.asarray(pids).reshape(len(pids), -1)
        uniq = np.unique(arr[arr >= 0])
        lut = np.zeros(int(uniq.max()) + 1 if uniq.size else 1, np.int64)
        lut[uniq] = self._partition_sizes(uniq.tolist())
        pol.hit_count_tracker_.add_batch(arr, lut[np.clip(arr, 0, None)])

    def maintenance(self):
        """MaintenancePolicy::perform_maintenance (maintenance_policies.cpp:33-177): nothing until window_size queries have
        been recorded (:36-41), then delete / split / local refinement as the cost model decides."""
        if self.maintenance_policy_params_ is None:
            raise RuntimeError("[QuakeIndex::maintenance()] No maintenance policy set.")
        if self.parent is None:
            return MaintenanceTimingInfo()
        self._flush_hits()
        info = self._policy().perform_maintenance()
        self._publish()
        return info

    def _publish(self):
        """what a modification left for the next search to do (list table upload, the parent's row-major copy) is done now: the
        queries after an add / remove / maintenance do not pay for it"""
        for st in (self._store, self.parent._store if self.parent is not None else None):
            if st is not None and hasattr(st, "publish"):
                st.publish()

    def _partition_sizes(self, pids):
        return self._store.list_sizes(np.asarray([int(p) for p in pids], np.int64)).tolist()  # one call (qk_store_list_sizes)

    def _list_ids(self):
        return [int(p) for p in self._store.list_ids()]

    def _reassign_targets(self, pid):
        """where would the vectors of partition `pid` go if it were deleted: the other partitions among every vector's two
        nearest centroids and how many vectors name each (maintenance_policies.cpp:79-101).  -> (pids, counts) lists."""
        vecs, _ = self._store.get_list(int(pid))
        near, _ = self._ctx.coarse(self.parent._store, torch.from_numpy(vecs).cuda(self._device), 2, self.metric_, values=False)
        flat = near.reshape(-1)
        flat = flat[(flat != int(pid)) & (flat >= 0)]
        uniq, counts = torch.unique(flat, return_counts=True)
        return [int(v) for v in uniq.tolist()], [int(v) for v in counts.tolist()]

    def _reassign_targets_many(self, pids, chunk_rows=1 << 18):
        """_reassign_targets for many partitions with a handful of device calls instead of two host round trips per partition (a
        50M index has hundreds of delete candidates per maintenance call): the lists are extracted on the device, one nearest-two
        search per chunk of ~2^18 rows, one unique over (partition, target) keys.  -> {pid: (pids, counts)}."""
        pids = [int(p) for p in pids]
        # (the policy asks for the SAME candidates call after call -- the partitions its delete model dislikes are the large, hot ones,
        #  and they stay that -- : answers are kept while nothing changed a list or a centroid; 16 ms per call at 10M otherwise)
        if self._reassign_cache.get("epoch") != self._mutations:
            self._reassign_cache = {"epoch": self._mutations, "targets": {}}
        known = self._reassign_cache["targets"]
        out = {p: known[p] for p in pids if p in known}
        pids = [p for p in pids if p not in known]
        for p in pids:
            out[p] = ([], [])
        if not pids:
            return out
        all_ids = [int(v) for v in self._list_ids()]
        npart = max(all_ids) + 2 if all_ids else 2
        keys = []
        # the candidates' rows in ONE device buffer per ~2^23 rows (qk_store_get_lists: no per-list Python, no per-list tensor),
        # the nearest-two search in chunks of rows, (candidate, target) keys counted by one unique
        sizes_all = np.asarray(self._partition_sizes(pids), np.int64)
        group_rows = 1 << 23
        g0 = 0
        while g0 < len(pids):
            g1, rows = g0, 0
            while g1 < len(pids) and (rows == 0 or rows + sizes_all[g1] <= group_rows):
                rows += int(sizes_all[g1])
                g1 += 1
            if rows > 0:
                x, sz = self._store.get_lists_device(pids[g0:g1])
                dev = x.device
                seg = torch.repeat_interleave(torch.arange(g0, g1, dtype=torch.int64, device=dev), torch.as_tensor(sz, device=dev))
                own = torch.as_tensor(pids[g0:g1], dtype=torch.int64, device=dev)[seg - g0]
                for c0 in range(0, rows, chunk_rows):
                    c1 = min(rows, c0 + chunk_rows)
                    near, _ = self._ctx.coarse(self.parent._store, x[c0:c1], 2, self.metric_, values=False)
                    ok = (near >= 0) & (near != own[c0:c1, None])
                    keys.append((seg[c0:c1, None] * npart + near)[ok])
            g0 = g1
        if keys:
            uniq, counts = torch.unique(torch.cat(keys), return_counts=True)
            uniq, counts = uniq.cpu().numpy(), counts.cpu().numpy()
            segs, tgt = uniq // npart, uniq % npart
            cut = np.nonzero(np.diff(segs))[0] + 1  # (uniq is sorted: a candidate's targets are one run)
            for lo, hi in zip(np.concatenate([[0], cut]).tolist(), np.concatenate([cut, [len(segs)]]).tolist()):
                out[pids[int(segs[lo])]] = (tgt[lo:hi].tolist(), counts[lo:hi].tolist())
        for p in pids:
            known[p] = out[p]
        return out

    def _neighbour_partitions(self, pids, radius):
        """the partitions whose centroids are among the `radius` nearest of each given partition's centroid, sorted
        (maintenance_policies.cpp:187-202)."""
        cent = self.parent.get(torch.tensor([int(p) for p in pids], dtype=torch.int64))
        near, _ = self._ctx.coarse(self.parent._store, cent.cuda(self._device), int(radius), self.metric_, values=False)
        out = torch.unique(near.reshape(-1))
        return [int(v) for v in out[out != -1].tolist()]

    def _select_partitions(self, pids):  # partition_manager.cpp:344-390
        """(vectors, ids) of the given partitions: vectors as CUDA tensors extracted on the device (no host hop: a maintenance call
        that splits 200 partitions of a 10M index moved 0.5 GB through the host here), ids from the store's host mirror"""
        vecs, ids = [], []
        for p in pids:
            v, i = self._store.get_list_device(int(p))
            vecs.append(v)
            ids.append(i)
        return vecs, ids

    def _split_partitions(self, pids):  # :392-444: 2-means of every partition (qk_kmeans on the GPU)
        vecs, ids = self._select_partitions(pids)
        out_c, out_v, out_i = [], [], []
        for xd, i in zip(vecs, ids):
            assert xd.shape[0] >= 4, "Partition must have at least 8 vectors to split."  # (the reference's message, :412)
            cent, assign, xd = self._ctx.kmeans(xd, 2, self.metric_, niter=5, seed=1234)
            left = assign == 0
            lh = left.cpu().numpy()  # (n bytes: the one transfer of a split)
            cent = cent.cpu().numpy()
            for j, (md, mh) in enumerate(((left, lh), (~left, ~lh))):
                out_c.append(cent[j])
                out_v.append(xd[md])                       # stays on the device: add_entries ingests it from there
                out_i.append(np.ascontiguousarray(i[mh]))
        return {"centroids": np.stack(out_c), "vectors": out_v, "vector_ids": out_i}

    def _split_partitions_in_place(self, pids):
        """_split_partitions + _delete_partitions(.., reassign=False) + _add_partitions in one step with the rows where they are: the
        lists in ONE device buffer, the same 2-means on each slice of it, one ingest with every row's new list number -- no boolean
        gather and no host synchronisation per partition (a split was 1.3 ms; ~65 of them per maintenance call of a 50M index).
        Same partitions, row order and centroid bits as the three calls.  None: not applicable (a device group), take the three calls."""
        st = self._store
        if not isinstance(st, capi.Store) or not pids:
            return None
        pids = [int(p) for p in pids]
        x, sz, idd = st.get_lists_device(pids, with_ids=True)
        assert int(sz.min()) >= 4, "Partition must have at least 8 vectors to split."  # (the reference's message, :412)
        n = len(pids)
        dev = x.device
        cents = torch.empty((2 * n, self._d), dtype=torch.float32, device=dev)
        assign = torch.empty((x.shape[0],), dtype=torch.int64, device=dev)
        starts = np.concatenate([[0], np.cumsum(np.asarray(sz, dtype=np.int64))])
        nw = min(int(os.environ.get("QUAKE_SPLIT_THREADS", "8")), n)
        if nw >= 2:
            ctxs, pool = _split_workers(dev.index or 0, nw)
            self._ctx.synchronize()  # the rows were gathered on this context's stream; the workers run on streams of their own

            def run(w):
                for i in range(w, n, nw):
                    a, b = int(starts[i]), int(starts[i + 1])
                    ctxs[w].kmeans_inplace(x[a:b], 2, self.metric_, cents[2 * i:2 * i + 2], assign[a:b], niter=5, seed=1234)
                ctxs[w].synchronize()

            for f in [pool.submit(run, w) for w in range(nw)]:
                f.result()
        else:
            for i in range(n):
                a, b = int(starts[i]), int(starts[i + 1])
                self._ctx.kmeans_inplace(x[a:b], 2, self.metric_, cents[2 * i:2 * i + 2], assign[a:b], niter=5, seed=1234)
        new_pids = list(range(self._next_pid, self._next_pid + 2 * n))
        self._next_pid += 2 * n
        first = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=dev) * 2 + new_pids[0], torch.as_tensor(sz, device=dev))
        listno = first + (assign != 0).to(torch.int64)
        self.parent.remove(torch.tensor(pids, dtype=torch.int64))
        for p in pids:
            st.remove_list(p)
        for p in new_pids:
            st.add_list(p)
        st.add_batch(idd, x, listno)
        self.parent.add(cents.cpu(), torch.tensor(new_pids, dtype=torch.int64))
        self._mutations += 1
        return new_pids

    def _add_partitions(self, clustering):  # :489-520
        n = len(clustering["vectors"])
        new_pids = list(range(self._next_pid, self._next_pid + n))
        self._next_pid += n
        vs, ii = clustering["vectors"], clustering["vector_ids"]
        for pid in new_pids:
            self._store.add_list(pid)
        if n and all(torch.is_tensor(v) and v.is_cuda for v in vs):
            # device rows (the halves of split partitions): ONE ingest for all the new partitions -- rows in partition order, so a
            # partition's append order is its own row order, exactly what one add_entries per partition leaves behind (each of those was
            # a launch, a copy of the ids and a synchronisation: 0.27 ms x 2 per split)
            keep = [j for j in range(n) if vs[j].shape[0]]
            if keep:
                dev = vs[keep[0]].device
                x = torch.cat([vs[j] for j in keep], 0)
                idd = torch.from_numpy(np.concatenate([np.ascontiguousarray(ii[j], dtype=np.int64) for j in keep])).to(dev)
                assign = torch.repeat_interleave(torch.tensor([new_pids[j] for j in keep], dtype=torch.int64, device=dev),
                                                 torch.tensor([vs[j].shape[0] for j in keep], dtype=torch.int64, device=dev))
                self._store.add_batch(idd, x, assign)
        else:
            for pid, v, i in zip(new_pids, vs, ii):
                if v.shape[0]:
                    self._store.add_entries(pid, i, v)
        self.parent.add(torch.from_numpy(np.ascontiguousarray(clustering["centroids"])),
                        torch.tensor(new_pids, dtype=torch.int64))
        self._mutations += 1
        return new_pids

    def _delete_partitions(self, pids, reassign=True):  # :522-554
        if self.parent is None:
            raise RuntimeError("Index is not partitioned")
        vecs, ids = self._select_partitions(pids) if reassign else ([], [])
        self.parent.remove(torch.tensor([int(p) for p in pids], dtype=torch.int64))
        for p in pids:
            self._store.remove_list(int(p))
        self._mutations += 1
        if reassign:
            # PartitionManager::add(vectors, ids, {}, check_uniques = false): nearest remaining centroid -- of ALL the deleted
            # partitions' vectors at once (one coarse step, one add_batch: per-list append order = input order, the order of the
            # one-by-one loop)
            keep = [(v, i) for v, i in zip(vecs, ids) if v.shape[0]]
            if keep:
                xd = torch.cat([v for v, _ in keep], 0)
                idd = torch.from_numpy(np.concatenate([i for _, i in keep])).cuda(self._device)
                near, _ = self._ctx.coarse(self.parent._store, xd, 1, self.metric_, values=False)
                self._store.add_batch(idd, xd, near.reshape(-1).contiguous())

    # -- sizes ---------------------------------------------------------------------------------------------------------------
    def ntotal(self):
        return int(self._store.ntotal()) if self._store is not None else 0

    def nlist(self):
        return int(self._store.nlist()) if self._store is not None else 0

    def d(self):
        return self._d

    # -- save / load: the reference's directory format (quake_index.cpp:170-267, dynamic_inverted_list.cpp:338-520) ------------
    def save(self, dir_path):
        self._require_built("Cannot save an index that was not built")
        if os.path.exists(dir_path) and not os.path.isdir(dir_path):
            raise RuntimeError("save path exists but is not a directory: " + dir_path)
        os.makedirs(dir_path, exist_ok=True)
        with open(os.path.join(dir_path, "metadata.txt"), "w") as f:
            f.write("metric=%d\nlevel=%d\nntotal=%d\nnlist=%d\n" % (self.metric_, self.current_level, self.ntotal(), self.nlist()))
        pids = [int(p) for p in self._store.list_ids()]
        # header and offsets come from the partition sizes alone; the chunks are then streamed one partition at a time, so
        # the host never holds more than one partition (10M x 768 would otherwise cost ~30 GB of RAM)
        rec = self._d * 4 + 8
        offsets = np.concatenate([[0], np.cumsum([self._store.list_size(p) * rec for p in pids])]).astype("<u8")
        with open(os.path.join(dir_path, "partitions"), "wb") as f:
            f.write(struct.pack("<IIQQQ", SERIALIZATION_MAGIC, SERIALIZATION_VERSION, len(pids), self._d * 4, len(pids)))
            f.write(offsets.tobytes())
            f.write(np.asarray(pids, "<u8").tobytes())
            for p in pids:
                vecs, ids = self._store.get_list(p)
                f.write(vecs.astype("<f4").tobytes())  # [codes | ids] per partition
                f.write(ids.astype("<i8").tobytes())
        if self.parent is not None:
            self.parent.save(os.path.join(dir_path, "parent"))

    def load(self, dir_path, n_workers=0):
        if not os.path.isdir(dir_path):
            raise RuntimeError("Cannot load QuakeIndex, directory does not exist: " + dir_path)
        meta = {}
        with open(os.path.join(dir_path, "metadata.txt")) as f:
            for line in f:
                if "=" in line:
                    kk, vv = line.strip().split("=", 1)
                    meta[kk] = vv
        self.metric_ = int(meta.get("metric", 1))
        self.current_level = int(meta.get("level", 0))
        # The file is STREAMED into the device arena (dynamic_inverted_list.cpp:421-520 reads partition by partition too): header
        # and tables first, then windows of whole partitions of at most ~256 MB -- each window is one read into a reusable host
        # buffer and one qk_store_add_batch (rows, ids and their list numbers; append order = file order).  The host never holds
        # more than a window (a 10M x 768 index is a 30 GB file).
        with open(os.path.join(dir_path, "partitions"), "rb") as f:
            head = f.read(32)
            if len(head) < 32:
                raise RuntimeError("Invalid file format (truncated header).")
            magic, version, nlist, code_size, nparts = struct.unpack_from("<IIQQQ", head, 0)
            if magic != SERIALIZATION_MAGIC:
                raise RuntimeError("Invalid file format (bad magic number).")
            if version != SERIALIZATION_VERSION:
                raise RuntimeError("Unsupported file version: %d" % version)
            d = code_size // 4
            rec = code_size + 8
            # nothing is read (or allocated) on the word of a header field before it has been checked against the file's size
            end = os.fstat(f.fileno()).st_size
            start_of_chunks = 32 + 8 * (nparts + 1) + 8 * nparts
            if d <= 0 or code_size % 4 != 0 or start_of_chunks > end:
                raise RuntimeError("Invalid file format (truncated offset / partition id table).")
            raw_offs, raw_pids = f.read(8 * (nparts + 1)), f.read(8 * nparts)
            if len(raw_offs) != 8 * (nparts + 1) or len(raw_pids) != 8 * nparts:
                raise RuntimeError("Invalid file format (truncated offset / partition id table).")
            offs = np.frombuffer(raw_offs, "<u8", nparts + 1)
            pids = np.frombuffer(raw_pids, "<u8", nparts)
            if nparts and (int(offs[0]) > end or int(offs[-1]) > end):
                raise RuntimeError("Invalid file format (truncated partition data).")
            sizes = np.diff(offs.astype(np.int64))
            # the reference seeks to start_of_chunks + offsets[i] per partition (dynamic_inverted_list.cpp:481-494); this loader reads
            # the chunks in one forward pass, which is the same thing exactly when the offsets ascend (chunk i ends where chunk
            # i + 1 starts: one cumulative table) -- anything else is refused rather than misparsed
            if nparts and (sizes < 0).any():
                raise RuntimeError("Invalid file format (partition offsets are not ascending).")
            if (sizes % rec != 0).any():
                raise RuntimeError("Partition chunk size not divisible by (code_size+sizeof(idx_t))")
            if nparts:
                if start_of_chunks + int(offs[-1]) > end:
                    raise RuntimeError("Invalid file format (truncated partition data).")
                f.seek(start_of_chunks + int(offs[0]))
            nvs = sizes // rec
            self._has_ctx = True
            self._d = int(d)
            self._store = self._new_lists(int(d), n_workers, os.path.isdir(os.path.join(dir_path, "parent")))
            self._resident = _ResidentIds()
            window = max(int(256 << 20), int(sizes.max()) if nparts else 0)
            raw = np.empty(window, np.uint8)
            i = 0
            while i < nparts:
                j, nbytes = i, 0
                while j < nparts and (j == i or nbytes + int(sizes[j]) <= window):
                    nbytes += int(sizes[j])
                    j += 1
                got = f.readinto(memoryview(raw)[:nbytes])
                if got != nbytes:
                    raise RuntimeError("Invalid file format (truncated partition data).")
                nrows = int(nvs[i:j].sum())
                wv = np.empty((nrows, d), np.float32)
                wi = np.empty(nrows, np.int64)
                wa = np.empty(nrows, np.int64)
                r0, b0 = 0, 0
                for t in range(i, j):
                    nv = int(nvs[t])
                    self._store.add_list(int(pids[t]))
                    if nv:
                        wv[r0:r0 + nv] = raw[b0:b0 + nv * code_size].view("<f4").reshape(nv, d)
                        wi[r0:r0 + nv] = raw[b0 + nv * code_size:b0 + nv * rec].view("<i8")
                        wa[r0:r0 + nv] = int(pids[t])
                    r0 += nv
                    b0 += int(sizes[t])
                if nrows:
                    self._store.add_batch(wi, wv, wa)
                    self._resident.update(wi)
                i = j
            del raw
        self._next_pid = (int(pids.max()) + 1) if nparts else 0
        pdir = os.path.join(dir_path, "parent")
        if os.path.isdir(pdir):
            self.parent = QuakeIndex(self.current_level + 1, self._device)
            self.parent.load(pdir, n_workers)
        else:
            self.parent = None
        self.initialize_maintenance_policy(MaintenancePolicyParams())  # load resets the policy to defaults (:250-251)

    def __repr__(self):
        return '{"current_level": %d, }' % self.current_level


def compute_recall(ids, gt_ids, k):
    """src/python/utils.py:162-177: per-query |set(ids[:k]) & set(gt[:k])| / k."""
    ids, gt_ids = torch.as_tensor(ids)[:, :k], torch.as_tensor(gt_ids)[:, :k]
    assert ids.shape == gt_ids.shape, (ids.shape, gt_ids.shape)
    out = torch.zeros(ids.shape[0])
    for i in range(ids.shape[0]):
        out[i] = len(set(ids[i].tolist()) & set(gt_ids[i].tolist())) / k
    return out


__all__ = ["QuakeIndex", "IndexBuildParams", "SearchParams", "SearchResult", "SearchTimingInfo", "BuildTimingInfo",
           "ModifyTimingInfo", "MaintenanceTimingInfo", "MaintenancePolicyParams", "compute_recall", "QuakeHipError"]
