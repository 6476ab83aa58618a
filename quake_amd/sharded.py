"""Cluster-sharded search across the GPUs of one node.

One process per GPU.  Partitions (lists) are sharded by list number -- the multi-GPU analogue of the reference's
partition -> core map (PartitionManager::distribute_partitions, partition_manager.cpp:599-602) -- the centroids are
replicated, every rank runs the coarse step for the whole batch and scans only the probed lists it owns (worker_scan's
per-core jobs, query_coordinator.cpp:243-469), then the per-rank top-k (ids and merge keys) are exchanged -- an all-gather when every rank
wants the whole answer, an all-to-all when each rank keeps the answer of its own slice of the batch -- and merged under the
same (key, id) total order (the cross-worker batch_add, query_coordinator.cpp:167-173,231-235).  Besides the all-gather of
the [Q, nprobe] partition lists no other collective is on the search path.

The arithmetic lives in an *engine*:
  GpuEngine   -- libquake_hip.so (qk_search with squared-L2 keys + qk_merge_topk), the product path
  any object with .search_local(q, nprobe, k) -> (ids, keys) and .merge(ids[G,Q,k], keys[G,Q,k]) -> (ids, dist)
so the orchestration can be exercised with world_size-2 gloo tests on CPU (tests/test_sharded_gloo.py), where the
test injects an oracle-backed engine.
"""
import os

import numpy as np


def collectives_active(dist, world):
    """do the collectives run?  With several ranks, always.  With ONE rank only when QUAKE_FORCE_COLLECTIVES=1: the whole
    exchange path -- device all-gather, async all-to-all, all-reduce, broadcasts, stream ordering against the library's kernels
    -- then executes on the real backend (RCCL on one GPU) instead of being short-circuited, which is how the one-GPU test box
    covers the branches an 8-GPU node takes (tests/test_rccl_world1_gpu.py)."""
    return dist is not None and (int(world) > 1 or os.environ.get("QUAKE_FORCE_COLLECTIVES", "0") not in ("", "0"))


def owner_of_list(list_no, world, lists_per_rank=None):
    """block layout when lists_per_rank is given (rank r owns [r*L, (r+1)*L)), else list_no % world
    (partition_manager.cpp:599-602: partition i -> core i % num_workers)."""
    if lists_per_rank:
        return int(list_no) // int(lists_per_rank)
    return int(list_no) % int(world)


def owners_of_lists(list_nos, world, lists_per_rank=None):
    """owner_of_list for an array of list numbers"""
    a = np.asarray(list_nos, dtype=np.int64)
    return a // int(lists_per_rank) if lists_per_rank else a % int(world)


def shard_offsets(offsets, rank, world, lists_per_rank=None):
    """CSR of a GLOBAL index -> (local_offsets [nlist+1], row_selector) keeping every list number but emptying the lists
    this rank does not own (an empty list is skipped by the scan, like the reference skips empty partitions)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    nlist = offsets.shape[0] - 1
    sizes = np.diff(offsets)
    own = owners_of_lists(np.arange(nlist), world, lists_per_rank) == rank
    local_sizes = np.where(own, sizes, 0)
    local_offsets = np.zeros(nlist + 1, np.int64)
    local_offsets[1:] = np.cumsum(local_sizes)
    rows = np.concatenate([np.arange(offsets[p], offsets[p + 1]) for p in range(nlist) if own[p]]) if own.any() else np.zeros(0, np.int64)
    return local_offsets, rows.astype(np.int64)


class GpuEngine:
    """Per-rank engine on libquake_hip.so."""

    def __init__(self, ctx, parent, store, metric):
        self.ctx, self.parent, self.store, self.metric = ctx, parent, store, metric

    def coarse(self, q, nprobe):
        """[n, nprobe] global partition numbers for a slice of the batch (replicated centroids)."""
        return self.ctx.coarse(self.parent, q, nprobe, self.metric, values=False)[0]

    def scan(self, q, pids, k, out=None):
        """local top-k over the probed lists this rank owns: (ids [Q,k], merge keys [Q,k]).  Ranks exchange the merge key
        (squared L2 / dot; sqrt happens after the merge): the context returns squared distances for THIS call only, so a
        context shared with plain searches keeps returning sqrt distances there."""
        self.ctx.set_squared_l2(True)
        try:
            if out is not None:
                return self.ctx.scan_into(self.store, q, pids, k, self.metric, out)
            return self.ctx.scan(self.store, q, pids, k, self.metric)
        finally:
            self.ctx.set_squared_l2(False)

    def merge(self, ids, keys):
        return self.ctx.merge_topk(ids, keys, self.metric)

    def pack(self, ids, keys, world, out=None):
        """the send buffer of the one-collective exchange: uint8 [world][block], block j = ids then keys of queries
        [j*per, (j+1)*per) (qk_pack_topk)"""
        return self.ctx.pack_topk(ids, keys, world, out=out)

    def merge_packed(self, packed, per, k):
        """merge of the receive buffer (block r = rank r's entries for this rank's queries; qk_merge_topk_packed)"""
        return self.ctx.merge_topk_packed(packed, per, k, self.metric)

    def assign(self, x):
        """nearest list (global number) of every vector: the k = 1 parent search of PartitionManager::add"""
        return self.ctx.coarse(self.parent, x, 1, self.metric, values=False)[0].reshape(-1)

    def add_local(self, ids, x, assign):
        self.store.add_batch(ids, x, assign)

    def remove_local(self, ids):
        return self.store.remove_ids(ids)

    def last_scan_bytes(self, q, nprobe, k):
        """algorithmic bytes of this rank's partition scan for the batch q (one synchronising call; bench.py's roofline)"""
        pids = self.coarse(q, nprobe)
        return int(self.ctx.scan(self.store, q, pids, k, self.metric, timing=True)[2]["scan_bytes"])


# ---- k-means across shards ------------------------------------------------------------------------------------
def _all_gather_cat(dist, world, t):
    """[n, ...] per rank -> [world * n, ...] on every rank, rank order (staged through the host when gloo is handed a
    device tensor: the functional tests run two ranks on one GPU)."""
    import torch
    stage = t.is_cuda and str(dist.get_backend()).lower() == "gloo"
    src = t.contiguous().cpu() if stage else t.contiguous()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out.view(world * src.shape[0], -1) if src.dim() > 1 else out, src.view(src.shape[0], -1) if src.dim() > 1 else src)
    return out.to(t.device) if stage else out


def _reduce_partials(dist, world, sums, counts, ordered=True):
    """global sums / counts from the per-rank partials.  ordered: the partials are all-gathered and added in RANK ORDER on
    every rank -- a fixed fp32 summation order, so every rank (and a single-process restatement) gets the same bits; else one
    all-reduce (the order is the collective's: deterministic per topology, not specified)."""
    import torch
    if not collectives_active(dist, world):
        return sums, counts
    ts = sums if torch.is_tensor(sums) else torch.from_numpy(sums)
    tc = counts if torch.is_tensor(counts) else torch.from_numpy(counts)
    if ordered:
        gs = _all_gather_cat(dist, world, ts).view((world,) + tuple(ts.shape))
        gc = _all_gather_cat(dist, world, tc).view((world,) + tuple(tc.shape))
        s = gs[0].clone()
        for r in range(1, world):
            s += gs[r]
        c = gc.sum(0)
    else:
        s, c = ts.clone(), tc.clone()
        dist.all_reduce(s)
        dist.all_reduce(c)
    return (s, c) if torch.is_tensor(sums) else (s.numpy(), c.numpy())


def sharded_kmeans(eng, dist, x, m, metric, niter=5, seed=1234, rank=0, world=1, ordered=True):
    """kmeans() (clustering.cpp:13-97) over a corpus split across ranks: every rank holds x [n_r, d] (equal n_r), the result
    is ONE set of m centroids, identical on every rank, and the assignment of the local rows.

    Per Lloyd iteration: local nearest-centroid assignment and local per-centroid partial sums / counts (the MFMA assign
    and the segmented accumulate of the single-GPU build), then the [m, d] sums and [m] counts are reduced over the ranks
     and every rank applies the same mean update + empty-cluster split.  Training subsample (FAISS: at most 256
    points per centroid) and initial centroids come from each rank's own splitmix64 permutation (seed + rank): rank r
    trains on the first 256*m/world rows of it and contributes centroid slots [r*m/world, (r+1)*m/world).  With world = 1
    this is exactly qk_kmeans.

    IP: the local rows are normalised IN PLACE (eng.normalize_rows) -- the caller stores the normalised copy, as
    QuakeIndex::build does -- so hand over a copy if the raw rows are still needed.

    eng: the per-rank arithmetic -- quake_amd.capi.Context (libquake_hip.so), or any object with normalize_rows,
    rand_perm, kmeans_assign, kmeans_accumulate, kmeans_update (the gloo tests inject an oracle-backed one).
    Returns (centroids [m, d], assign [n_r])."""
    import torch
    is_t = torch.is_tensor(x)
    n, d = x.shape
    if m % world != 0:
        raise ValueError("the number of centroids must be a multiple of the number of ranks")
    m_r = m // world
    if m_r > n:
        raise ValueError("fewer local vectors than centroid slots per rank")
    if metric == "ip":
        x = eng.normalize_rows(x)
    max_pts = 256
    sub = n * world > max_pts * m
    ntrain = (max_pts * m) // world if sub else n
    perm = eng.rand_perm(n, ntrain if sub else m_r, seed + rank)
    if is_t:
        perm_t = torch.from_numpy(perm).to(x.device)
        xt = x[perm_t[:ntrain]].contiguous() if sub else x
        c_local = x[perm_t[:m_r]].contiguous()
    else:
        xt = np.ascontiguousarray(x[perm[:ntrain]]) if sub else x
        c_local = np.ascontiguousarray(x[perm[:m_r]])
    if collectives_active(dist, world):
        call = _all_gather_cat(dist, world, c_local if is_t else torch.from_numpy(c_local))
        c = call if is_t else call.numpy()
    else:
        c = c_local.clone() if is_t else c_local.copy()
    # (assignments only: an engine that can skip the distances -- capi.Context, val = NULL -- says so with kmeans_assign_only)
    assign_of = getattr(eng, "kmeans_assign_only", None) or (lambda xs, cs, mt: eng.kmeans_assign(xs, cs, mt)[0])
    for _ in range(niter):
        a = assign_of(xt, c, metric)
        sums, counts = eng.kmeans_accumulate(xt, a, m, blocked=True)  # the Lloyd driver's blocked order (qk_kmeans's)
        sums, counts = _reduce_partials(dist, world, sums, counts, ordered)
        c, _ = eng.kmeans_update(sums, counts, c)
    if metric == "ip":
        c = eng.normalize_rows(c)
    assign = assign_of(x, c, metric)
    return c, assign


class _Done:
    def wait(self):
        return True


def topk_block_bytes(per, k):
    """bytes of one block of the packed exchange: per*k int64 ids, then per*k float32 keys, padded to 16 (qk_topk_block_bytes)"""
    return (int(per) * int(k) * 12 + 15) & ~15


def pack_topk_host(ids, keys, world):
    """the layout of qk_pack_topk with torch ops (engines without a native pack: the gloo tests' oracle engine)"""
    import torch
    ids = ids if torch.is_tensor(ids) else torch.from_numpy(np.ascontiguousarray(ids))
    keys = keys if torch.is_tensor(keys) else torch.from_numpy(np.ascontiguousarray(keys))
    Q, k = ids.shape
    per = Q // world
    buf = torch.zeros((world, topk_block_bytes(per, k)), dtype=torch.uint8, device=ids.device)
    buf[:, :per * k * 8] = ids.contiguous().view(world, per * k).view(torch.uint8)
    buf[:, per * k * 8:per * k * 12] = keys.contiguous().view(world, per * k).view(torch.uint8)
    return buf


def unpack_topk_host(buf, per, k):
    """receive buffer [G][block] -> (ids [G, per, k], keys [G, per, k])"""
    import torch
    G = buf.shape[0]
    ids = buf[:, :per * k * 8].contiguous().view(torch.int64).view(G, per, k)
    keys = buf[:, per * k * 8:per * k * 12].contiguous().view(torch.float32).view(G, per, k)
    return ids, keys


class ShardedIndex:
    """search() = sharded coarse + all-gather(pids) + local scan + ONE exchange of the per-rank top-k + merge.
    `dist` is torch.distributed (nccl = RCCL on ROCm, or gloo).  The coarse step is split by QUERIES (rank r takes
    the r-th slice of the batch against the replicated centroids) so its cost per rank does not grow with the number of
    ranks; the [Q, nprobe] partition lists are then all-gathered -- the one real exchange the path has besides the
    final top-k gather."""

    def __init__(self, engine, dist=None, world=1, rank=0, result="all"):
        """result = "all": every rank ends with the whole [Q, k] answer (all-gather of the per-rank top-k, G*Q*k*12 bytes
        received per rank).  result = "owner": rank r ends with the answer of ITS slice of the batch, rows
        [r*Q/G, (r+1)*Q/G) -- the per-rank top-k are exchanged with an all-to-all (Q*k*12 bytes per rank, independent of
        the number of ranks) and each rank merges only its own queries; this is the serving layout (a query's answer goes
        back to the rank that received it) and what bench.py times."""
        self.engine, self.dist, self.world, self.rank = engine, dist, int(world), int(rank)
        if result not in ("all", "owner"):
            raise ValueError("result must be 'all' or 'owner'")
        self.result = result
        self._g_ids = self._g_keys = self._g_pids = None
        self._x_send = self._x_recv = None
        self.last_pids = None
        # gloo (two ranks sharing one GPU in the functional tests) has no device all-to-all: stage through the host there
        self._stage_host = False
        if collectives_active(dist, self.world):
            try:
                self._stage_host = str(dist.get_backend()).lower() == "gloo"
            except Exception:
                self._stage_host = False

    def _exchange(self, buf_name, t):
        """all-to-all of [G, per, k] blocks: block j goes to rank j; returns [G(source), per, k]."""
        import torch
        t = t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t))
        if self._stage_host and t.is_cuda:
            h = t.contiguous().cpu()
            hb = torch.empty_like(h)
            self.dist.all_to_all_single(hb.view(-1), h.view(-1))
            return hb.to(t.device), _Done()
        buf = getattr(self, buf_name)
        if buf is None or tuple(buf.shape) != tuple(t.shape) or buf.device != t.device or buf.dtype != t.dtype:
            buf = torch.empty_like(t)
            setattr(self, buf_name, buf)
        return buf, self.dist.all_to_all_single(buf.view(-1), t.contiguous().view(-1), async_op=True)

    def _gather(self, buf_name, t, dtype):
        import torch
        t = t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t))
        shape = (self.world,) + tuple(t.shape)
        if self._stage_host and t.is_cuda:
            h = t.contiguous().cpu()
            hb = torch.empty(shape, dtype=dtype)
            self.dist.all_gather_into_tensor(hb.view(-1, h.shape[-1]), h)
            return hb.to(t.device)
        buf = getattr(self, buf_name)
        if buf is None or tuple(buf.shape) != shape or buf.device != t.device:
            buf = torch.empty(shape, dtype=dtype, device=t.device)
            setattr(self, buf_name, buf)
        # concatenation form ([world*n, c]) is accepted by both the RCCL and the gloo backend
        self.dist.all_gather_into_tensor(buf.view(-1, t.shape[-1]), t.contiguous())
        return buf

    def search(self, q, nprobe, k, out=None):
        import torch
        Q = q.shape[0]
        if not collectives_active(self.dist, self.world):
            pids = self.last_pids = self.engine.coarse(q, nprobe)
            ids, keys = self.engine.scan(q, pids, k, out=out)
            return self.engine.merge(ids.reshape((1,) + tuple(ids.shape)), keys.reshape((1,) + tuple(keys.shape)))
        if Q % self.world != 0:
            raise ValueError("batch size must be a multiple of the number of ranks")
        per = Q // self.world
        pl = self.engine.coarse(q[self.rank * per:(self.rank + 1) * per], nprobe)
        pids = self._gather("_g_pids", pl, torch.int64).view(Q, -1)
        if not torch.is_tensor(q):
            pids = pids.numpy()
        self.last_pids = pids  # [Q, nprobe] of the whole batch, on every rank: what hit tracking records
        ids, keys = self.engine.scan(q, pids, k, out=out)
        if self.result == "owner":
            # ONE all-to-all: an entry's id and key travel together (12-byte records, block j = this rank's results for the
            # queries rank j owns); the receive buffer is merged where it lands
            native = hasattr(self.engine, "pack")
            send = self.engine.pack(ids, keys, self.world, out=self._x_send) if native else pack_topk_host(ids, keys, self.world)
            self._x_send = send if native else None
            recv, w = self._exchange("_x_recv", send)
            w.wait()
            if native:
                return self.engine.merge_packed(recv, per, k)
            x_ids, x_keys = unpack_topk_host(recv, per, k)
            return self.engine.merge(x_ids, x_keys)
        g_ids = self._gather("_g_ids", ids, torch.int64)
        g_keys = self._gather("_g_keys", keys, torch.float32)
        return self.engine.merge(g_ids, g_keys)

    # -- dynamic updates (partition_manager.cpp:123-320), sharded: the batch is known to every rank (like the queries), each
    # rank applies the part that concerns the lists it owns; no collective is needed.  Maintenance (split / delete / refine) changes
    # the replicated centroids: quake_amd/sharded_maintenance.py.
    def add(self, x, ids, lists_per_rank=None):
        """x [n, d], ids [n] on every rank.  Returns the number of vectors this rank stored."""
        import torch
        assign = self.engine.assign(x)
        a = assign.cpu().numpy() if torch.is_tensor(assign) else np.asarray(assign)
        own = owners_of_lists(a, self.world, lists_per_rank) == self.rank
        if not own.any():
            return 0
        sel = np.nonzero(own)[0]
        if torch.is_tensor(x):
            st = torch.as_tensor(sel, device=x.device)
            self.engine.add_local(torch.as_tensor(ids).to(x.device)[st].contiguous(), x[st].contiguous(), assign[st].contiguous())
        else:
            self.engine.add_local(np.asarray(ids)[sel], np.asarray(x)[sel], a[sel])
        return int(own.sum())

    def remove(self, ids):
        """ids on every rank; a rank removes the ones it holds.  Returns how many this rank removed."""
        import torch
        h = ids.cpu().numpy() if torch.is_tensor(ids) else np.asarray(ids)
        return int(self.engine.remove_local(np.ascontiguousarray(h, dtype=np.int64)))
