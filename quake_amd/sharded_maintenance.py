"""Maintenance of a cluster-sharded index: split / delete / local refinement and the policy that
drives them, when list p lives on rank owner(p) = p % world and the centroids are replicated on every rank.

The reference mutates one PartitionManager (partition_manager.cpp:392-554) and one parent index; here every primitive is
a COLLECTIVE call -- all ranks enter it with the same arguments and leave it with the same replicated state (centroids,
partition numbers, next partition id) -- and the rows move between ranks only where ownership changes:

  split   the owner runs the 2-means (qk_kmeans) on its list; the two centroids are all-gathered; the halves are routed to
          the owners of the two NEW partition numbers (all-to-all with uneven splits)
  delete  the centroids disappear everywhere; the owner assigns its rows to the nearest remaining centroid (replicated, so
          a local qk_coarse) and routes each row to the owner of its target list
  refine  Lloyd iterations over the union of the named lists: local assignment + local per-centroid partial sums, the
          [m, d] sums / [m] counts reduced in rank order (the same ordered reduction as the cross-shard k-means), the same
          mean update on every rank; at the end every row is routed to the owner of the list it was assigned to
  policy  MaintenancePolicy (quake_amd.maintenance) runs unchanged on every rank over GLOBAL partition sizes (one
          all-reduce per mutation) and the hit window of the whole batch (every rank sees the full [Q, nprobe] partition
          lists after the search's all-gather); the latency grid is profiled on rank 0 and broadcast, so every rank takes
          the same decisions

With world = 1 every primitive does exactly what index.QuakeIndex does (tests/test_sharded_maintenance_gpu.py).

The per-rank arithmetic and storage sit behind a small protocol (`local`):
  d, metric ("l2" | "ip")
  list_ids(), list_size(p), get_list(p) -> (vecs, ids), add_list(p), remove_list(p), add_entries(p, ids, vecs), ntotal()
  centroids(pids) -> [m, d], add_centroids(c, pids), remove_centroids(pids), set_centroids(pids, c)
  nearest(x, k) -> [n, k] partition numbers (over the replicated centroids)
  two_means(x) -> (centroids [2, d], assign [n], x as stored)
  kmeans_assign(x, c) -> assign [n];  kmeans_accumulate(x, assign, m) -> (sums [m, d], counts [m])
GpuPartitions implements it on libquake_hip.so (over an index.QuakeIndex); the gloo tests inject an oracle-backed one.
"""
import numpy as np

from .sharded import collectives_active, owner_of_list, owners_of_lists


_SEQ = 1 << 40  # row-order keys: (position of the source list) * _SEQ + row


# Row blocks (the [n, d] vectors: 512 bytes a row at d = 128) stay in whatever memory the `local` keeps them in -- CUDA tensors
# for GpuPartitions: extracted from the arena, routed between ranks and appended again without ever crossing PCIe -- while the
# control data (ids, assignments, order keys: 8 bytes a row) is host numpy, where the bookkeeping that reads it lives.  The
# three things the orchestration does with a row block, for both kinds:
def _is_t(a):
    import torch
    return torch.is_tensor(a)


def _cat_rows(blocks):
    """concatenation of [n_i, d] blocks of one kind (at least one block)"""
    if _is_t(blocks[0]):
        import torch
        return torch.cat(list(blocks), 0) if len(blocks) > 1 else blocks[0]
    return np.concatenate(blocks) if len(blocks) > 1 else np.ascontiguousarray(blocks[0])


def _take_rows(v, sel):
    """v[sel] for a host selector (boolean mask or index array) over a row block of either kind"""
    sel = np.asarray(sel)
    if not _is_t(v):
        return np.ascontiguousarray(v[sel])
    import torch
    idx = np.nonzero(sel)[0] if sel.dtype == np.bool_ else sel.astype(np.int64)
    return v[torch.from_numpy(np.ascontiguousarray(idx)).to(v.device)]


class Comm:
    """The collective shapes maintenance needs over torch.distributed (RCCL or gloo).  Control data is host numpy in and out
    (staged on `device` for the collective: RCCL wants device memory, gloo takes host); row blocks handed to route_rows as
    device tensors are exchanged and returned in device memory."""

    def __init__(self, dist=None, world=1, rank=0, device=None):
        self.dist, self.world, self.rank = dist, int(world), int(rank)
        self.active = collectives_active(dist, self.world)
        self.device = device
        if self.active and device is None:
            import torch
            self.device = torch.device("cuda", torch.cuda.current_device()) if str(dist.get_backend()).lower() == "nccl" else torch.device("cpu")

    def _t(self, a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def all_gather(self, a):
        """same-shape array per rank -> [world, ...]"""
        a = np.ascontiguousarray(a)
        if not self.active:
            return a[None]
        import torch
        t = self._t(a.reshape(-1))
        out = torch.empty((self.world * t.shape[0],), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape((self.world,) + a.shape)

    def all_gather_var(self, a):
        """1-D int64 arrays of different lengths -> their concatenation in rank order, on every rank"""
        a = np.ascontiguousarray(a, dtype=np.int64)
        if not self.active:
            return a
        n = self.all_gather(np.array([len(a)], np.int64)).reshape(-1)
        pad = np.full(max(int(n.max()), 1), -1, np.int64)
        pad[:len(a)] = a
        g = self.all_gather(pad)
        return np.concatenate([g[r, :int(n[r])] for r in range(self.world)])

    def all_sum(self, a):
        """integer sum over ranks (exact)"""
        a = np.ascontiguousarray(a, dtype=np.int64)
        if not self.active:
            return a
        t = self._t(a)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def from_owner(self, a, src, dtype):
        """1-D array known on rank `src` only -> on every rank (length first, then the data)"""
        if not self.active:
            return np.asarray(a, dtype=dtype)
        import torch
        n = torch.tensor([len(a) if self.rank == src else 0], dtype=torch.int64, device=self.device)
        self.dist.broadcast(n, src)
        n = int(n.item())
        t = self._t(np.asarray(a, dtype=dtype)) if self.rank == src else torch.empty((n,), dtype=getattr(torch, np.dtype(dtype).name), device=self.device)
        if n:
            self.dist.broadcast(t, src)
        return t.cpu().numpy()

    def route_rows(self, dest, *arrays, key=None):
        """arrays[i] [n, ...] per rank, row j goes to rank dest[j] (an all-to-all with uneven splits).  Returns the received
        arrays: source-rank-major, rows of one source in their original order -- or, with key [n] (int64, unique over all
        ranks), in ascending key order, i.e. the order ONE rank holding every row would have seen them in."""
        dest = np.asarray(dest, dtype=np.int64)
        if not self.active:
            return tuple(a if _is_t(a) else np.ascontiguousarray(a) for a in arrays)
        if key is not None:
            got = self.route_rows(dest, np.asarray(key, dtype=np.int64), *arrays)
            o = np.argsort(got[0], kind="stable")
            return tuple(_take_rows(a, o) for a in got[1:])
        import torch
        order = np.argsort(dest, kind="stable")
        send = np.bincount(dest, minlength=self.world).astype(np.int64)
        recv = self.all_gather(send)[:, self.rank]
        nrecv = int(recv.sum())
        splits = dict(output_split_sizes=[int(v) for v in recv], input_split_sizes=[int(v) for v in send])
        out = []
        for a in arrays:
            if _is_t(a):
                # a row block in device memory: permuted, exchanged and handed back THERE (RCCL moves it GPU to GPU; under gloo --
                # ranks sharing one GPU in the functional tests -- the collective itself is staged through the host)
                src = _take_rows(a, order).contiguous()
                stage = src.is_cuda and str(self.dist.get_backend()).lower() == "gloo"
                sbuf = src.cpu() if stage else src
                dst = torch.empty((nrecv,) + tuple(src.shape[1:]), dtype=src.dtype, device=sbuf.device)
                self.dist.all_to_all_single(dst, sbuf, **splits)
                out.append(dst.to(src.device) if stage else dst)
                continue
            a = np.ascontiguousarray(a)
            w = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
            src = self._t(a[order].reshape(-1, w) if a.shape[0] else a.reshape(0, w))
            dst = torch.empty((nrecv, w), dtype=src.dtype, device=src.device)
            self.dist.all_to_all_single(dst, src, **splits)
            out.append(dst.cpu().numpy().reshape((nrecv,) + a.shape[1:]))
        return tuple(out)


class GpuPartitions:
    """`local` protocol on libquake_hip.so: the partitions one rank holds, as an index.QuakeIndex whose store keeps EVERY
    partition number (lists of other ranks are empty) and whose parent holds all the centroids."""

    def __init__(self, index):
        self.ix = index
        self.d = index.d()
        self.metric = "ip" if int(index.metric_) == 0 else "l2"

    # storage
    def list_ids(self):
        return [int(p) for p in self.ix._store.list_ids()]

    def list_size(self, p):
        return int(self.ix._store.list_size(int(p)))

    def get_list(self, p):
        """(vectors: a CUDA tensor extracted from the arena on the device, ids: host array)"""
        return self.ix._store.get_list_device(int(p))

    def empty_rows(self):
        import torch
        return torch.empty((0, self.d), dtype=torch.float32, device=torch.device("cuda", self.ix._device))

    def add_list(self, p):
        self.ix._store.add_list(int(p))
        self.ix._next_pid = max(self.ix._next_pid, int(p) + 1)

    def remove_list(self, p):
        ids = self.ix._store.get_list_ids(int(p))  # (the store's host mirror of the ids: no row leaves the arena for this)
        self.ix._resident.discard_all(np.asarray(ids, dtype=np.int64))
        self.ix._store.remove_list(int(p))

    def add_entries(self, p, ids, vecs):
        import torch
        if len(ids):
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            vd = vecs if torch.is_tensor(vecs) else torch.from_numpy(np.ascontiguousarray(vecs, dtype=np.float32))
            vd = vd.to(device=torch.device("cuda", self.ix._device), dtype=torch.float32).contiguous()
            self.ix._store.add_entries(int(p), torch.from_numpy(ids).to(vd.device), vd)
            self.ix._resident.update(ids)

    def ntotal(self):
        return self.ix.ntotal()

    # replicated centroids
    def _pid_t(self, pids):
        import torch
        return torch.tensor([int(p) for p in pids], dtype=torch.int64)

    def centroids(self, pids):
        return self.ix.parent.get(self._pid_t(pids)).numpy()

    def add_centroids(self, c, pids):
        import torch
        self.ix.parent.add(torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)), self._pid_t(pids))

    def remove_centroids(self, pids):
        self.ix.parent.remove(self._pid_t(pids))

    def set_centroids(self, pids, c):
        import torch
        self.ix.parent.modify(self._pid_t(pids), torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)))

    # arithmetic
    def _dev(self, x):
        import torch
        if torch.is_tensor(x):
            return x.to(device=torch.device("cuda", self.ix._device), dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda(self.ix._device)

    def nearest(self, x, k):
        if len(x) == 0:
            return np.zeros((0, int(k)), np.int64)
        return self.ix._ctx.coarse(self.ix.parent._store, self._dev(x), int(k), self.ix.metric_)[0].cpu().numpy()

    def two_means(self, x):
        """-> (centroids [2, d] host, assign [n] host, the rows as stored -- normalised for IP -- where x lives)"""
        cent, assign, xs = self.ix._ctx.kmeans(self._dev(x), 2, self.ix.metric_, niter=5, seed=1234)
        return cent.cpu().numpy(), assign.cpu().numpy(), xs

    def kmeans_assign(self, x, c):
        if len(x) == 0:
            return np.zeros(0, np.int64)
        return self.ix._ctx.kmeans_assign(self._dev(x), self._dev(c), self.metric, values=False)[0].cpu().numpy()

    def kmeans_accumulate(self, x, assign, m):
        import torch
        if len(x) == 0:
            return np.zeros((int(m), self.d), np.float32), np.zeros(int(m), np.int64)
        s, c = self.ix._ctx.kmeans_accumulate(self._dev(x), torch.from_numpy(np.ascontiguousarray(assign, dtype=np.int64)).cuda(self.ix._device), int(m))
        return s.cpu().numpy(), c.cpu().numpy()


class ShardedPartitions:
    """The collective maintenance primitives + the interface MaintenancePolicy drives (the private surface of
    index.QuakeIndex: _list_ids, _partition_sizes, _reassign_targets, _neighbour_partitions, _split_partitions,
    _delete_partitions, _add_partitions, refine_partitions, nlist, ntotal, d)."""

    def __init__(self, local, dist=None, world=1, rank=0, device=None):
        self.local = local
        self.comm = Comm(dist, world, rank, device)
        self.world, self.rank = int(world), int(rank)
        ids = local.list_ids()
        self._next_pid = (max(ids) + 1) if ids else 0
        self._gsizes = None
        self._refresh_sizes()  # (construction is collective: every rank builds its ShardedPartitions at the same point)
        self.maintenance_policy_params_ = None
        self.maintenance_policy_ = None
        self._policy_cost_estimator = None

    def owner(self, p):
        return owner_of_list(p, self.world)

    def _empty_rows(self):
        """a row block of no rows, of the kind the `local` keeps its rows in (device tensor / host array)"""
        f = getattr(self.local, "empty_rows", None)
        return f() if f is not None else np.zeros((0, self.d()), np.float32)

    def owns(self, p):
        return self.owner(p) == self.rank

    # -- replicated facts -------------------------------------------------------------------------------------------------
    def d(self):
        return int(self.local.d)

    def _list_ids(self):
        return self.local.list_ids()

    def nlist(self):
        return len(self.local.list_ids())

    def _refresh_sizes(self):
        """COLLECTIVE: global list sizes (one all-reduce).  Called at the end of every mutating collective (construction, add,
        remove, split, delete, refine), so that the getters below -- ntotal(), _partition_sizes(), record_query_hits() -- never
        start a collective themselves and may be called on one rank only (logging on rank 0 would otherwise hang the job)."""
        pids = self.local.list_ids()
        g = self.comm.all_sum(np.array([self.local.list_size(p) for p in pids], np.int64))
        self._gsizes = dict(zip(pids, (int(v) for v in g)))

    def _sizes(self):
        if self._gsizes is None:
            raise RuntimeError("global list sizes are stale: a mutating collective did not finish with _refresh_sizes()")
        return self._gsizes

    def _partition_sizes(self, pids):
        s = self._sizes()
        return [s[int(p)] for p in pids]

    def ntotal(self):
        return int(sum(self._sizes().values()))

    def _reassign_targets(self, pid):
        """index.QuakeIndex._reassign_targets, computed by the owner of `pid` and handed to every rank"""
        src = self.owner(pid)
        pk = np.zeros(0, np.int64)
        if self.rank == src:
            vecs, _ = self.local.get_list(pid)
            near = self.local.nearest(vecs, 2).reshape(-1)
            near = near[(near != int(pid)) & (near >= 0)]
            u, c = np.unique(near, return_counts=True)
            pk = np.concatenate([u, c]).astype(np.int64)
        pk = self.comm.from_owner(pk, src, np.int64)
        h = len(pk) // 2
        return [int(v) for v in pk[:h]], [int(v) for v in pk[h:]]

    def _neighbour_partitions(self, pids, radius):
        near = self.local.nearest(self.local.centroids(pids), int(radius)).reshape(-1)  # replicated: no exchange
        return sorted({int(v) for v in near if v >= 0})

    # -- split (partition_manager.cpp:392-444) -------------------------------------------------------------------------------
    def _split_partitions(self, pids):
        """-> {"centroids" [2n, d] (replicated), "vectors"/"vector_ids": the halves THIS rank computed (empty for lists of
        other ranks)}; _add_partitions routes them to the owners of the new partition numbers."""
        d = self.d()
        cent = np.zeros((2 * len(pids), d), np.float32)
        vecs, ids = [], []
        sizes = self._partition_sizes(pids)
        for j, (p, size) in enumerate(zip(pids, sizes)):
            assert size >= 4, "Partition must have at least 8 vectors to split."  # (the reference's message, :412)
            if self.owns(p):
                v, i = self.local.get_list(p)
                c, a, vs = self.local.two_means(v)
                cent[2 * j:2 * j + 2] = c
                for h in range(2):
                    vecs.append(_take_rows(vs, a == h))
                    ids.append(np.ascontiguousarray(i[a == h]))
            else:
                vecs += [self._empty_rows()] * 2
                ids += [np.zeros(0, np.int64)] * 2
        g = self.comm.all_gather(cent)  # rows of a split come from the rank that owns the partition
        for j, p in enumerate(pids):
            cent[2 * j:2 * j + 2] = g[self.owner(p), 2 * j:2 * j + 2]
        return {"centroids": cent, "vectors": vecs, "vector_ids": ids}

    def _add_partitions(self, clustering):  # :489-520
        n = len(clustering["vectors"])
        new_pids = list(range(self._next_pid, self._next_pid + n))
        self._next_pid += n
        slot = np.concatenate([np.full(len(i), j, np.int64) for j, i in enumerate(clustering["vector_ids"])]) if n else np.zeros(0, np.int64)
        dest = owners_of_lists(np.asarray(new_pids, np.int64)[slot], self.world)
        v = _cat_rows(clustering["vectors"]) if n else self._empty_rows()
        i = np.concatenate(clustering["vector_ids"]) if n else np.zeros(0, np.int64)
        rslot, rv, ri = self.comm.route_rows(dest, slot, v, i)
        for j, pid in enumerate(new_pids):
            self.local.add_list(pid)
            if self.owns(pid):
                m = rslot == j
                self.local.add_entries(pid, ri[m], _take_rows(rv, m))
        self.local.add_centroids(clustering["centroids"], new_pids)
        self._refresh_sizes()
        return new_pids

    # -- delete (:522-554) -----------------------------------------------------------------------------------------------------
    def _delete_partitions(self, pids, reassign=True):
        held_v, held_i, held_k = [self._empty_rows()], [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
        for j, p in enumerate(pids):
            if self.owns(p) and reassign:
                v, i = self.local.get_list(p)
                held_v.append(v)
                held_i.append(i)
                held_k.append(_SEQ * j + np.arange(len(i), dtype=np.int64))  # the order one rank would re-add them in
        self.local.remove_centroids(pids)
        for p in pids:
            self.local.remove_list(p)
        if reassign:
            v, i = _cat_rows(held_v), np.concatenate(held_i)
            target = self.local.nearest(v, 1).reshape(-1)  # nearest REMAINING centroid (PartitionManager::add, :219-230)
            dest = owners_of_lists(target, self.world)
            rt, rv, ri = self.comm.route_rows(dest, target, v, i, key=np.concatenate(held_k))
            for t in np.unique(rt):
                m = rt == t
                self.local.add_entries(int(t), ri[m], _take_rows(rv, m))
        self._refresh_sizes()

    # -- local refinement (:446-487 -> kmeans_refine_partitions, clustering.cpp:99-182) -----------------------------------------
    def refine_partitions(self, partition_ids=None, iterations=0):
        import torch
        pids = self.local.list_ids() if partition_ids is None else [int(p) for p in (partition_ids.tolist() if torch.is_tensor(partition_ids) else partition_ids)]
        if not pids:
            return
        if len(set(pids)) != len(pids):
            raise RuntimeError("refine_partitions: duplicate partition")
        m, d = len(pids), self.d()
        c = np.ascontiguousarray(self.local.centroids(pids), dtype=np.float32)
        xs, ids, seq = [self._empty_rows()], [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
        for j, p in enumerate(pids):
            if self.owns(p):
                v, i = self.local.get_list(p)
                xs.append(v)
                ids.append(i)
                seq.append(_SEQ * j + np.arange(len(i), dtype=np.int64))  # position in the concatenation of the lists (:104-108)
        x, ids, seq = _cat_rows(xs), np.concatenate(ids), np.concatenate(seq)
        total = int(self.comm.all_sum(np.array([len(ids)], np.int64))[0])
        sums = counts = a = None
        for it in range(max(int(iterations), 1)):  # clustering.cpp:110
            if it > 0:
                # centroids = sums / counts, a count of 0 gives NaN exactly like the reference (:122-124)
                with np.errstate(invalid="ignore", divide="ignore"):
                    c = (sums / counts.astype(np.float32)[:, None]).astype(np.float32)
            a = self.local.kmeans_assign(x, c)
            ps, pc = self.local.kmeans_accumulate(x, a, m)
            g = self.comm.all_gather(ps)  # ordered reduction: partials added in rank order, same bits on every rank
            sums = g[0].copy()
            for r in range(1, g.shape[0]):
                sums += g[r]
            counts = self.comm.all_sum(pc)
            order = np.argsort(a, kind="stable")  # the per-vector append into the new partitions (:174)
            x, ids, a, seq = _take_rows(x, order), ids[order], a[order], seq[order]
            # position of every row in the order ONE process would hold them in after this pass: by (list it was appended to,
            # position before the pass) -- the stable sort above, made global: its rank among all ranks' previous positions
            # (local rows stay in ascending position order, which the next pass's stable sort relies on)
            if self.comm.active:
                seq = a.astype(np.int64) * _SEQ + np.searchsorted(np.sort(self.comm.all_gather_var(seq)), seq)
        if int(counts.sum()) != total:
            raise RuntimeError("refine_partitions: %d of %d vectors could not be assigned (NaN centroid from an emptied cluster)"
                               % (total - int(counts.sum()), total))
        dest = owners_of_lists(np.asarray(pids, np.int64)[a], self.world)
        # rows of a new list arrive in ascending position: the order a single rank would append them in
        ra, rx, ri = self.comm.route_rows(dest, a, x, ids, key=seq)
        # replace the partitions (:481-483): every old list goes before the first new row comes in (a row that changes lists
        # must not be forgotten again when its old list is dropped)
        for p in pids:
            self.local.remove_list(p)
            self.local.add_list(p)
        for j, p in enumerate(pids):
            if self.owns(p):
                sel = ra == j
                self.local.add_entries(p, ri[sel], _take_rows(rx, sel))
        self.local.set_centroids(pids, c)  # parent_->modify (:478): "the centroids used for the last assignment"
        self._refresh_sizes()

    # -- sharded add / remove bookkeeping -----------------------------------------------------------------------------------------
    def invalidate_sizes(self):
        """after ShardedIndex.add / remove changed the lists (collective, like the add / remove itself)"""
        self._refresh_sizes()

    # -- the policy (quake_index.cpp:152-168; maintenance_policies.cpp) ---------------------------------------------------------
    def initialize_maintenance_policy(self, params, cost_estimator=None):
        self.maintenance_policy_params_ = params
        self.maintenance_policy_ = None
        self._policy_cost_estimator = cost_estimator

    def _policy(self):
        if self.maintenance_policy_ is None:
            from .maintenance import (DEFAULT_LATENCY_ESTIMATOR_RANGE_K, DEFAULT_LATENCY_ESTIMATOR_RANGE_N, ListScanLatencyEstimator,
                                      MaintenanceCostEstimator, MaintenancePolicy)
            ce = self._policy_cost_estimator
            if ce is None:
                # one latency grid for all ranks: rank 0 profiles its device, the others take its numbers
                p = self.maintenance_policy_params_
                nv, kv = DEFAULT_LATENCY_ESTIMATOR_RANGE_N, DEFAULT_LATENCY_ESTIMATOR_RANGE_K
                if self.rank == 0:
                    est = ListScanLatencyEstimator(self.d(), nv, kv)
                    grid = np.array(est.scan_latency_model_, np.float64).reshape(-1)
                else:
                    est = ListScanLatencyEstimator(self.d(), nv, kv, profile_fn=lambda n, k: 0.0)
                    grid = np.zeros(0, np.float64)
                grid = self.comm.from_owner(grid, 0, np.float64).reshape(len(nv), len(kv))
                est.scan_latency_model_ = [[float(v) for v in row] for row in grid]
                ce = MaintenanceCostEstimator(self.d(), p.alpha, 10, latency_estimator=est)
            self.maintenance_policy_ = MaintenancePolicy(self, self.maintenance_policy_params_, ce)
        return self.maintenance_policy_

    def record_query_hits(self, pids):
        """pids [Q, nprobe]: the partition lists of the WHOLE batch (what every rank holds after the search's all-gather)"""
        arr = np.asarray(pids).reshape(len(pids), -1)
        s = self._sizes()
        uniq = np.unique(arr[arr >= 0])
        lut = np.zeros(int(uniq.max()) + 1 if uniq.size else 1, np.int64)
        lut[uniq] = [s[int(u)] for u in uniq]
        self._policy().hit_count_tracker_.add_batch(arr, lut[np.clip(arr, 0, None)])

    def maintenance(self):
        if self.maintenance_policy_params_ is None:
            raise RuntimeError("[QuakeIndex::maintenance()] No maintenance policy set.")
        return self._policy().perform_maintenance()


class ShardedQuakeIndex:
    """One rank of a cluster-sharded dynamic index on libquake_hip.so: build / search / add / remove / maintenance with the
    method names of QuakeIndex (quake_index.h:18-142).  Every method is collective: all ranks call it with the same
    arguments (build: each with its own rows)."""

    def __init__(self, index, dist=None, world=1, rank=0, result="all"):
        from .sharded import GpuEngine, ShardedIndex
        self.index = index  # this rank's index.QuakeIndex: every partition number, rows of the owned partitions only
        self.dist, self.world, self.rank = dist, int(world), int(rank)
        self.local = GpuPartitions(index)
        self.partitions = ShardedPartitions(self.local, dist, world, rank)
        self.engine = GpuEngine(index._ctx, index.parent._store, index._store, self.local.metric)
        self.searcher = ShardedIndex(self.engine, dist, world, rank, result=result)
        self.track_hits = False
        self._pending_hits = []  # [Q, nprobe] lists of tracked searches not yet handed to the policy (device tensors)

    # -- construction ---------------------------------------------------------------------------------------------------------
    @classmethod
    def from_global(cls, dist, world, rank, centroids, offsets, ids, vecs, metric, device=0, result="all"):
        """every rank is handed the same clustering (CSR over all lists) and keeps the lists it owns"""
        from .index import QuakeIndex
        from .sharded import shard_offsets
        lo, rows = shard_offsets(offsets, rank, world)
        ix = QuakeIndex.from_partitions(centroids, lo, np.asarray(ids)[rows], np.asarray(vecs)[rows], metric, device)
        return cls(ix, dist, world, rank, result)

    @classmethod
    def build(cls, dist, world, rank, x, ids, nlist, metric="l2", niter=5, seed=1234, device=0, result="all"):
        """QuakeIndex::build (quake_index.cpp:29-90) over a corpus split across ranks: x [n_r, d], ids [n_r] are THIS rank's
        rows.  Cross-shard k-means (sharded_kmeans), then every row travels once, to the owner of its list."""
        import torch
        from . import capi
        from .index import QuakeIndex, _context
        from .sharded import sharded_kmeans
        ctx = _context(device)
        xd = torch.as_tensor(x, dtype=torch.float32).cuda(device).contiguous()
        c, a = sharded_kmeans(ctx, dist, xd, int(nlist), metric, niter=int(niter), seed=int(seed), rank=rank, world=world)
        a = a.cpu().numpy()
        comm = Comm(dist, world, rank)
        dest = owners_of_lists(a, world)
        # the rows travel GPU to GPU (IP: the normalised copy); assignments and ids -- 16 bytes a row -- through the host
        ra, rx, ri = comm.route_rows(dest, a, xd, np.asarray(ids, dtype=np.int64))
        order = np.argsort(ra, kind="stable")
        offsets = np.zeros(int(nlist) + 1, np.int64)
        offsets[1:] = np.cumsum(np.bincount(ra, minlength=int(nlist)))
        ix = QuakeIndex.from_partitions(c, offsets, ri[order], _take_rows(rx, order), metric, device)
        return cls(ix, dist, world, rank, result)

    # -- search / add / remove ---------------------------------------------------------------------------------------------------
    def search(self, q, nprobe, k=None):
        """search(q, nprobe, k) -> (ids, distances) on the device; or, with the reference's signature, search(q, search_params)
        -> an object with .ids / .distances on the host (what quake_amd.workload.replay_workload drives): the batch is padded
        to a multiple of the number of ranks and the padding rows are dropped again."""
        import torch
        if k is None and hasattr(nprobe, "k"):
            sp = nprobe
            kk = int(sp.k) if sp.k and sp.k > 0 else 1
            xd = torch.as_tensor(q, dtype=torch.float32).cuda(self.index._device).contiguous()
            n = int(xd.shape[0])
            pad = (-n) % self.world
            if pad:
                xd = torch.cat([xd, xd[-1:].expand(pad, -1)]).contiguous()
            was, was_track = self.searcher.result, self.track_hits
            self.searcher.result = "all"  # the harness wants the whole answer on every rank
            self.track_hits = False       # (the padding rows are not queries: hits are recorded below, for the first n rows)
            try:
                ids, dist = self.search(xd, min(max(int(sp.nprobe), 1), self.nlist()), kk)
            finally:
                self.searcher.result, self.track_hits = was, was_track
            if self.track_hits:
                p = self.searcher.last_pids
                self._pending_hits.append(p[:n].clone() if torch.is_tensor(p) else np.array(p[:n]))
                if len(self._pending_hits) >= 64:
                    self._flush_hits()
            from .index import SearchResult
            res = SearchResult()
            res.ids, res.distances = ids[:n].cpu(), dist[:n].cpu()
            return res
        out = self.searcher.search(q, int(nprobe), int(k))
        if self.track_hits:
            # the batch's [Q, nprobe] lists stay where the all-gather left them (a copy: the buffer is reused by the next search);
            # they cross to the host in ONE transfer when the policy needs them (maintenance(), or 64 batches pending) -- a
            # search with hit tracking on does not synchronise the device
            p = self.searcher.last_pids
            self._pending_hits.append(p.clone() if torch.is_tensor(p) else np.array(p))
            if len(self._pending_hits) >= 64:
                self._flush_hits()
        return out

    def _flush_hits(self):
        import torch
        pend, self._pending_hits = self._pending_hits, []
        if not pend:
            return
        if all(torch.is_tensor(p) for p in pend) and len({tuple(p.shape[1:]) for p in pend}) == 1:
            sizes = [int(p.shape[0]) for p in pend]
            host = torch.cat(pend, 0).cpu().numpy()  # one transfer
            at = 0
            for n in sizes:
                self.partitions.record_query_hits(host[at:at + n])
                at += n
            return
        for p in pend:
            self.partitions.record_query_hits(p.cpu().numpy() if torch.is_tensor(p) else np.asarray(p))

    def add(self, x, ids):
        """x [n, d], ids [n] on every rank; each rank stores the rows whose nearest list it owns"""
        import torch
        xd = torch.as_tensor(x, dtype=torch.float32).cuda(self.index._device).contiguous()
        idd = torch.as_tensor(ids, dtype=torch.int64).cuda(self.index._device).contiguous()
        assign = self.engine.assign(xd)
        own = (assign % self.world) == self.rank
        if bool(own.any()):
            self.index._store.add_batch(idd[own].contiguous(), xd[own].contiguous(), assign[own].contiguous())
            self.index._resident.update(idd[own].cpu().numpy())
        self.partitions.invalidate_sizes()
        return int(own.sum().item())

    def remove(self, ids):
        import torch
        h = np.ascontiguousarray(ids.cpu().numpy() if torch.is_tensor(ids) else ids, dtype=np.int64)
        n = int(self.index._store.remove_ids(h))
        self.index._resident.discard_present(h)
        self.partitions.invalidate_sizes()
        return n

    # -- maintenance -------------------------------------------------------------------------------------------------------------
    def initialize_maintenance_policy(self, params, cost_estimator=None):
        # hits recorded under the OLD policy belong to it: they reach it before it is replaced (the reference records hits inside
        # search, so a policy never sees queries answered before it existed)
        self._flush_hits()
        self.partitions.initialize_maintenance_policy(params, cost_estimator)

    def maintenance(self):
        self._flush_hits()
        return self.partitions.maintenance()

    def refine_partitions(self, partition_ids=None, iterations=0):
        self._flush_hits()  # (list sizes are about to change: pending hits are credited with the sizes they scanned)
        self.partitions.refine_partitions(partition_ids, iterations)

    def ntotal(self):
        return self.partitions.ntotal()

    def nlist(self):
        return self.partitions.nlist()

    def d(self):
        return self.partitions.d()
