"""Cluster-sharded search across the GPUs of one node.

One process per GPU.  Partitions (lists) are sharded by list number -- the multi-GPU analogue of the reference's
partition -> core map (PartitionManager::distribute_partitions, partition_manager.cpp:599-602) -- the centroids are
replicated, every rank runs the coarse step for the whole batch and scans only the probed lists it owns (worker_scan's
per-core jobs, query_coordinator.cpp:243-469), then the per-rank top-k are exchanged with ONE all-gather of ids and
merge keys (the cross-worker batch_add, query_coordinator.cpp:167-173,231-235) and merged under the same (key, id)
total order.  No other collective is on the search path.

The arithmetic lives in an *engine*:
  GpuEngine   -- libquake_hip.so (qk_search with squared-L2 keys + qk_merge_topk), the product path
  any object with .search_local(q, nprobe, k) -> (ids, keys) and .merge(ids[G,Q,k], keys[G,Q,k]) -> (ids, dist)
so the orchestration can be exercised with world_size-2 gloo tests on CPU (tests/test_sharded_gloo.py), where the
test injects an oracle-backed engine.
"""
import numpy as np


def owner_of_list(list_no, world, lists_per_rank=None):
    """block layout when lists_per_rank is given (rank r owns [r*L, (r+1)*L)), else list_no % world
    (partition_manager.cpp:599-602: partition i -> core i % num_workers)."""
    if lists_per_rank:
        return int(list_no) // int(lists_per_rank)
    return int(list_no) % int(world)


def shard_offsets(offsets, rank, world, lists_per_rank=None):
    """CSR of a GLOBAL index -> (local_offsets [nlist+1], row_selector) keeping every list number but emptying the lists
    this rank does not own (an empty list is skipped by the scan, like the reference skips empty partitions)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    nlist = offsets.shape[0] - 1
    sizes = np.diff(offsets)
    own = np.array([owner_of_list(p, world, lists_per_rank) == rank for p in range(nlist)], dtype=bool)
    local_sizes = np.where(own, sizes, 0)
    local_offsets = np.zeros(nlist + 1, np.int64)
    local_offsets[1:] = np.cumsum(local_sizes)
    rows = np.concatenate([np.arange(offsets[p], offsets[p + 1]) for p in range(nlist) if own[p]]) if own.any() else np.zeros(0, np.int64)
    return local_offsets, rows.astype(np.int64)


class GpuEngine:
    """Per-rank engine on libquake_hip.so."""

    def __init__(self, ctx, parent, store, metric):
        self.ctx, self.parent, self.store, self.metric = ctx, parent, store, metric
        ctx.set_squared_l2(True)  # ranks exchange the merge key; sqrt happens after the merge

    def search_local(self, q, nprobe, k, out=None):
        return self.ctx.search(self.parent, self.store, q, nprobe, k, self.metric, out=out)

    def merge(self, ids, keys):
        return self.ctx.merge_topk(ids, keys, self.metric)


class ShardedIndex:
    """search() = local scan + all-gather + merge.  `dist` is torch.distributed (nccl = RCCL on ROCm, or gloo)."""

    def __init__(self, engine, dist=None, world=1, rank=0):
        self.engine, self.dist, self.world, self.rank = engine, dist, int(world), int(rank)
        self._g_ids = self._g_keys = None

    def search(self, q, nprobe, k, out=None):
        import torch
        ids, keys = self.engine.search_local(q, nprobe, k, out=out)
        if self.world == 1 or self.dist is None:
            return self.engine.merge(ids.reshape((1,) + tuple(ids.shape)), keys.reshape((1,) + tuple(keys.shape)))
        t_ids = ids if torch.is_tensor(ids) else torch.from_numpy(np.ascontiguousarray(ids))
        t_keys = keys if torch.is_tensor(keys) else torch.from_numpy(np.ascontiguousarray(keys))
        shape = (self.world,) + tuple(t_ids.shape)
        if self._g_ids is None or tuple(self._g_ids.shape) != shape or self._g_ids.device != t_ids.device:
            self._g_ids = torch.empty(shape, dtype=torch.int64, device=t_ids.device)
            self._g_keys = torch.empty(shape, dtype=torch.float32, device=t_ids.device)
        # concatenation form ([world*Q, k]) is accepted by both the RCCL and the gloo backend
        self.dist.all_gather_into_tensor(self._g_ids.view(-1, t_ids.shape[-1]), t_ids.contiguous())
        self.dist.all_gather_into_tensor(self._g_keys.view(-1, t_keys.shape[-1]), t_keys.contiguous())
        return self.engine.merge(self._g_ids, self._g_keys)
