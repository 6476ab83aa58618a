"""QuakeWrapper: the keyword-argument front end the reference's harness drives (src/python/index_wrappers/quake.py:11-213,
interface src/python/index_wrappers/wrapper.py), over quake_amd.QuakeIndex."""
import torch

from . import index as _index


class QuakeWrapper:
    def __init__(self, device=0):
        self.index = None
        self.index_type = None
        self.assignments = None
        self._device = device

    def n_total(self):
        return self.index.ntotal()

    def d(self):
        return self.index.d()

    def index_state(self):  # quake.py:35-47
        return {"n_list": self.index.nlist(), "n_total": self.index.ntotal()}

    def build(self, vectors, nc, metric="l2", ids=None, num_workers=0, m=-1, code_size=8):
        assert vectors.ndim == 2
        assert nc > 0
        bp = _index.IndexBuildParams()
        bp.metric = metric.lower()
        bp.nlist = nc
        bp.num_workers = num_workers
        self.index = _index.QuakeIndex(device=self._device)
        if ids is None:
            ids = torch.arange(vectors.shape[0], dtype=torch.int64)
        return self.index.build(vectors, ids.to(torch.int64), bp)

    def add(self, vectors, ids=None, num_threads=0):
        assert self.index is not None
        assert vectors.ndim == 2
        if ids is None:
            curr = self.n_total()
            ids = torch.arange(curr, curr + vectors.shape[0], dtype=torch.int64)
        return self.index.add(vectors, ids)

    def remove(self, ids):
        assert self.index is not None
        assert ids.ndim == 1
        return self.index.remove(ids)

    def search(self, query, k, nprobe=1, batched_scan=False, recall_target=-1, k_factor=4.0, use_precomputed=True,
               initial_search_fraction=0.05, recompute_threshold=0.1, aps_flush_period_us=50, n_threads=1):
        sp = _index.SearchParams()
        sp.nprobe = nprobe
        sp.recall_target = recall_target
        sp.use_precomputed = use_precomputed
        sp.batched_scan = batched_scan
        sp.initial_search_fraction = initial_search_fraction
        sp.recompute_threshold = recompute_threshold
        sp.aps_flush_period_us = aps_flush_period_us
        sp.k = k
        sp.num_threads = n_threads
        return self.index.search(query, sp)

    def maintenance(self):
        return self.index.maintenance()

    def save(self, filename):
        self.index.save(str(filename))

    def load(self, filename, n_workers=0, use_numa=False, verbose=False, verify_numa=False, same_core=True,
             use_centroid_workers=False, use_adaptive_n_probe=False):
        self.index = _index.QuakeIndex(device=self._device)
        self.index.load(str(filename), n_workers)

    def centroids(self):
        cids = self.index.parent.get_ids()
        return self.index.parent.get(cids)

    def metric(self):
        return "l2" if self.index.metric_ == 1 else "ip"
