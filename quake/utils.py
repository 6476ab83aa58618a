"""quake.utils -- the two helpers of the reference's src/python/utils.py the tests and harness use."""
import torch


def compute_recall(ids, gt_ids, k):
    """per-query |ids[:k] intersect gt_ids[:k]| / k (utils.py:162-177)."""
    from quake_amd.index import compute_recall as _cr
    return _cr(ids, gt_ids, k)


def knn(queries, vectors, k, metric="l2"):
    """exact neighbours by brute force (utils.py:194-229): (ids [nq, k], distances [nq, k])."""
    from quake_amd.workload import exact_knn
    if not torch.cuda.is_available():
        raise RuntimeError("quake.utils.knn runs on the GPU")
    return exact_knn(queries, vectors, k, metric)
