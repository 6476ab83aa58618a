"""Generates tests/golden/utils_knn.npz  -- run ONLY in the authoring container.

It imports the reference's pure-python helpers (src/python/utils.py: knn, compute_recall,
compute_distance) from /root/reference and records their outputs on seeded inputs.  The .npz holds
data only (inputs + expected outputs); /root/reference never travels to the GPU box.

    python tests/golden/make_golden.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/src/python/utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils_knn.npz")


def main():
    spec = importlib.util.spec_from_file_location("quake_ref_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rng = np.random.default_rng(20250523)
    nq, n, d, k = 24, 3000, 64, 10
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    vectors = rng.standard_normal((n, d)).astype(np.float32)
    out = {"queries": queries, "vectors": vectors, "k": np.int64(k)}
    for metric in ("l2", "ip"):
        idx, val = ref.knn(torch.from_numpy(queries), torch.from_numpy(vectors), k, metric)
        out[f"knn_{metric}_ids"] = idx.numpy().astype(np.int64)
        out[f"knn_{metric}_dist"] = val.numpy().astype(np.float32)
        # float64 re-check: smallest gap between consecutive true distances in the top-(k+1); tests use it to
        # demand id-exactness only where the ranks are separated by more than fp32 noise
        q64, v64 = queries.astype(np.float64), vectors.astype(np.float64)
        if metric == "l2":
            dm = np.sqrt(((q64[:, None, :] - v64[None, :, :]) ** 2).sum(-1))
            srt = np.sort(dm, axis=1)[:, : k + 1]
        else:
            dm = q64 @ v64.T
            srt = -np.sort(-dm, axis=1)[:, : k + 1]
        out[f"knn_{metric}_mingap"] = np.abs(np.diff(srt, axis=1)).min(axis=1)
    # compute_recall(ids, gt, k): perturb the L2 result so recall is not trivially 1
    ids = out["knn_l2_ids"].copy()
    ids[::2, 3] = -7
    ids[1::3, 0] = ids[1::3, 1]
    rec = ref.compute_recall(torch.from_numpy(ids), torch.from_numpy(out["knn_l2_ids"]), k)
    out["recall_ids"] = ids
    out["recall_expected"] = rec.numpy().astype(np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k_: getattr(v, "shape", None) for k_, v in out.items()})


if __name__ == "__main__":
    main()
