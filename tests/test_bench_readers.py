"""bench.py --sift-dir: the TEXMEX .fvecs / .ivecs layout (int32 dimension in front of every vector; the reference reads the same
files in src/python/utils.py and datasets/ann_datasets.py:44) read back exactly, and malformed files refused."""
import numpy as np
import pytest

import bench as B


def _write_vecs(path, a):
    a = np.ascontiguousarray(a)
    rec = np.empty((a.shape[0], 4 + a.shape[1] * a.dtype.itemsize), np.uint8)
    rec[:, :4] = np.array([a.shape[1]], np.int32).view(np.uint8)
    rec[:, 4:] = a.view(np.uint8).reshape(a.shape[0], -1)
    rec.tofile(path)


def test_vecs_round_trip_and_directory(tmp_path):
    rng = np.random.default_rng(0)
    base = rng.standard_normal((37, 128)).astype(np.float32)
    query = rng.standard_normal((5, 128)).astype(np.float32)
    gt = rng.integers(0, 37, (5, 100)).astype(np.int32)
    _write_vecs(tmp_path / "sift_base.fvecs", base)
    _write_vecs(tmp_path / "sift_query.fvecs", query)
    _write_vecs(tmp_path / "sift_groundtruth.ivecs", gt)
    b, q, g = B.load_sift_dir(str(tmp_path))
    assert b.dtype == np.float32 and np.array_equal(b.view(np.uint32), base.view(np.uint32))
    assert np.array_equal(q, query) and g.dtype == np.int32 and np.array_equal(g, gt)
    (tmp_path / "sift_groundtruth.ivecs").unlink()
    assert B.load_sift_dir(str(tmp_path))[2] is None  # the ground-truth file is optional (the run computes its own)


def test_malformed_vecs_are_refused(tmp_path):
    p = tmp_path / "x_base.fvecs"
    np.array([128, 1, 2, 3], np.int32).tofile(p)  # says d = 128, holds 3 components
    with pytest.raises(SystemExit):
        B.read_vecs(str(p), np.float32)
    p.write_bytes(b"")
    with pytest.raises(SystemExit):
        B.read_vecs(str(p), np.float32)
