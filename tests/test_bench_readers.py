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


@pytest.mark.gpu
def test_bench_runs_on_a_dataset_directory(tmp_path):
    """`bench.py --sift-dir DIR`: the headline workload on the files of a TEXMEX-layout directory (here a 200k x 128 integer-valued
    stand-in written by this test), end to end on the GPU: one JSON line, `data` says where the vectors came from, recall at the
    target, parity against the oracle enforced inside the run (bench.py exits non-zero otherwise)."""
    import json
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(3)
    cent = rng.integers(0, 200, (256, 128)).astype(np.float32)
    base = np.clip(cent[rng.integers(0, 256, 200000)] + rng.normal(0, 12, (200000, 128)), 0, 255).round().astype(np.float32)
    query = np.clip(cent[rng.integers(0, 256, 2048)] + rng.normal(0, 12, (2048, 128)), 0, 255).round().astype(np.float32)
    _write_vecs(tmp_path / "sift_base.fvecs", base)
    _write_vecs(tmp_path / "sift_query.fvecs", query)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--sift-dir", str(tmp_path), "--nlist", "256", "--batch", "512",
                        "--steps", "5", "--warmup", "2", "--settle", "5", "--no-extra", "--no-pmc", "--cpu-seconds", "1", "--inflight", "1"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["data"] == "dataset files (--sift-dir)" and line["config"]["nvec"] == 200000 and line["config"]["dim"] == 128
    assert line["config"]["recall_at_k"] >= 0.9 and line["value"] > 0 and line["cpu_baseline"]["ids_equal_to_gpu_frac"] == 1.0
