"""The C-ABI library must load without a GPU and export every symbol include/quake_hip.h declares
(no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "quake_hip.h")).read()
    return sorted(set(re.findall(r"QK_API\s+[\w\s\*]+?\b(qk_\w+)\s*\(", txt)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ("qk_ctx_create", "qk_store_build_csr", "qk_store_add_entries", "qk_store_remove_ids", "qk_coarse",
                 "qk_scan", "qk_search", "qk_merge_topk", "qk_kmeans_assign", "qk_kmeans_accumulate", "qk_kmeans"):
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    from quake_amd import _lib
    from quake_amd.build import build_lib
    build_lib()
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/quake_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in quake_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"
    assert lib.qk_version().startswith(b"quake_hip")


def test_no_gpu_calls_fail_cleanly():
    """Without a GPU qk_ctx_create must return an error code + message, not crash."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from quake_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    st = lib.qk_ctx_create(0, C.byref(ctx))
    assert st != 0
    assert len(lib.qk_last_error()) > 0


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under quake_amd/ may reference it."""
    pkg = os.path.join(ROOT, "quake_amd")
    for dp, _, files in os.walk(pkg):
        if os.path.basename(dp) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), os.path.join(dp, f)
                assert "libquake_oracle" not in txt, os.path.join(dp, f)
