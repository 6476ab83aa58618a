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


def test_compiled_surface_builds_and_exposes_reference_names():
    """quake_amd/_bindings.so (C++ host mirror + pybind11) must build on CPU and expose the class / attribute names of
    the reference's quake._bindings (src/cpp/bindings/wrap.cpp:57-368).  No device call is made."""
    from quake_amd.build_ext import build_bindings
    build_bindings()
    import quake_amd.bindings as b
    for cls in ("QuakeIndex", "IndexBuildParams", "SearchParams", "MaintenancePolicyParams", "SearchResult",
                "SearchTimingInfo", "BuildTimingInfo", "ModifyTimingInfo", "MaintenanceTimingInfo"):
        assert hasattr(b, cls), cls
    for m in ("build", "search", "get", "get_ids", "add", "remove", "maintenance", "initialize_maintenance_policy",
              "save", "load", "ntotal", "nlist", "parent", "current_level"):
        assert hasattr(b.QuakeIndex, m), m
    sp = b.SearchParams()
    for a, v in (("k", 1), ("nprobe", 1), ("batched_scan", False), ("num_threads", 1), ("use_precomputed", True)):
        assert getattr(sp, a) == v
    assert abs(sp.recall_target + 1.0) < 1e-6 and abs(sp.initial_search_fraction - 0.02) < 1e-6
    bp = b.IndexBuildParams()
    assert (bp.nlist, bp.niter, bp.metric, bp.num_workers) == (0, 5, "l2", 0)
    mp = b.MaintenancePolicyParams()
    assert (mp.window_size, mp.refinement_radius, mp.refinement_iterations, mp.min_partition_size) == (1000, 25, 3, 32)
    idx = b.QuakeIndex()
    assert idx.ntotal() == 0 and idx.nlist() == 0 and idx.parent is None and idx.current_level == 0


def test_quake_package_alias():
    """`import quake` (the reference's package name, src/python/__init__.py) gives the compiled surface; `quake._bindings` is
    importable like upstream's extension module (wrap.cpp:48)."""
    from quake_amd.build_ext import build_bindings
    build_bindings()
    import quake
    from quake import _bindings
    import quake._bindings as qb2
    assert _bindings is qb2 and quake.QuakeIndex is _bindings.QuakeIndex
    idx = quake.QuakeIndex()
    assert idx.parent is None and idx.partition_manager is None and idx.query_coordinator is None and idx.ntotal() == 0
    sp, bp = quake.SearchParams(), quake.IndexBuildParams()
    assert (sp.k, sp.nprobe, sp.batched_scan, bp.nlist, bp.niter, bp.metric) == (1, 1, False, 0, 5, "l2")
    for name in ("PartitionManager", "QueryCoordinator", "batched_scan_list", "MaintenancePolicyParams", "SearchResult"):
        assert hasattr(_bindings, name), name


def test_summaries_of_both_mirrors_agree():
    """the JSON-style __repr__ of every bound value class (wrap.cpp:122-350): the compiled mirror and the Python mirror print the
    same line, and what must be JSON is JSON (two of the reference's summaries end in ", }" -- reproduced, not parsed)."""
    import json
    import quake._bindings as qb
    import quake_amd as qa
    for name in ("IndexBuildParams", "SearchParams", "SearchTimingInfo", "MaintenancePolicyParams"):
        a, b = repr(getattr(qb, name)()), repr(getattr(qa, name)())
        assert a == b, (name, a, b)
        if not a.endswith(", }"):
            json.loads(a)
    assert repr(qb.MaintenancePolicyParams()).endswith('"split_threshold_ns": 10, }')
    t = qb.SearchTimingInfo()
    t.parent_info = qb.SearchTimingInfo()
    assert '"parent_scan_time_ns": 0' in repr(t)
    bi = qa.BuildTimingInfo()
    assert bi.code_size == -1 and bi.n_codebooks == -1 and json.loads(repr(bi))["n_codebooks"] == -1
    assert json.loads(repr(qa.ModifyTimingInfo()))["modify_count"] == 0 and "n_splits" in json.loads(repr(qa.MaintenanceTimingInfo()))
    for cls in ("BuildTimingInfo", "ModifyTimingInfo", "MaintenanceTimingInfo"):
        assert hasattr(getattr(qb, cls), "__repr__")
    assert hasattr(qb.BuildTimingInfo, "code_size") and hasattr(qb.BuildTimingInfo, "n_codebooks")


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md is the reference-side binding story: every symbol of the C ABI has a row saying which reference interface
    it replaces (or that it has none)"""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in declared_symbols() if s not in txt]
    assert not missing, missing
