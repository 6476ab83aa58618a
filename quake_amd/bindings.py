"""Loader of the compiled surface: `from quake_amd.bindings import QuakeIndex, SearchParams, ...` gives the pybind11
classes of quake_amd/_bindings.so (C++ host mirror, quake_amd/cpp/), the counterpart of `quake._bindings`."""
import os

from . import _lib

_lib.load()  # libquake_hip.so first (also makes the extension independent of its rpath)
import torch  # noqa: E402,F401  (libtorch must be loaded before the extension)

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bindings.so")
if not os.path.exists(_SO):
    raise ImportError(f"{_SO} is missing: run `python -m quake_amd.build_ext` (or __graft_entry__.build())")

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("quake_amd._bindings", _SO)
_bindings = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bindings)

# every public name of the extension module: the classes (QuakeIndex, IndexBuildParams, SearchParams, ..., PartitionManager,
# QueryCoordinator) and the list_scanning seam (batched_scan_list)
for _name in dir(_bindings):
    if not _name.startswith("_"):
        globals()[_name] = getattr(_bindings, _name)
