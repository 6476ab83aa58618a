"""`import quake` -- the reference's package name (src/python/__init__.py:1-8) for the MI355X-native implementation:
`quake.QuakeIndex`, `quake.SearchParams`, ... are the pybind11 classes of quake_amd/_bindings.so (the compiled C++ host
mirror over libquake_hip.so), and `quake._bindings` is that module, like upstream's.  Code written against the reference
(`import quake; idx = quake.QuakeIndex(); idx.build(x, ids, params)`) runs unchanged."""
import sys as _sys

try:
    import torch  # noqa: F401
    from quake_amd import bindings as _loader

    _bindings = _loader._bindings
    _sys.modules[__name__ + "._bindings"] = _bindings
    from quake_amd.bindings import (BuildTimingInfo, IndexBuildParams, MaintenancePolicyParams, MaintenanceTimingInfo,  # noqa: F401
                                    ModifyTimingInfo, QuakeIndex, SearchParams, SearchResult, SearchTimingInfo)
except (ImportError, ModuleNotFoundError) as e:  # the reference prints and carries on (src/python/__init__.py:5-8)
    print(e)
    print("Bindings not installed")
