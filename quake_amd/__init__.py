"""quake_amd -- MI355X-native implementation of Quake's search / k-means hot path.

Drop-in surface (same names as the reference's `quake` package, src/python/__init__.py:1-8 -> quake._bindings):

    import quake_amd as quake
    index = quake.QuakeIndex(); index.build(x, ids, params); index.search(q, search_params)

Everything computes in libquake_hip.so (hand-written HIP for gfx950, quake_amd/csrc/); there is no CPU fallback.
"""
from .index import (BuildTimingInfo, IndexBuildParams, MaintenancePolicyParams, MaintenanceTimingInfo,  # noqa: F401
                    ModifyTimingInfo, QuakeHipError, QuakeIndex, SearchParams, SearchResult, SearchTimingInfo,
                    compute_recall)
