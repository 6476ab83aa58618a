"""ctypes binding of the C ABI (include/quake_hip.h).  There is NO CPU fallback: if libquake_hip.so is
missing this module raises, so a GPU box can never silently run something else."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QUAKE_HIP_LIB") or os.path.join(HERE, "lib", "libquake_hip.so")  # override: A/B builds

QK_OK = 0
QK_METRIC_IP = 0
QK_METRIC_L2 = 1
QK_MEM_HOST = 0
QK_MEM_DEVICE = 1
QK_MAX_K = 448
QK_MAX_NPROBE = 8192

STATUS_NAMES = {1: "QK_ERR_INVALID", 2: "QK_ERR_NOT_FOUND", 3: "QK_ERR_HIP", 4: "QK_ERR_UNSUPPORTED", 5: "QK_ERR_OOM"}


class QkTiming(C.Structure):
    _fields_ = [("coarse_ms", C.c_float), ("group_ms", C.c_float), ("scan_ms", C.c_float), ("merge_ms", C.c_float),
                ("total_ms", C.c_float), ("n_items", C.c_int64), ("scan_bytes", C.c_int64),
                ("partitions_scanned", C.c_int64)]


class QuakeHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int

# name -> (restype, argtypes); the single source the symbol test checks against include/quake_hip.h
SIGNATURES = {
    "qk_last_error": (C.c_char_p, []),
    "qk_version": (C.c_char_p, []),
    "qk_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "qk_ctx_destroy": (_int, [_vp]),
    "qk_ctx_set_stream": (_int, [_vp, _vp]),
    "qk_ctx_set_null_stream": (_int, [_vp]),
    "qk_ctx_get_stream": (_int, [_vp, C.POINTER(_vp), C.POINTER(_int)]),
    "qk_ctx_set_form_feedback": (_int, [_vp, _int]),
    "qk_ctx_set_form_times": (_int, [_vp, C.POINTER(C.c_float)]),
    "qk_ctx_synchronize": (_int, [_vp]),
    "qk_ctx_set_timing": (_int, [_vp, _int]),
    "qk_ctx_get_timing": (_int, [_vp, C.POINTER(_int)]),
    "qk_ctx_set_squared_l2": (_int, [_vp, _int]),
    "qk_ctx_read_timing": (_int, [_vp, C.POINTER(QkTiming), C.POINTER(_i64)]),
    "qk_ctx_device_info": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_i64), C.c_char_p, _int]),
    "qk_ctx_last_scan_kernel": (_int, [_vp, C.c_char_p, _int]),
    "qk_store_create": (_int, [_vp, _int, C.POINTER(_vp)]),
    "qk_store_destroy": (_int, [_vp]),
    "qk_store_reset": (_int, [_vp]),
    "qk_store_add_list": (_int, [_vp, _i64]),
    "qk_store_remove_list": (_int, [_vp, _i64]),
    "qk_store_add_entries": (_int, [_vp, _i64, _i64, _vp, _vp, _int]),
    "qk_store_add_batch": (_int, [_vp, _i64, _vp, _vp, _vp, _int]),
    "qk_store_build_csr": (_int, [_vp, _i64, _vp, _vp, _vp, _int]),
    "qk_store_remove_ids": (_int, [_vp, _i64, _vp, C.POINTER(_i64)]),
    "qk_store_list_size": (_int, [_vp, _i64, C.POINTER(_i64)]),
    "qk_store_list_sizes": (_int, [_vp, _vp, _i64, _vp]),
    "qk_store_get_lists": (_int, [_vp, _vp, _i64, _vp, _vp, _int]),
    "qk_store_publish": (_int, [_vp]),
    "qk_store_get_vectors": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "qk_store_ntotal": (_i64, [_vp]),
    "qk_store_nlist": (_i64, [_vp]),
    "qk_store_d": (_int, [_vp]),
    "qk_store_list_ids": (_int, [_vp, _vp, C.POINTER(_i64)]),
    "qk_store_get_list": (_int, [_vp, _i64, _vp, _vp, _int]),
    "qk_store_get_vector": (_int, [_vp, _i64, _vp, C.POINTER(_int)]),
    "qk_store_device_bytes": (_i64, [_vp]),
    "qk_store_counters": (_int, [_vp, _vp, _int]),
    "qk_coarse": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _int]),
    "qk_scan": (_int, [_vp, _vp, _vp, _i64, _vp, _int, _int, _int, _vp, _vp, _int, C.POINTER(QkTiming)]),
    "qk_search": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp, _int, C.POINTER(QkTiming)]),
    "qk_search_tracked": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp, _vp, _int, C.POINTER(QkTiming)]),
    "qk_search_aps": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, C.c_float, C.c_float, _int, C.c_float, _vp, _vp, _vp, _int,
                      C.POINTER(QkTiming)]),
    "qk_merge_topk": (_int, [_vp, _vp, _vp, _int, _i64, _int, _int, _vp, _vp]),
    "qk_topk_block_bytes": (C.c_size_t, [_i64, _int]),
    "qk_pack_topk": (_int, [_vp, _vp, _vp, _int, _i64, _int, _vp]),
    "qk_merge_topk_packed": (_int, [_vp, _vp, _int, _i64, _int, _int, _vp, _vp]),
    "qk_kmeans_assign": (_int, [_vp, _vp, _i64, _vp, _i64, _int, _int, _vp, _vp, _int]),
    "qk_kmeans_accumulate": (_int, [_vp, _vp, _i64, _int, _vp, _i64, _vp, _vp, _int]),
    "qk_kmeans_accumulate_blocked": (_int, [_vp, _vp, _i64, _int, _vp, _i64, _vp, _vp, _int]),
    "qk_store_refine_lists": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int]),
    "qk_kmeans": (_int, [_vp, _vp, _i64, _int, _i64, _int, _int, C.c_uint64, _vp, _vp, _int]),
    "qk_normalize_rows": (_int, [_vp, _vp, _i64, _int, _int]),
    "qk_kmeans_update": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _int]),
    "qk_kmeans_last_timing": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "qk_rand_perm": (_int, [_i64, _i64, C.c_uint64, _vp]),
    "qk_group_create": (_int, [C.POINTER(_int), _int, _int, C.POINTER(_vp)]),
    "qk_group_destroy": (_int, [_vp]),
    "qk_group_size": (_int, [_vp]),
    "qk_group_member": (_int, [_vp, _int, C.POINTER(_vp), C.POINTER(_vp)]),
    "qk_group_owner": (_int, [_vp, _i64]),
    "qk_group_set_stream": (_int, [_vp, _vp]),
    "qk_group_set_null_stream": (_int, [_vp]),
    "qk_group_get_stream": (_int, [_vp, C.POINTER(_vp), C.POINTER(_int)]),
    "qk_group_synchronize": (_int, [_vp]),
    "qk_group_set_form_feedback": (_int, [_vp, _int]),
    "qk_group_set_submit_threads": (_int, [_vp, _int]),
    "qk_group_reset": (_int, [_vp]),
    "qk_group_add_list": (_int, [_vp, _i64]),
    "qk_group_remove_list": (_int, [_vp, _i64]),
    "qk_group_add_entries": (_int, [_vp, _i64, _i64, _vp, _vp, _int]),
    "qk_group_add_batch": (_int, [_vp, _i64, _vp, _vp, _vp, _int]),
    "qk_group_build_csr": (_int, [_vp, _i64, _vp, _vp, _vp, _int]),
    "qk_group_remove_ids": (_int, [_vp, _i64, _vp, C.POINTER(_i64)]),
    "qk_group_list_size": (_int, [_vp, _i64, C.POINTER(_i64)]),
    "qk_group_list_sizes": (_int, [_vp, _vp, _i64, _vp]),
    "qk_group_get_lists": (_int, [_vp, _vp, _i64, _vp, _vp, _int]),
    "qk_group_ntotal": (_i64, [_vp]),
    "qk_group_nlist": (_i64, [_vp]),
    "qk_group_d": (_int, [_vp]),
    "qk_group_list_ids": (_int, [_vp, _vp, C.POINTER(_i64)]),
    "qk_group_get_list": (_int, [_vp, _i64, _vp, _vp, _int]),
    "qk_group_get_vector": (_int, [_vp, _i64, _vp, C.POINTER(_int)]),
    "qk_group_device_bytes": (_i64, [_vp]),
    "qk_group_refine_lists": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int]),
    "qk_group_scan": (_int, [_vp, _vp, _i64, _vp, _int, _int, _int, _vp, _vp, _int, C.POINTER(QkTiming)]),
    "qk_group_search": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp, _int, C.POINTER(QkTiming)]),
    "qk_group_search_aps": (_int, [_vp, _vp, _vp, _i64, _int, _int, C.c_float, C.c_float, _int, C.c_float, _vp, _vp, _vp, _int,
                            C.POINTER(QkTiming)]),
}

_lib = None


def load():
    """dlopen libquake_hip.so; raises if it has not been built (python -m quake_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (run `python -m quake_amd.build` or "
                "__graft_entry__.build()). quake_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64 and loads it by path; if ours binds the system copy first, two HIP
        # runtimes end up in one process and the second one finds no device.  Import torch first so that both sides
        # share the runtime that is already loaded (same SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status):
    if status != QK_OK:
        raise QuakeHipError(status, load().qk_last_error().decode("utf-8", "replace"))
