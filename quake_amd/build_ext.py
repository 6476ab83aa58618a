"""Builds quake_amd/_bindings.so: the C++ host mirror (quake_amd/cpp/) + pybind11 module, linked against the in-tree
libquake_hip.so.  `python -m quake_amd.build_ext`.  Uses torch.utils.cpp_extension (host C++ only: no device code here; the HIP runtime
headers come in for c10::hip::getCurrentHIPStream alone)."""
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CPP = os.path.join(HERE, "cpp")
OUT = os.path.join(HERE, "_bindings.so")


def build_bindings(force=False, verbose=False):
    from .build import build_lib
    lib = build_lib()
    srcs = [os.path.join(CPP, f) for f in ("bindings.cpp", "quake_index.cpp", "partition_manager.cpp", "query_coordinator.cpp",
                                           "maintenance_policies.cpp", "list_scanning.cpp")]
    deps = srcs + glob.glob(os.path.join(CPP, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "quake_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    from torch.utils import cpp_extension
    bdir = os.path.join(HERE, "build", "ext")
    os.makedirs(bdir, exist_ok=True)
    libdir = os.path.dirname(lib)
    cpp_extension.load(
        name="_bindings", sources=srcs, build_directory=bdir, verbose=verbose, is_python_module=False,
        extra_cflags=["-O2", "-std=c++17"],
        extra_include_paths=["/opt/rocm/include"],  # (c10/hip/HIPStream.h: torch's current stream)
        extra_ldflags=[f"-L{libdir}", "-lquake_hip", "-lc10_hip", "-Wl,-rpath,$ORIGIN/lib", f"-Wl,-rpath,{libdir}"],
        with_cuda=False)
    built = glob.glob(os.path.join(bdir, "_bindings*.so"))
    if not built:
        raise RuntimeError("extension build produced no .so")
    shutil.copy(built[0], OUT)
    return OUT


if __name__ == "__main__":
    print(build_bindings(force="--force" in sys.argv, verbose=True))
