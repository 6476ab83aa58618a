"""Builds libquake_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m quake_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the snapshot.  No CPU fallback exists: if the
library is missing the package fails loudly (quake_amd/_lib.py).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libquake_hip.so")
SOURCES = ["qk_ctx.hip", "qk_store.hip", "qk_scan.hip", "qk_scan_plan.hip", "qk_merge.hip", "qk_scan_rl.hip", "qk_small.hip", "qk_dense.hip", "qk_dense_pf.hip", "qk_dense_fused.hip", "qk_kmeans.hip", "qk_assign_pf.hip", "qk_aps.hip", "qk_api.hip", "qk_group.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value"]
# QK_BUILD_PROBES=1: the tuning build -- QK_* environment switches and the wave-clock printouts compiled in (qk_internal.h).
# The product build has none of them.
if os.environ.get("QK_BUILD_PROBES", "0") not in ("", "0"):
    FLAGS.append("-DQK_PROBES")
# QK_BUILD_DEV=1: development build -- the row-per-lane scan is instantiated for d = 128 only (never ship it)
if os.environ.get("QK_BUILD_DEV", "0") not in ("", "0"):
    FLAGS.append("-DQK_RL_DEV")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, variant=None, extra_flags=()):
    """variant: a side build (objects under build/<variant>, library lib/libquake_hip_<variant>.so, selected at run time with
    QUAKE_HIP_LIB) with extra_flags on top of the product flags -- the probe / development builds of scripts/."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build", variant) if variant else os.path.join(HERE, "build")
    lib = os.path.join(LIBDIR, f"libquake_hip_{variant}.so") if variant else LIB
    flags = FLAGS + list(extra_flags)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "quake_hip.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    # a change of flags (product <-> probe build) rebuilds everything
    flag_file = os.path.join(objdir, "flags.txt")
    flag_text = " ".join(flags)
    if not os.path.exists(flag_file) or open(flag_file).read() != flag_text:
        force = True
    objs = []
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose or res.returncode != 0:
                    sys.stderr.write(" ".join(cmd) + "\n" + res.stdout + res.stderr)
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed for " + cmd[-3])
    if jobs or force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("link failed")
        with open(flag_file, "w") as f:
            f.write(flag_text)
    return lib


if __name__ == "__main__":
    if "--probe" in sys.argv:  # tuning build: QK_* environment switches compiled in, row-per-lane scan for d = 128 only
        # (QK_VARIANT / QK_EXTRA_FLAGS: a second side build with other -D switches, for A/B runs in one box visit)
        print(build_lib(force="--force" in sys.argv, verbose=True, variant=os.environ.get("QK_VARIANT", "probe"),
                        extra_flags=["-DQK_PROBES", "-DQK_RL_DEV"] + os.environ.get("QK_EXTRA_FLAGS", "").split()))
    else:
        print(build_lib(force="--force" in sys.argv, verbose=True))
