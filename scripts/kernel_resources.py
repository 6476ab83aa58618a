"""Register / spill / scratch / LDS report of every kernel in libquake_hip.so (or another fat binary / object file).

    python scripts/kernel_resources.py [path] [--filter substring] [--csv]

Walks the clang offload bundles inside the file (one per translation unit), takes the gfx950 code objects and reads the AMDGPU
metadata notes with llvm-readelf: .vgpr_count, .agpr_count, .sgpr_count, .vgpr_spill_count, .sgpr_spill_count,
.private_segment_fixed_size (scratch bytes per lane), .group_segment_fixed_size (static LDS).  What the round reviews quote
('k_scan_rl<8,true,true>: 509 VGPRs, 183 spilled SGPRs') comes from here."""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "/usr/bin/c++filt"


def code_objects(blob, arch="gfx950"):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if arch in triple and size:
                yield blob[i + off:i + off + size]
        pos = i + 24


def kernels_of(co):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(co)
        path = f.name
    try:
        txt = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
    finally:
        os.unlink(path)
    out = []
    for block in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        block = ".agpr_count:" + block
        d = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                    "group_segment_fixed_size", "max_flat_workgroup_size"):
            m = re.search(r"\.%s:\s+(\d+)" % key, block)
            d[key] = int(m.group(1)) if m else 0
        m = re.search(r"\.name:\s+(\S+)", block)
        d["name"] = m.group(1) if m else "?"
        out.append(d)
    return out


def main():
    argv = sys.argv[1:]
    flt = ""
    if "--filter" in argv:
        i = argv.index("--filter")
        flt = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = args[0] if args else os.path.join(here, "quake_amd", "lib", "libquake_hip.so")
    blob = open(path, "rb").read()
    rows = []
    for co in code_objects(blob):
        rows += kernels_of(co)
    names = subprocess.run([CXXFILT], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, nm in zip(rows, names):
        # (kernels of an anonymous namespace demangle to "(anonymous namespace)::name<..>(args)": cut the ARGUMENT list, not that)
        r["demangled"] = re.sub(r"^void ", "", nm).replace("(anonymous namespace)::", "").split("(")[0]
    rows = [r for r in rows if flt in r["demangled"]]
    rows.sort(key=lambda r: r["demangled"])
    sep = "," if "--csv" in sys.argv else "  "
    print(sep.join(["vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds", "kernel"]))
    for r in rows:
        print(sep.join(str(v) for v in (r["vgpr_count"], r["agpr_count"], r["sgpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"],
                                         r["private_segment_fixed_size"], r["group_segment_fixed_size"], r["demangled"])))


if __name__ == "__main__":
    main()
