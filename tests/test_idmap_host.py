"""The store's host-side id -> list map (quake_amd/csrc/qk_idmap.h) against std::unordered_map: g++ build of
tests/native/idmap_check.cpp, no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_idmap_matches_unordered_map(tmp_path):
    exe = str(tmp_path / "idmap_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "quake_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "idmap_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
