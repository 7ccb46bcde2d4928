"""CPU: the optional-dependency build (make NO_RCCL=1 NO_ROCTX=1) must keep compiling — the translation unit that holds the
RCCL exchanges is compiled with both switches and must still define every loamx_dist_* entry point include/loamx.h declares."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_dist_unit_compiles_without_rccl_and_roctx(tmp_path):
    obj = tmp_path / "api_dist.o"
    src = os.path.join(ROOT, "loam_velodyne_amd", "csrc", "api_dist.hip")
    cmd = [HIPCC, "-DLOAMX_NO_RCCL", "-DLOAMX_NO_ROCTX", "-O1", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Werror",
           "-Wno-unused-result", "-I/opt/rocm/include", "-c", src, "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    defined = set(re.findall(r" T (loamx_dist_\w+)", subprocess.run(["nm", "--defined-only", str(obj)], capture_output=True, text=True).stdout))
    header = open(os.path.join(ROOT, "include", "loamx.h")).read()
    declared = set(re.findall(r"\b(loamx_dist_\w+)\s*\(", header))
    assert declared and declared <= defined, sorted(declared - defined)
