"""CPU: the C-ABI library loads and exports every symbol include/loamx.h declares; without a GPU every constructor
fails loudly (LOAMX_E_NOGPU) — there is no CPU fallback to fall into."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "loamx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(loamx_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol():
    from loam_velodyne_amd import loamx
    L = loamx.lib()
    names = _declared()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/loamx.h but not exported: {missing}"
    assert L.loamx_abi_version() == 6
    L.loamx_build_info.restype = __import__("ctypes").c_char_p
    info = dict(kv.split("=") for kv in L.loamx_build_info().decode().split(";"))
    assert info["abi"] == "6" and info["diag"] == "0", info   # the shipped library is not a diagnostic build


def test_no_oracle_in_product():
    """The product library and package must not link, import or execute anything under oracle/."""
    lib = os.path.join(ROOT, "loam_velodyne_amd", "libloamx.so")
    out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "loam_velodyne_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and '#include "oracle' not in txt, fn
                assert "oracle_mock" not in txt and "mock_loamx" not in txt, fn     # the C-ABI test double belongs to tests only


def test_header_is_plain_c():
    """include/loamx.h is a C header (extern "C", plain pointers and sizes): it must compile as C99 and as C++11 on its own"""
    hdr = os.path.join(ROOT, "include", "loamx.h")
    for cmd in (["gcc", "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], ["g++", "-std=c++11", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_fails_loudly_without_gpu():
    from loam_velodyne_amd import loamx
    if loamx.device_count() > 0:
        pytest.skip("a GPU is visible")
    L = loamx.lib()
    for name in ("loamx_scanreg_create", "loamx_odom_create", "loamx_map_create"):
        fn = getattr(L, name)
        fn.restype = C.c_void_p
        assert fn(None) is None
        assert b"no HIP device" in L.loamx_last_error()
    L.loamx_batch_create.restype = C.c_void_p
    assert L.loamx_batch_create(None, 4) is None
    L.loamx_pipeline_create.restype = C.c_void_p
    assert L.loamx_pipeline_create(None, None, None, 2) is None
    with pytest.raises(loamx.LoamxError):
        loamx.Batch(2)


def test_default_configs_match_reference_defaults():
    from loam_velodyne_amd import loamx
    L = loamx.lib()
    f, o, m = loamx.ScanRegConfig(), loamx.OdomConfig(), loamx.MapConfig()
    L.loamx_scanreg_default_config(C.byref(f))
    L.loamx_odom_default_config(C.byref(o))
    L.loamx_map_default_config(C.byref(m))
    # BasicScanRegistration.h:37-44, BasicLaserOdometry.cpp:20-26, BasicLaserMapping.cpp:51-59, :98-99
    assert (f.n_feature_regions, f.curvature_region, f.max_corner_sharp, f.max_surface_flat) == (6, 5, 2, 4)
    assert abs(f.scan_period - 0.1) < 1e-7 and abs(f.less_flat_filter_size - 0.2) < 1e-7 and abs(f.surface_curvature_threshold - 0.1) < 1e-7
    assert o.max_iterations == 25 and abs(o.delta_t_abort - 0.1) < 1e-7 and abs(o.delta_r_abort - 0.1) < 1e-7
    assert m.max_iterations == 10 and abs(m.delta_t_abort - 0.05) < 1e-7 and abs(m.corner_filter_size - 0.2) < 1e-7
    assert abs(m.surf_filter_size - 0.4) < 1e-7
