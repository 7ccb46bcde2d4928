"""CPU: the drop-in boundary, checked with the reference's own callers.  oracle/dropin_check.sh compiles the
reference's ROS wrapper translation units (src/lib/ScanRegistration.cpp, LaserOdometry.cpp, LaserMapping.cpp,
TransformMaintenance.cpp) UNCHANGED, where they lie, against the adapter classes in place of the reference's Basic* headers and
links them against libloamx.so with no undefined symbol (ROS / tf / PCL headers are the inert stand-ins of oracle/ref_stubs).
Nothing runs here — the wrappers would need a GPU behind the library; the adapter's behaviour is tested in test_gpu_pipeline.py."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "lib", "LaserMapping.cpp")), reason="no /root/reference on this machine")


def test_reference_wrappers_compile_and_link_against_the_adapter():
    if not os.path.exists(os.path.join(ROOT, "loam_velodyne_amd", "libloamx.so")):
        pytest.skip("libloamx.so not built (run __graft_entry__.build())")
    r = subprocess.run(["bash", os.path.join(ROOT, "oracle", "dropin_check.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    so = os.path.join(ROOT, "oracle", "_ref", "libloam_dropin.so")
    assert os.path.exists(so)
    syms = subprocess.run(["nm", "-DC", "--defined-only", so], capture_output=True, text=True).stdout
    for want in ("loam::LaserMapping::process()", "loam::LaserOdometry::process()", "loam::ScanRegistration::handleIMUMessage",
                 "loam::TransformMaintenance::laserOdometryHandler", "loam::LaserMapping::imuHandler"):
        assert want in syms, want
    # the whole node graph over the product (the four wrappers + the swapped MultiScanRegistration unit + the test harness)
    nodes = os.path.join(ROOT, "oracle", "_ref", "libloam_nodes.so")
    assert os.path.exists(nodes)
    nsyms = subprocess.run(["nm", "-DC", "--defined-only", nodes], capture_output=True, text=True).stdout
    for want in ("loam::MultiScanRegistration::handleCloudMessage", "nodes_push_cloud", "loam::LaserMapping::process()"):
        assert want in nsyms, want
    assert "loamx_scanreg_process_raw" in subprocess.run(["nm", "-D", "--undefined-only", nodes], capture_output=True, text=True).stdout
    # and the library calls behind them are the C-ABI's, left undefined for libloamx.so to provide
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    for want in ("loamx_map_process", "loamx_odom_process", "loamx_scanreg_update_imu", "loamx_tm_associate_to_map", "loamx_map_update_imu"):
        assert want in und, want
