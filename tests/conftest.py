import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# poses must match the reference CPU path within 1e-4 m / 1e-4 rad (BASELINE.json north_star)
POSE_TOL = 1e-4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the checker and the library exist (no-ops when already built; hipcc cross-compiles without a GPU)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    if not os.path.exists(os.path.join(ROOT, "loam_velodyne_amd", "libloamx.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "loam_velodyne_amd", "csrc"), "-j8", "-s"], check=True)


@pytest.fixture(scope="session")
def orc():
    import oracle_py
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def small_world():
    from loam_velodyne_amd import synth
    return synth.World(half_extent=45.0)


def sorted_rows(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])]
