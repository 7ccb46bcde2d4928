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


def collinear_previous_surf(f1):
    """a previous less-flat cloud whose points all lie on ONE straight line through the current flat features (dyadic coordinates: the
    differences, and with them every tripod's cross product, are exact zeros) — 24 points on each of the 16 rings, no two at the same
    place, so no neighbour search meets a tie"""
    fl = f1["flat"]
    c0 = np.round(fl[len(fl) // 2, :3].astype(np.float64) * 4) / 4
    rows = [[c0[0] + 0.5 * t, c0[1], c0[2] + 0.25 * t, r + 0.01 * k]
            for r in range(16) for k in range(24) for t in [(-768 + 64 * k + r) / 256.0]]   # (distinct for every (ring, k))
    line = np.array(rows, np.float32)
    assert np.array_equal(line[:, :3].astype(np.float64), np.array(rows)[:, :3]) and len(np.unique(line[:, :3], axis=0)) == len(line)
    return line
