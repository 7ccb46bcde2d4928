"""CPU: a dry run of the opt-in GPU tests' LOGIC.  tests/test_gpu_next.py and the C-ABI half of tests/test_gpu_nodes.py were written
without a GPU at hand; here their bodies run with the ctypes binding pointed (for these tests only) at the oracle-backed test double
of the C-ABI (oracle/mock_loamx_capi.cpp).  Device arithmetic is not exercised — the double IS the oracle — but a wrong method name,
argument order, state hand-over or expectation in those tests shows up now instead of on the GPU box."""
import ctypes as C
import os

import pytest

import oracle_py as op
from loam_velodyne_amd import loamx, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "oracle", "_ref", "libloamx_oracle_mock.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(MOCK) and op.RefNodes.available()), reason="oracle/_ref test double / node graph not built")


@pytest.fixture
def over_the_mock(monkeypatch):
    L = C.CDLL(MOCK)
    L.loamx_last_error.restype = C.c_char_p
    for n in ("loamx_scanreg_create", "loamx_odom_create", "loamx_map_create", "loamx_tm_create"):
        getattr(L, n).restype = C.c_void_p
    monkeypatch.setattr(loamx, "_lib", L)


def test_next_round_cases(orc, small_world, over_the_mock):
    import test_gpu_next as T
    T.test_degenerate_corridor_odometry_and_registration(orc)
    T.test_odometry_too_few_rows(orc, small_world)
    T.test_mapping_too_few_rows(orc, small_world)


def test_c_abi_node_graph(small_world, over_the_mock):
    import test_gpu_nodes as N
    N.test_c_abi_composition_vs_the_reference_nodes(small_world)
