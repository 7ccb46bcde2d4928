"""CPU: pose fusion after the path (BasicTransformMaintenance) and the nodes' wire convention.
Host arithmetic only — libloamx needs no device for these entry points."""
import numpy as np

import oracle_py as op
from loam_velodyne_amd import loamx, synth


def R_of(p):
    return synth.rot_zxy(p[0], p[1], p[2])


def test_associate_to_map_matches_oracle_bitwise(orc):
    rng = np.random.default_rng(7)
    tm = loamx.TransformMaintenance()
    for _ in range(200):
        s, b, a = (np.concatenate([rng.uniform(-0.6, 0.6, 3), rng.uniform(-50, 50, 3)]).astype(np.float32) for _ in range(3))
        tm.update_odometry(s)
        tm.update_mapping_transform(a, b)
        assert np.array_equal(tm.associate_to_map(), op.tm_associate(orc, s, b, a))


def test_associate_to_map_is_the_pose_composition(orc):
    """transformMapped = aftMapped o befMapped^-1 o transformSum (rigid transforms, R = Ry Rx Rz): checked in float64."""
    rng = np.random.default_rng(8)
    tm = loamx.TransformMaintenance()
    for _ in range(100):
        s, b, a = (np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-20, 20, 3)]).astype(np.float32) for _ in range(3))
        tm.update_odometry(s)
        tm.update_mapping_transform(a, b)
        m = tm.associate_to_map()
        Rm = R_of(a) @ R_of(b).T @ R_of(s)
        tmap = a[3:] - Rm @ (R_of(s).T @ (b[3:] - s[3:]).astype(np.float64))
        assert np.abs(R_of(m) - Rm).max() < 2e-6
        assert np.abs(m[3:] - tmap).max() < 2e-4
    # no correction yet (bef == sum): the mapped pose is the mapping result itself
    s = np.array([0.01, 0.3, -0.02, 1, 2, 3], np.float32)
    a = np.array([0.02, 0.31, -0.01, 1.1, 2.0, 3.2], np.float32)
    tm.update_odometry(s)
    tm.update_mapping_transform(a, s)
    assert np.abs(tm.associate_to_map() - a).max() < 1e-6


def test_wire_convention_round_trip_and_oracle(orc):
    rng = np.random.default_rng(9)
    for _ in range(200):
        r = rng.uniform(-1.2, 1.2, 3).astype(np.float32)
        q = loamx.wire_pose_to_quat(r)
        assert np.array_equal(q, op.wire_pose_to_quat(orc, r))
        assert abs(np.linalg.norm(q) - 1) < 1e-12
        back = loamx.wire_quat_to_pose(q)
        assert np.array_equal(back, op.wire_quat_to_pose(orc, q))
        assert np.abs(back - r).max() < 1e-6
    # identity pose <-> identity quaternion
    assert np.allclose(loamx.wire_pose_to_quat(np.zeros(3, np.float32)), [0, 0, 0, 1])
