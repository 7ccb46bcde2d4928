"""GPU: BasicLaserOdometry replacement vs the oracle on identical feature clouds (tolerance 1e-4 m / 1e-4 rad)."""
import numpy as np
import pytest

import oracle_py as op
from conftest import POSE_TOL
from loam_velodyne_amd import loamx, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sensor,n", [("VLP-16", 6), ("HDL-64E", 3)])
def test_sequence_parity(orc, small_world, sensor, n):
    poses = synth.trajectory(n)
    osr, ood, god = op.ScanRegistration(orc), op.LaserOdometry(orc), loamx.LaserOdometry()
    for k in range(n):
        sw = synth.make_sweep(small_world, sensor, poses[k], poses[k + 1], seed=k)
        f = osr.process(sw.points, sw.ring_sizes)
        ood.set_features(f)
        ood.process()
        rc = god.process(f)
        assert rc == (loamx.SKIPPED if k == 0 else loamx.OK)     # the first call only initialises (:198-211)
        assert np.abs(ood.transform - god.transform).max() < POSE_TOL
        assert np.abs(ood.transform_sum - god.transform_sum).max() < POSE_TOL
        oc, os_ = ood.last_corner(), ood.last_surf()
        gc, gs = god.last_clouds()
        assert oc.shape == gc.shape and os_.shape == gs.shape
        assert np.abs(oc - gc).max() < 1e-4 and np.abs(os_ - gs).max() < 1e-4
        assert np.array_equal(oc[:, 3], gc[:, 3])                # transformToEnd truncates intensity to the ring id (:70)
        st_o, st_g = ood.stats(), god.stats()
        assert st_o["iterations"] == st_g["iterations"] and st_o["sel"] == st_g["sel"]
        fe_o, fe_g = ood.full_to_end(), god.transform_to_end(f["full"])
        assert np.abs(fe_o - fe_g).max() < 1e-4


def test_too_few_points_guard(orc):
    """lastCorner <= 10 or lastSurf <= 100: no optimisation, the pose integration still runs (:224, :626-649)."""
    rng = np.random.default_rng(0)

    def cloud(n):
        p = np.zeros((n, 4), np.float32)
        p[:, :3] = rng.uniform(-5, 5, (n, 3))
        p[:, 3] = np.sort(rng.integers(0, 16, n)) + 0.05
        return p
    f = dict(full=cloud(300), sharp=cloud(8), less_sharp=cloud(9), flat=cloud(20), less_flat=cloud(90))
    ood, god = op.LaserOdometry(orc), loamx.LaserOdometry()
    for _ in range(3):
        ood.set_features(f)
        ood.process()
        god.process(f)
        assert np.array_equal(ood.transform, god.transform)
        assert np.abs(ood.transform_sum - god.transform_sum).max() < 1e-6
        assert god.stats()["iterations"] == 0


def test_imu_transform_plumbing(orc, small_world):
    """updateIMU values enter the start-up pose and the re-projection (identity check with a non-zero shift)."""
    sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=1, az_steps=600)
    f = op.ScanRegistration(orc).process(sw.points, sw.ring_sizes)
    t12 = np.array([0.01, 0.02, -0.01, 0.012, 0.018, -0.008, 0.05, -0.02, 0.03, 0.1, 0.0, -0.1], np.float32)
    ood, god = op.LaserOdometry(orc), loamx.LaserOdometry()
    ood.update_imu(t12)
    god.update_imu(t12)
    for _ in range(2):
        ood.set_features(f)
        ood.process()
        god.process(f)
    assert np.abs(ood.transform_sum - god.transform_sum).max() < POSE_TOL
    assert np.abs(ood.last_surf() - god.last_clouds()[1]).max() < 1e-4


def test_unordered_previous_clouds_take_the_general_scan(orc, small_world):
    """less_sharp / less_flat NOT ring-ordered (rings swapped block-wise): the ring-window walk of the reference depends on
    the point order; the GPU must fall back to the order-faithful scan kernel and still match the oracle."""
    poses = synth.trajectory(4)
    osr, ood, god = op.ScanRegistration(orc), op.LaserOdometry(orc), loamx.LaserOdometry()

    def scramble(c):
        ring = c[:, 3].astype(np.int32)
        blocks = [c[ring == r] for r in np.unique(ring)]
        order = np.random.default_rng(5).permutation(len(blocks))
        return np.ascontiguousarray(np.concatenate([blocks[i] for i in order]))
    for k in range(4):
        sw = synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=k)
        f = dict(osr.process(sw.points, sw.ring_sizes))
        f["less_sharp"], f["less_flat"] = scramble(f["less_sharp"]), scramble(f["less_flat"])
        ood.set_features(f)
        ood.process()
        god.process(f)
        assert np.abs(ood.transform - god.transform).max() < POSE_TOL
        assert np.abs(ood.transform_sum - god.transform_sum).max() < POSE_TOL
        st_o, st_g = ood.stats(), god.stats()
        assert st_o["iterations"] == st_g["iterations"] and st_o["sel"] == st_g["sel"]
