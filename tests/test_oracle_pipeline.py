"""CPU: end-to-end behaviour of the oracle on synthetic sweeps with known ground truth.  This validates the signs of
the Jacobians and the pose conventions independently of the reference (SURVEY.md §8c self-check 4)."""
import numpy as np

import oracle_py as op
from loam_velodyne_amd import synth


def test_slam_tracks_ground_truth(orc, small_world):
    n = 7
    poses = synth.trajectory(n)
    sr, od, mp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
    for k in range(n):
        sw = synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=k, az_steps=900)
        f = sr.process(sw.points, sw.ring_sizes)
        od.set_features(f)
        od.process()
        mp.set_inputs(od.last_corner(), od.last_surf(), od.full_to_end(), od.transform_sum)
        assert mp.process()
    aft = mp.transform("aft")
    # map frame = sensor frame at the (static) first sweep = world frame; pose at the end of the last sweep
    assert np.abs(aft[3:] - poses[n, 3:]).max() < 0.12
    assert np.abs(aft[:3] - poses[n, :3]).max() < 0.01
    st = mp.stats()
    assert st["optimized"] == 1 and st["sel"] > 1000 and st["degenerate"] == 0


def test_frozen_registration_recovers_pose(orc, small_world):
    corner_map, surf_map = small_world.make_map(60000)
    rng = np.random.default_rng(0)
    gt = np.array([0.01, 0.2, -0.005, 1.5, 0.03, -2.0])
    sw = synth.make_sweep(small_world, "VLP-16", gt, gt, seed=3)
    f = op.ScanRegistration(orc).process(sw.points, sw.ring_sizes)
    mp = op.LaserMapping(orc)
    mp.set_frozen(corner_map, surf_map)
    guess = gt + np.array([0.004, -0.004, 0.003, 0.06, -0.05, 0.07])
    pose = mp.register_frozen(f["less_sharp"], f["less_flat"], guess)
    assert np.abs(pose[3:] - gt[3:]).max() < 0.02
    assert np.abs(pose[:3] - gt[:3]).max() < 2e-3
    assert mp.stats()["iterations"] <= 10


def test_sparse_map_guard(orc):
    # <= 10 corner or <= 100 surf map points: optimisation is skipped and Bef/Aft stay stale (BasicLaserMapping.cpp:628)
    mp = op.LaserMapping(orc)
    pts = np.zeros((50, 4), np.float32)
    pts[:, 0] = np.arange(50)
    mp.set_frozen(pts[:5], pts)
    guess = np.array([0.0, 0.1, 0.0, 1.0, 2.0, 3.0], np.float32)
    pose = mp.register_frozen(pts, pts, guess)
    assert np.array_equal(pose, guess)
    assert mp.stats()["optimized"] == 0
    assert np.array_equal(mp.transform("aft"), np.zeros(6, np.float32))
