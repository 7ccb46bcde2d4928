"""GPU: the reference's edge-case branches on the device — the degeneracy projector in a featureless corridor
(BasicLaserOdometry.cpp:561-597, BasicLaserMapping.cpp:869-899), the too-few-rows guards (BasicLaserOdometry.cpp:485-488,
BasicLaserMapping.cpp:826-828) and the non-finite reset (BasicLaserOdometry.cpp:606-612)."""
import numpy as np
import pytest

import oracle_py as op
from conftest import POSE_TOL, collinear_previous_surf
from loam_velodyne_amd import loamx, synth

pytestmark = pytest.mark.gpu


def _corridor():
    w = synth.World(half_extent=400.0, pitch=1000.0)
    w.boxes = np.array([[-6.0, -5.0, -399.0, 399.0], [5.0, 6.0, -399.0, 399.0]])
    return w


def test_degenerate_corridor_odometry_and_registration(orc):
    """isDegenerate on the device: the projector from the row-zeroed eigenvector matrix (odometry thr 10, mapping thr 100),
    step by step from the oracle's state"""
    w = _corridor()
    poses = synth.trajectory(5, step=1.0, yaw_step_deg=0.0)
    osr, ood, omp, god = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc), loamx.LaserOdometry()
    degenerate = 0
    for k in range(5):
        sw = synth.make_sweep(w, "VLP-16", poses[k], poses[k + 1], seed=k, az_steps=900)
        f = osr.process(sw.points, sw.ring_sizes)
        ood.set_features(f)
        ood.process()
        if k > 0:
            god.set_transform(prev_transform)          # same seed as the oracle had before this step
            god.set_transform_sum(prev_sum)
        god.process(f)
        assert np.abs(ood.transform - god.transform).max() < POSE_TOL, k
        assert np.abs(ood.transform_sum - god.transform_sum).max() < POSE_TOL, k
        prev_transform, prev_sum = ood.transform, ood.transform_sum
        lc, ls, full, ts = ood.last_corner(), ood.last_surf(), ood.full_to_end(), ood.transform_sum
        g = loamx.LaserMapping()
        g.load_cubes(omp.cloud("corner_cubes"), omp.cloud("surf_cubes"))
        g.set_transform("aft", omp.transform("aft"))
        g.set_transform("bef", omp.transform("bef"))
        g.update_odometry(ts)
        omp.set_inputs(lc, ls, full, ts)
        omp.process()
        g.process(lc, ls, full)
        assert np.abs(omp.transform("aft") - g.transform("aft")).max() < POSE_TOL, k
        assert omp.stats()["degenerate"] == g.stats()["degenerate"], k
        degenerate += omp.stats()["degenerate"]
    assert degenerate >= 3


def test_odometry_too_few_rows(orc, small_world):
    """fewer than 10 selected rows: every iteration is skipped, the seeded motion estimate stays and is accumulated (:485-488)"""
    poses = synth.trajectory(2)
    sws = [synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=100 + k, az_steps=900) for k in range(2)]
    sr = op.ScanRegistration(orc)
    f0, f1 = sr.process(sws[0].points, sws[0].ring_sizes), sr.process(sws[1].points, sws[1].ring_sizes)
    far = {n: v + np.array([60.0, 0, 60.0, 0], np.float32) for n, v in f1.items()}
    seed = np.float32([0.001, 0.002, -0.001, 0.05, 0.0, -0.3])
    ood, god = op.LaserOdometry(orc), loamx.LaserOdometry()
    ood.set_features(f0)
    ood.process()
    god.process(f0)
    ood.set_features(far)
    ood.set_transform(seed)
    ood.process()
    god.set_transform(seed)
    god.process(far)
    assert np.array_equal(god.transform, seed) and np.array_equal(ood.transform, seed)
    assert np.abs(ood.transform_sum - god.transform_sum).max() < 1e-6


def test_mapping_too_few_rows(orc, small_world):
    """fewer than 50 selected rows: the iterations are skipped (:826-828), the pose prediction stands"""
    poses = synth.trajectory(1)
    sw = synth.make_sweep(small_world, "VLP-16", poses[0], poses[1], seed=100, az_steps=900)
    sr, od = op.ScanRegistration(orc), op.LaserOdometry(orc)
    od.set_features(sr.process(sw.points, sw.ring_sizes))
    od.process()
    lc, ls, full = od.last_corner(), od.last_surf(), od.full_to_end()
    o, g = op.LaserMapping(orc), loamx.LaserMapping()
    o.set_inputs(lc, ls, full, np.zeros(6, np.float32))
    o.process()
    g.update_odometry(np.zeros(6, np.float32))
    g.process(lc, ls, full)
    up = np.float32([0, 0, 0, 0, 40.0, 0])
    o.set_inputs(lc, ls, full, up)
    o.process()
    g.update_odometry(up)
    g.process(lc, ls, full)
    assert o.stats()["optimized"] == 1 and o.stats()["sel"] < 50 and g.stats()["sel"] == o.stats()["sel"]
    assert np.abs(o.transform("aft") - g.transform("aft")).max() < 1e-6


def test_nan_rows_through_the_odometry_solve(orc, small_world):
    """Duplicated tripod points (the same corner on two adjacent rings of the previous cloud) make l12 = 0: the rows of those
    features are NaN and — NaN != 0 — selected (BasicLaserOdometry.cpp:330-361).  The whole normal system is then non-finite,
    the pivoted QR (Eigen's semantics: maxima seeded with the first coefficient) answers NaN, and the non-finite reset of
    :606-612 zeroes the transform — in each of the first five iterations (the distance weights of the sixth deselect the NaN rows,
    and the run goes on as the run that starts from zero).  The device takes NaN through its reduction, its QR and
    its reset to the same end (tests/test_ref_pinning.py::test_non_finite_system_takes_the_reset_branch pins the oracle to the
    reference's own code on the same input)."""
    poses = synth.trajectory(2)
    sws = [synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=100 + k, az_steps=900) for k in range(2)]
    sr = op.ScanRegistration(orc)
    f0, f1 = sr.process(sws[0].points, sws[0].ring_sizes), sr.process(sws[1].points, sws[1].ring_sizes)
    ls = f0["less_sharp"]
    dup = ls.copy()
    dup[:, 3] += 1.0
    both = np.concatenate([ls, dup])
    f0d = dict(f0)
    f0d["less_sharp"] = both[np.argsort(np.floor(both[:, 3]), kind="stable")]
    seed = np.float32([0.001, 0.002, -0.001, 0.05, 0.0, -0.3])
    ood, zod, god, gzd = op.LaserOdometry(orc), op.LaserOdometry(orc), loamx.LaserOdometry(), loamx.LaserOdometry()
    for od in (ood, zod):
        od.set_features(f0d)
        od.process()
    god.process(f0d)
    gzd.process(f0d)
    for od, start in ((ood, seed), (zod, np.zeros(6, np.float32))):
        od.set_features(f1)
        od.set_transform(start)
        od.process()
    god.set_transform(seed)
    god.process(f1)
    gzd.set_transform(np.zeros(6, np.float32))
    gzd.process(f1)
    assert np.all(np.isfinite(god.transform)) and np.all(np.isfinite(god.transform_sum))
    # the seed is gone on both sides: the run equals the run that starts from zero, bit for bit — the reset branch was taken
    assert np.array_equal(ood.transform, zod.transform) and not np.array_equal(ood.transform, seed)
    assert np.array_equal(god.transform, gzd.transform) and np.array_equal(god.transform_sum, gzd.transform_sum)
    assert god.stats()["iterations"] == gzd.stats()["iterations"] > 5 and ood.stats()["iterations"] > 5
    # (device and oracle need not agree beyond that here: original and duplicate are exactly equidistant, which of the two a 1-NN
    # search returns is a property of the search structure — the kd-tree's traversal order there, the lowest index here — and once
    # the NaN rows are deselected the two runs optimise over different tripods.  Both recover the motion:)
    assert np.abs(ood.transform - god.transform).max() < 0.05


def test_reset_branch_device_equals_oracle(orc, small_world):
    """The non-finite reset (BasicLaserOdometry.cpp:606-612) taken WITHOUT a distance tie, so that device and oracle can be compared
    with each other: the previous surface cloud is a straight line (conftest.collinear_previous_surf — the
    reference's own build takes the branch on the same input, ::test_reset_branch_through_collinear_tripods), every plane tripod is
    collinear, its rows are NaN and selected in iterations 0-4, the transform is reset five times, then the edge rows carry the run."""
    poses = synth.trajectory(2)
    sws = [synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=100 + k, az_steps=900) for k in range(2)]
    sr = op.ScanRegistration(orc)
    f0, f1 = sr.process(sws[0].points, sws[0].ring_sizes), sr.process(sws[1].points, sws[1].ring_sizes)
    f0d = dict(f0)
    f0d["less_flat"] = collinear_previous_surf(f1)
    seed = np.float32([0.001, 0.002, -0.001, 0.05, 0.0, -0.3])
    ood, god, gzd = op.LaserOdometry(orc), loamx.LaserOdometry(), loamx.LaserOdometry()
    ood.set_features(f0d)
    ood.process()
    ood.set_features(f1)
    ood.set_transform(seed)
    ood.process()
    for g, start in ((god, seed), (gzd, np.zeros(6, np.float32))):
        g.process(f0d)
        g.set_transform(start)
        g.process(f1)
    assert np.array_equal(god.transform, gzd.transform) and not np.array_equal(god.transform, seed)   # reset taken on the device
    assert np.all(np.isfinite(god.transform))
    assert god.stats()["iterations"] == ood.stats()["iterations"] and god.stats()["sel"] == ood.stats()["sel"]
    assert np.abs(ood.transform - god.transform).max() < 1e-5 and np.abs(ood.transform_sum - god.transform_sum).max() < 1e-5
