"""GPU: batched scan-to-map registration against a frozen sub-map (loamx_batch_*) vs the oracle."""
import os

import numpy as np
import pytest

import oracle_py as op
from conftest import GOLDEN, POSE_TOL
from loam_velodyne_amd import loamx, synth

pytestmark = pytest.mark.gpu


def _inputs(orc, world, sensor, B, seed=1, az=None):
    rng = np.random.default_rng(seed)
    sr = op.ScanRegistration(orc)
    cl, sl, guesses, fulls = [], [], [], []
    for k in range(B):
        gt = np.array([0.01 * rng.normal(), 0.3 * rng.normal(), 0.01 * rng.normal(), 3 * rng.normal(), 0.05 * rng.normal(), 3 * rng.normal()])
        sw = synth.make_sweep(world, sensor, gt, gt, seed=seed * 100 + k, az_steps=az)
        f = sr.process(sw.points, sw.ring_sizes)
        c, s = f["less_sharp"].copy(), f["less_flat"].copy()
        c[:, 3] = np.floor(c[:, 3])
        s[:, 3] = np.floor(s[:, 3])
        cl.append(c)
        sl.append(s)
        fulls.append(sw.points[::4].copy())
        guesses.append(gt + np.array([0.003, 0.003, 0.003, 0.05, 0.05, 0.05]) * rng.normal(size=6))
    return cl, sl, np.array(guesses, np.float32), fulls


def _oracle(orc, cm, sm, cl, sl, guesses):
    mp = op.LaserMapping(orc)
    mp.set_frozen(cm, sm)
    poses, stats = [], []
    for k in range(len(cl)):
        poses.append(mp.register_frozen(cl[k], sl[k], guesses[k]))
        stats.append(mp.stats())
    return np.array(poses), stats


def test_parity_vlp16_100k(orc):
    world = synth.World(half_extent=65.0)
    cm, sm = world.make_map(100000)
    cl, sl, guesses, fulls = _inputs(orc, world, "VLP-16", 4)
    oposes, ostats = _oracle(orc, cm, sm, cl, sl, guesses)
    b = loamx.Batch(4)
    b.set_frozen(cm, sm)
    b.upload(cl, sl, guesses, full_res=fulls)
    assert b.run() == loamx.OK
    gposes, gstats = b.download()
    assert np.abs(gposes - oposes).max() < POSE_TOL
    for k in range(4):
        assert gstats[k, 0] == ostats[k]["iterations"] and gstats[k, 2] == ostats[k]["corner_ds"] and gstats[k, 3] == ostats[k]["surf_ds"]
        assert int(gstats[k, 1]) == ostats[k]["sel"]                    # rows selected in the last iteration: equal, no flipped threshold
        # transformFullResToMap with the final pose
        R = synth.rot_zxy(*gposes[k, :3])
        want = fulls[k][:, :3].astype(np.float64) @ R.T + gposes[k, 3:]
        got = b.download_full_res(k)
        assert np.abs(got[:, :3] - want).max() < 2e-4 and np.array_equal(got[:, 3], fulls[k][:, 3])


@pytest.mark.parametrize("env", [{"LOAMX_VOX_LEGACY": "1"}, {"LOAMX_VOX_LEGACY": "1", "LOAMX_VDS_WGS": "3"},
                                 {"LOAMX_VOX_LEGACY": "1", "LOAMX_VDS_WGS": "3", "LOAMX_VDS_GLOBAL": "1"}, {"LOAMX_VOX_LEGACY": "1", "LOAMX_VDS_GLOBAL": "1"}])
def test_voxel_grid_does_not_depend_on_grid_size(orc, env, monkeypatch):
    """The stack clouds' voxel grid has three implementations: the bucketed path (default), the persistent segmented kernel and the
    persistent general kernel (LOAMX_VOX_LEGACY / LOAMX_VDS_GLOBAL); the persistent ones claim their tiles from a counter, so they
    may run with 3 workgroups instead of one per tile (what a crowded device would leave resident).  Poses and per-sweep
    statistics (down-sampled sizes, selected rows) are bit-identical across all of them."""
    world = synth.World(half_extent=65.0)
    cm, sm = world.make_map(100000)
    cl, sl, guesses, _ = _inputs(orc, world, "VLP-16", 4, seed=5)

    def run():
        b = loamx.Batch(4)
        b.set_frozen(cm, sm)
        b.upload(cl, sl, guesses)
        assert b.run() == loamx.OK
        poses, stats = b.download()
        return poses, stats

    p0, s0 = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p1, s1 = run()
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)   # bit for bit: the voxel means feed every residual


def test_full_size_hdl64_1m_map(orc):
    """BASELINE full size: HDL-64E sweeps against a 1M-point sub-map — oracle parity on 2 sweeps plus size-independent
    properties on a batch of 8 (batch-composition invariance, permutation equivariance, idempotence)."""
    world = synth.World(half_extent=125.0)
    cm, sm = world.make_map(1_000_000)
    cl, sl, guesses, _ = _inputs(orc, world, "HDL-64E", 8, seed=3)
    b = loamx.Batch(8)
    b.set_frozen(cm, sm)
    b.upload(cl, sl, guesses)
    b.run()
    poses, stats = b.download()
    oposes, _ = _oracle(orc, cm, sm, cl[:2], sl[:2], guesses[:2])
    assert np.abs(poses[:2] - oposes).max() < POSE_TOL
    assert np.all(stats[:, 0] <= 10) and np.all(stats[:, 1] >= 50)
    # a sweep registered alone gives bit-identical results to the same sweep inside the batch
    b1 = loamx.Batch(1)
    b1.set_frozen(cm, sm)
    b1.upload(cl[5:6], sl[5:6], guesses[5:6])
    b1.run()
    p1, s1 = b1.download()
    assert np.array_equal(p1[0], poses[5]) and np.array_equal(s1[0], stats[5])
    # permuting the batch permutes the results
    perm = [3, 0, 7, 1, 6, 2, 5, 4]
    b.upload([cl[i] for i in perm], [sl[i] for i in perm], guesses[perm])
    b.run()
    pp, sp = b.download()
    assert np.array_equal(pp, poses[perm]) and np.array_equal(sp, stats[perm])
    # idempotence: restarting from the converged poses stays there and stops at once
    b.upload(cl, sl, poses)
    b.run()
    p2, s2 = b.download()
    assert np.abs(p2 - poses).max() < 5e-4 and np.all(s2[:, 0] <= 2)


@pytest.mark.parametrize("sensor,map_points,half", [("HDL-32", 500_000, 125.0), ("HDL-64E", 2_000_000, 125.0)])
def test_full_size_other_configs(orc, sensor, map_points, half):
    """BASELINE configs[2] (HDL-32, 500 k-pt map) and configs[4] (HDL-64E, 2 M-pt map): frozen-map registration vs the oracle"""
    world = synth.World(half_extent=half)
    cm, sm = world.make_map(map_points)
    cl, sl, guesses, _ = _inputs(orc, world, sensor, 3, seed=7)
    oposes, ostats = _oracle(orc, cm, sm, cl, sl, guesses)
    b = loamx.Batch(3)
    b.set_frozen(cm, sm)
    b.upload(cl, sl, guesses)
    assert b.run() == loamx.OK
    poses, stats = b.download()
    assert np.abs(poses - oposes).max() < POSE_TOL, float(np.abs(poses - oposes).max())
    for k in range(3):
        assert stats[k, 0] == ostats[k]["iterations"] and stats[k, 2] == ostats[k]["corner_ds"] and stats[k, 3] == ostats[k]["surf_ds"]
        assert int(stats[k, 1]) == ostats[k]["sel"]


def test_golden_pipeline_inputs(orc):
    """Registration stage on the committed fixture: features and odometry from the oracle, registration on the GPU."""
    g = np.load(os.path.join(GOLDEN, "pipeline_vlp16.npz"))
    b = loamx.Batch(1)
    b.set_frozen(g["corner_map"], g["surf_map"])
    for s in range(2):
        osr, ood, omp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
        omp.set_transform("aft", g[f"start_{s}"])
        for t in range(4):
            ood.set_features(osr.process(g[f"points_{s}_{t}"], g[f"rings_{s}_{t}"]))
            ood.process()
            if t > 0:
                omp.set_transform("sum", ood.transform_sum)
                guess = omp.associate()
                b.upload([ood.last_corner()], [ood.last_surf()], guess[None])
                b.run()
                pose, _ = b.download()
                assert np.abs(pose[0] - g[f"aft_{s}"][t]).max() < POSE_TOL
                omp.set_transform("bef", ood.transform_sum)
                omp.set_transform("aft", g[f"aft_{s}"][t])


def test_sparse_map_guard_and_empty_inputs(orc):
    pts = np.zeros((50, 4), np.float32)
    pts[:, 0] = np.arange(50)
    b = loamx.Batch(2)
    b.set_frozen(pts[:5], pts)                       # <= 10 corner points: optimisation skipped (BasicLaserMapping.cpp:628)
    guess = np.array([[0.0, 0.1, 0.0, 1.0, 2.0, 3.0], [0.2, 0.0, 0.1, -1.0, 0.0, 0.5]], np.float32)
    b.upload([pts, pts[:0]], [pts, pts[:0]], guess)  # second sweep: empty clouds
    assert b.run() == loamx.SKIPPED
    poses, stats = b.download()
    assert np.array_equal(poses, guess) and np.all(stats[:, 0] == 0)
    # enough map, but a sweep with too few features: iterations burn without moving the pose (:826-828)
    world = synth.World(half_extent=45.0)
    cm, sm = world.make_map(30000)
    b.set_frozen(cm, sm)
    few = cm[:20].copy()
    b.upload([few, pts[:0]], [sm[:20].copy(), pts[:0]], guess)
    assert b.run() == loamx.OK
    poses, stats = b.download()
    assert np.array_equal(poses, guess) and stats[0, 0] == 10 and stats[0, 1] < 50
    mp = op.LaserMapping(orc)
    mp.set_frozen(cm, sm)
    assert np.array_equal(mp.register_frozen(few, sm[:20].copy(), guess[0]), guess[0]) and mp.stats()["iterations"] == 10


def test_invalid_arguments():
    b = loamx.Batch(2)
    with pytest.raises(loamx.LoamxError):
        loamx.Batch(0)
    with pytest.raises(loamx.LoamxError):            # more sweeps than the handle was created for
        z = np.zeros((4, 4), np.float32)
        b.upload([z] * 3, [z] * 3, np.zeros((3, 6), np.float32))
    with pytest.raises(loamx.LoamxError):            # run before upload
        loamx.Batch(1).run()


def test_double_buffered_map_epochs(orc):
    """BASELINE configs[4]: the next epoch's sub-map is indexed in the background while sweeps are registered against the
    current one; after the swap the results are exactly those of a handle that was given the new map outright — and the
    registrations that ran in between still saw the OLD map."""
    world_a, world_b = synth.World(half_extent=65.0), synth.World(half_extent=65.0, seed=5)
    cm_a, sm_a = world_a.make_map(100000)
    cm_b, sm_b = world_b.make_map(120000)
    cl, sl, guesses, _ = _inputs(orc, world_a, "VLP-16", 4)

    def fresh(cm, sm):
        b = loamx.Batch(4)
        b.set_frozen(cm, sm)
        b.upload(cl, sl, guesses)
        assert b.run() == loamx.OK
        return b.download()
    ref_a, ref_b = fresh(cm_a, sm_a), fresh(cm_b, sm_b)
    assert not np.array_equal(ref_a[0], ref_b[0])
    b = loamx.Batch(4)
    b.set_frozen(cm_a, sm_a)
    assert b.swap_frozen() is False                                  # nothing staged yet
    b.stage_frozen(cm_b, sm_b)
    b.upload(cl, sl, guesses)
    assert b.run() == loamx.OK                                       # still epoch A
    pa, sa = b.download()
    assert np.array_equal(pa, ref_a[0]) and np.array_equal(sa, ref_a[1])
    assert b.swap_frozen() is True
    b.upload(cl, sl, guesses)
    assert b.run() == loamx.OK                                       # epoch B
    pb, sb = b.download()
    assert np.array_equal(pb, ref_b[0]) and np.array_equal(sb, ref_b[1])
    # and back again: the buffers of epoch A are recycled for the next staging
    b.stage_frozen(cm_a, sm_a)
    assert b.swap_frozen() is True
    b.upload(cl, sl, guesses)
    assert b.run() == loamx.OK
    pa2, sa2 = b.download()
    assert np.array_equal(pa2, ref_a[0]) and np.array_equal(sa2, ref_a[1])
