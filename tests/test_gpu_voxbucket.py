"""GPU: the bucketed voxel grid of the registration's stack clouds (voxbucket.hip) — pcl::VoxelGrid on
laserCloudCornerStack / laserCloudSurfStack, BasicLaserMapping.cpp:512-527 — against the oracle's clouds point for point,
against the general kernel bit for bit, and through its give-up cases (the run is repeated through the general kernel)."""
import numpy as np
import pytest

import oracle_py as op
from loam_velodyne_amd import loamx, synth
from test_gpu_batch import _inputs

pytestmark = pytest.mark.gpu


def _run(cm, sm, cl, sl, guesses, **cfg):
    b = loamx.Batch(len(cl), **cfg)
    b.set_frozen(cm, sm)
    b.upload(cl, sl, guesses)
    assert b.run() == loamx.OK
    poses, stats = b.download()
    ds = [b.download_ds(k) for k in range(len(cl))]
    return poses, stats, ds


def _oracle_ds(orc, cm, sm, c, s, g, **cfg):
    mp = op.LaserMapping(orc, **cfg)
    mp.set_frozen(cm, sm)
    mp.register_frozen(c, s, g)
    return mp.cloud("corner_stack_ds"), mp.cloud("surf_stack_ds")


@pytest.mark.parametrize("sensor,B,map_points,half", [("VLP-16", 4, 100_000, 65.0), ("HDL-32", 3, 200_000, 125.0), ("HDL-64E", 8, 300_000, 125.0)])
def test_ds_clouds_equal_the_oracles_bit_for_bit(orc, sensor, B, map_points, half):
    world = synth.World(half_extent=half)
    cm, sm = world.make_map(map_points)
    cl, sl, guesses, _ = _inputs(orc, world, sensor, B, seed=11)
    poses, stats, ds = _run(cm, sm, cl, sl, guesses)
    for k in range(min(B, 3)):
        oc, osf = _oracle_ds(orc, cm, sm, cl[k], sl[k], guesses[k])
        assert ds[k][0].shape == oc.shape and ds[k][1].shape == osf.shape
        assert np.array_equal(ds[k][0], oc) and np.array_equal(ds[k][1], osf)   # same voxels, same order, same float sums


def test_bucketed_path_equals_the_general_kernel(orc, monkeypatch):
    world = synth.World(half_extent=125.0)
    cm, sm = world.make_map(300_000)
    cl, sl, guesses, _ = _inputs(orc, world, "HDL-64E", 8, seed=13)
    p0, s0, d0 = _run(cm, sm, cl, sl, guesses)
    monkeypatch.setenv("LOAMX_VOX_LEGACY", "1")
    p1, s1, d1 = _run(cm, sm, cl, sl, guesses)
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
    for a, b in zip(d0, d1):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_gives_up_cleanly_and_repeats_through_the_general_kernel(orc, monkeypatch):
    """Three ways out of the bucketed path, each compared with a handle that never takes it:
    (a) a leaf so small that the box has more than INT_MAX voxels — PCL's pass-through case;
    (b) thousands of points inside a few voxels — one histogram bin larger than a bucket;
    (c) sweeps of the batch that are empty or tiny next to ordinary ones (no give-up: empty segments own an empty bucket)."""
    world = synth.World(half_extent=65.0)
    cm, sm = world.make_map(100_000)
    cl, sl, guesses, _ = _inputs(orc, world, "VLP-16", 3, seed=17)
    rng = np.random.default_rng(3)
    blob = np.zeros((6000, 4), np.float32)
    blob[:, :3] = np.array([5.0, 0.3, 7.0], np.float32) + 0.15 * rng.random((6000, 3), dtype=np.float32)
    cases = {
        "tiny_leaf": (cl, sl, guesses, dict(corner_filter_size=0.0011, surf_filter_size=0.0011)),
        "dense_blob": ([cl[0], cl[1]], [np.concatenate([sl[0], blob]), sl[1]], guesses[:2], {}),
        "ragged": ([cl[0], cl[1][:0], cl[2][:7]], [sl[0], sl[1][:0], sl[2][:5]], guesses, {}),
    }
    for name, (c, s, g, cfg) in cases.items():
        monkeypatch.delenv("LOAMX_VOX_LEGACY", raising=False)
        p0, s0, d0 = _run(cm, sm, c, s, g, **cfg)
        monkeypatch.setenv("LOAMX_VOX_LEGACY", "1")
        p1, s1, d1 = _run(cm, sm, c, s, g, **cfg)
        assert np.array_equal(p0, p1) and np.array_equal(s0, s1), name
        for a, b in zip(d0, d1):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), name
    # the blob really is down-sampled as the oracle does it (a handful of voxels holding thousands of points each)
    monkeypatch.delenv("LOAMX_VOX_LEGACY", raising=False)
    c, s, g, _ = cases["dense_blob"]
    _, _, d = _run(cm, sm, c, s, g)
    oc, osf = _oracle_ds(orc, cm, sm, c[0], s[0], g[0])
    assert np.array_equal(d[0][0], oc) and np.array_equal(d[0][1], osf)
