"""GPU: feature extraction through the C-ABI vs the oracle — integer/index work, so the bar is bit-exact."""
import os

import numpy as np
import pytest

import oracle_py as op
from conftest import GOLDEN
from loam_velodyne_amd import loamx, synth

pytestmark = pytest.mark.gpu
NAMES = ("sharp", "less_sharp", "flat", "less_flat")


def _same(fo, fg):
    for n in NAMES:
        assert fo[n].shape == fg[n].shape, n
        assert np.array_equal(fo[n], fg[n]), n


@pytest.mark.parametrize("sensor", ["VLP-16", "HDL-32", "HDL-64E"])
def test_full_size_sweeps_bit_exact(orc, small_world, sensor):
    poses = synth.trajectory(2)
    osr, gsr = op.ScanRegistration(orc), loamx.ScanRegistration()
    for k in range(2):
        sw = synth.make_sweep(small_world, sensor, poses[k], poses[k + 1], seed=k)
        _same(osr.process(sw.points, sw.ring_sizes), gsr.process(sw.points, sw.ring_sizes))


def test_golden_fixture(orc):
    g = np.load(os.path.join(GOLDEN, "features_vlp16.npz"))
    fg = loamx.ScanRegistration().process(g["points"], g["ring_sizes"])
    for n in NAMES:
        assert np.array_equal(fg[n], g[n]), n


def test_ragged_empty_and_short_rings(orc, small_world):
    sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=4, az_steps=700)
    pts = sw.points.reshape(16, 700, 4)
    rng = np.random.default_rng(0)
    sizes = [700, 0, 11, 10, 350, 699, 1, 64, 700, 12, 257, 0, 700, 33, 500, 128]   # <= 2*5+1 points => ring skipped
    rings = [pts[r, :n] for r, n in enumerate(sizes)]
    cloud = np.concatenate(rings, 0)
    _same(op.ScanRegistration(orc).process(cloud, sizes), loamx.ScanRegistration().process(cloud, sizes))


def test_nondefault_parameters(orc, small_world):
    sw = synth.make_sweep(small_world, "HDL-32", np.zeros(6), np.zeros(6), seed=5, az_steps=1000)
    cfg = dict(nFeatureRegions=4, curvatureRegion=3, maxCornerSharp=3, maxSurfaceFlat=2, lessFlatFilterSize=0.3,
               surfaceCurvatureThreshold=0.2)
    g = loamx.ScanRegistration(n_feature_regions=4, curvature_region=3, max_corner_sharp=3, max_surface_flat=2,
                               less_flat_filter_size=0.3, surface_curvature_threshold=0.2)
    _same(op.ScanRegistration(orc, **cfg).process(sw.points, sw.ring_sizes), g.process(sw.points, sw.ring_sizes))


@pytest.mark.parametrize("nreg,cr,sharp,flat,thr", [(8, 5, 2, 4, 0.1), (13, 5, 2, 4, 0.1), (6, 5, 6, 12, 0.1), (6, 8, 4, 8, 0.02), (7, 2, 3, 5, 0.5),
                                                   (12, 6, 5, 9, 0.05)])
def test_regions_side_by_side_equal_the_sequential_walk(orc, small_world, nreg, cr, sharp, flat, thr):
    """k_feat_ring makes the picks of a ring's regions side by side and settles the marks that cross a region boundary afterwards
    (markAsPicked writes into a flag array of the whole ring, BasicScanRegistration.cpp:367-386).  More regions than waves (several
    groups, the last one partial), many picks per region and wide suppression windows make such crossings frequent; short rings whose
    regions are smaller than the window take the sequential walk.  All of it bit-exact against the oracle's sequential loop."""
    cfg = dict(nFeatureRegions=nreg, curvatureRegion=cr, maxCornerSharp=sharp, maxSurfaceFlat=flat, surfaceCurvatureThreshold=thr)
    g = loamx.ScanRegistration(n_feature_regions=nreg, curvature_region=cr, max_corner_sharp=sharp, max_corner_less_sharp=0, max_surface_flat=flat,
                               surface_curvature_threshold=thr)   # (0 = 10 x max_corner_sharp, the reference constructor's rule)
    o = op.ScanRegistration(orc, **cfg)
    for seed, sensor, az in ((21, "HDL-32", 1200), (22, "VLP-16", 400)):
        sw = synth.make_sweep(small_world, sensor, np.zeros(6), np.zeros(6), seed=seed, az_steps=az)
        _same(o.process(sw.points, sw.ring_sizes), g.process(sw.points, sw.ring_sizes))
    # ragged rings: some long, some just long enough for the region formula, some with regions shorter than the window
    sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=23, az_steps=600)
    pts = sw.points.reshape(16, 600, 4)
    sizes = [600, 2 * cr + 2, 2 * cr + 1, nreg * cr + 2 * cr, nreg * cr + 2 * cr + 1, nreg * (cr - 1) + 2 * cr + 3, 97, 600, 45, 300, 0, 2 * cr + nreg, 599, 128, 64, 31]
    cloud = np.concatenate([pts[r, :n] for r, n in enumerate(sizes)], 0)
    _same(o.process(cloud, sizes), g.process(cloud, sizes))


def test_max_corner_less_sharp_and_reconfigure(orc, small_world):
    """RegistrationParams::maxCornerLessSharp is parsed on its own (ScanRegistration.cpp:100-109) and configure() on an existing
    object keeps its state (BasicScanRegistration.cpp:49-53)"""
    sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=8, az_steps=900)
    g = loamx.ScanRegistration(max_corner_sharp=3, max_corner_less_sharp=7)
    _same(op.ScanRegistration(orc, maxCornerSharp=3, maxCornerLessSharp=7).process(sw.points, sw.ring_sizes), g.process(sw.points, sw.ring_sizes))
    g.configure(max_corner_sharp=2, max_corner_less_sharp=0)           # 0 = 10 x max_corner_sharp, the constructor's rule (.cpp:22)
    _same(op.ScanRegistration(orc).process(sw.points, sw.ring_sizes), g.process(sw.points, sw.ring_sizes))
    with pytest.raises(loamx.LoamxError):                              # less-sharp limit below the sharp limit
        loamx.ScanRegistration(max_corner_sharp=4, max_corner_less_sharp=3)
    with pytest.raises(loamx.LoamxError):
        loamx.ScanRegistration(imu_history_size=0)


def test_pcl_layout_io(small_world):
    """32-byte pcl::PointXYZI records in and out give the same points as packed float4."""
    sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=6, az_steps=600)
    g = loamx.ScanRegistration()
    a = g.process(sw.points, sw.ring_sizes)
    b = g.process(loamx.to_pcl_layout(sw.points), sw.ring_sizes, pcl_layout=True)
    for n in NAMES:
        assert np.array_equal(a[n][:, :3], b[n][:, :3]) and np.array_equal(a[n][:, 3], b[n][:, 4])
        assert np.all(b[n][:, 3] == 1.0)


def test_error_behaviour(small_world):
    import ctypes as C
    g = loamx.ScanRegistration()
    pts = np.zeros((100, 4), np.float32)
    with pytest.raises(loamx.LoamxError):          # ring sizes do not add up
        g.process(pts, [50, 40])
    with pytest.raises(loamx.LoamxError):          # invalid parameter, same rule as the reference's parameter parsing
        loamx.ScanRegistration(max_corner_sharp=0)
    # undersized output buffer: LOAMX_E_CAPACITY with the needed count reported
    sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=7, az_steps=600)
    rs = np.ascontiguousarray(sw.ring_sizes, np.uint32)
    cin = loamx.cloud_of(sw.points)
    small = np.zeros((4, 4), np.float32)
    cs = loamx.cloud_of(small)
    rc = loamx.lib().loamx_scanreg_process(g.h, C.byref(cin), rs.ctypes.data_as(C.c_void_p), len(rs), C.byref(cs), None, None, None)
    assert rc == loamx.E_CAPACITY and cs.count > 4


def test_non_finite_input_is_rejected(small_world):
    world = small_world
    """BasicLaserOdometry.cpp:230, :252 / MultiScanRegistration.cpp:187-196: the reference never lets a NaN or Inf coordinate reach this
    stage; the binned-ring and feature-cloud entry points declare finite input and say so when it is not (LOAMX_E_INVALID) — the
    rings on the device, inside the curvature pass, the feature clouds on the host."""
    sw = synth.make_sweep(world, "VLP-16", synth.trajectory(1)[0], synth.trajectory(1)[1], seed=3, az_steps=600)
    g = loamx.ScanRegistration()
    ok = g.process(sw.points, sw.ring_sizes)
    for bad_value, where in ((np.nan, 1234), (np.inf, 0), (-np.inf, len(sw.points) - 1)):
        pts = sw.points.copy()
        pts[where, where % 3] = bad_value
        with pytest.raises(loamx.LoamxError) as e:
            g.process(pts, sw.ring_sizes)
        assert e.value.code == loamx.E_INVALID, e.value
    again = g.process(sw.points, sw.ring_sizes)   # the handle goes on working, results unchanged
    for name in ("sharp", "less_sharp", "flat", "less_flat"):
        assert np.array_equal(ok[name], again[name])
    od = loamx.LaserOdometry()
    od.process(ok)
    f = {k: v.copy() for k, v in ok.items() if k in ("sharp", "less_sharp", "flat", "less_flat")}
    f["flat"][5, 1] = np.nan
    with pytest.raises(loamx.LoamxError) as e:
        od.process(f)
    assert e.value.code == loamx.E_INVALID, e.value
