"""CPU: the oracle (and the product's host-side pose fusion) pinned against the REFERENCE'S OWN CODE where that compiles
without ROS / PCL / Eigen (oracle/Makefile target `ref`, outputs under oracle/_ref/, sources compiled where they lie):
  libref_loam_small.so  src/lib/BasicTransformMaintenance.cpp, src/lib/math_utils.h, include/loam_velodyne/Angle.h, CircularBuffer.h
  libref_scanreg.so     src/lib/BasicScanRegistration.cpp — feature extraction and the IMU state machine
The only stand-ins are the minimal <pcl/...> headers in oracle/ref_stubs (a point struct, a std::vector cloud, and a VoxelGrid
interface that forwards to the oracle's voxel grid: the grid itself therefore stays unpinned).  Every comparison is bit for bit.
The vendored nanoflann is pinned in test_oracle_primitives.py."""
import ctypes as C
import os

import numpy as np
import pytest

import conftest
import oracle_py as op
from loam_velodyne_amd import loamx, synth

REF = op.ref_small()
pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/libref_loam_small.so not built (no /root/reference here)")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_transform_maintenance_oracle_and_product_equal_the_reference_bitwise(orc):
    rng = np.random.default_rng(11)
    tm = loamx.TransformMaintenance()
    for _ in range(500):
        s, b, a = (np.concatenate([rng.uniform(-1.2, 1.2, 3), rng.uniform(-80, 80, 3)]).astype(np.float32) for _ in range(3))
        want = np.zeros(6, np.float32)
        REF.ref_tm_associate(_p(s), _p(b), _p(a), _p(want))
        assert np.array_equal(op.tm_associate(orc, s, b, a), want)
        tm.update_odometry(s)
        tm.update_mapping_transform(a, b)
        assert np.array_equal(tm.associate_to_map(), want)


def test_angle_semantics(orc):
    """Angle.h: cached float sin / cos, unary minus flips the sine only, += re-derives both."""
    rng = np.random.default_rng(12)
    for _ in range(300):
        rad, add, neg = np.float32(rng.uniform(-7, 7)), np.float32(rng.choice([0.0, rng.uniform(-1, 1)])), int(rng.integers(0, 2))
        r, o = np.zeros(3, np.float32), np.zeros(3, np.float32)
        REF.ref_angle(C.c_float(rad), neg, C.c_float(add), _p(r))
        orc.L.orc_angle(C.c_float(rad), neg, C.c_float(add), _p(o))
        assert np.array_equal(r, o)


def test_rotations(orc):
    """math_utils.h rotX / rotY / rotZ / rotateZXY / rotateYXZ: same operand order, same float results."""
    rng = np.random.default_rng(13)
    for _ in range(500):
        which = int(rng.integers(0, 5))
        p = rng.uniform(-60, 60, 3).astype(np.float32)
        a = rng.uniform(-3.2, 3.2, 3).astype(np.float32)
        r, o = p.copy(), p.copy()
        REF.ref_rotate(which, _p(r), C.c_float(a[0]), C.c_float(a[1]), C.c_float(a[2]))
        orc.L.orc_rotate(which, _p(o), C.c_float(a[0]), C.c_float(a[1]), C.c_float(a[2]))
        assert np.array_equal(r, o), which
    for v in (0.1, -2.5, 3.1415927):     # rad2deg / deg2rad go through double (math_utils.h:30-47)
        f = float(np.float32(v))   # (double arithmetic on the float value, then rounded once)
        assert REF.ref_rad2deg(C.c_float(v)) == np.float32(f * 180.0 / np.pi)
        assert REF.ref_deg2rad(C.c_float(v)) == np.float32(f * np.pi / 180.0)


def test_circular_buffer_is_a_bounded_fifo():
    """CircularBuffer.h: push() overwrites the oldest element once full and operator[] counts from the oldest — i.e. the
    bounded deque (pop_front on overflow) that the oracle and the product keep their IMU histories in."""
    from collections import deque
    for cap, n in ((5, 3), (5, 5), (5, 12), (200, 333), (1, 4)):
        vals = np.arange(100, 100 + n, dtype=np.int32)
        out = np.zeros(cap, np.int32)
        size = REF.ref_circular(cap, _p(vals), n, _p(out), cap)
        d = deque(maxlen=cap)
        for v in vals:
            d.append(int(v))
        assert size == len(d) and out[:size].tolist() == list(d)


# ---- the reference's own BasicScanRegistration.cpp (feature extraction + IMU bookkeeping) ---------------------------------
needs_sr = pytest.mark.skipif(not op.RefScanRegistration.available(), reason="oracle/_ref/libref_scanreg.so not built")


@needs_sr
@pytest.mark.parametrize("sensor,az,cfg", [("VLP-16", 900, {}), ("HDL-32", 700, {}), ("HDL-64E", 512, {}),
                                           ("VLP-16", 600, dict(nFeatureRegions=4, curvatureRegion=3, maxCornerSharp=3, maxSurfaceFlat=2,
                                                                surfaceCurvatureThreshold=0.2)),
                                           ("VLP-16", 600, dict(maxCornerSharp=3, maxCornerLessSharp=7)),      # parsed on its own, ScanRegistration.cpp:100-109
                                           ("HDL-32", 500, dict(maxCornerSharp=1, maxCornerLessSharp=40))])
def test_feature_extraction_equals_the_reference(orc, small_world, sensor, az, cfg):
    """processScanlines / extractFeatures (BasicScanRegistration.cpp:28-46, :155-386) run by the reference's own code: sharp,
    less-sharp and flat picks identical point for point; the less-flat cloud identical too (its candidate set is the
    reference's, its voxel grid is the oracle's on both sides — see oracle/ref_stubs/pcl/filters/voxel_grid.h)."""
    from loam_velodyne_amd import synth
    for seed in (1, 2):
        sw = synth.make_sweep(small_world, sensor, np.zeros(6), np.array([0.002, 0.02, -0.001, 0.2, 0.01, 0.8]), seed=seed, az_steps=az)
        o = op.ScanRegistration(orc, **cfg).process(sw.points, sw.ring_sizes)
        r = op.RefScanRegistration(**cfg).process(sw.points, sw.ring_sizes)
        for name in ("full", "sharp", "less_sharp", "flat", "less_flat"):
            assert o[name].shape == r[name].shape, (name, o[name].shape, r[name].shape)
            assert np.array_equal(o[name], r[name]), name


@needs_sr
def test_ragged_and_short_rings_equal_the_reference(orc):
    """rings shorter than 2 x curvatureRegion + 1 are skipped (:163-165); empty rings; duplicate / degenerate geometry"""
    rng = np.random.default_rng(5)
    sizes = [0, 7, 11, 12, 40, 3, 300]
    rings = []
    for r, n in enumerate(sizes):
        ang = np.sort(rng.uniform(-np.pi, np.pi, n))
        rad = 5 + 2 * np.sin(3 * ang) + (rng.random(n) < 0.1) * 3          # jumps -> occlusion masks
        p = np.stack([rad * np.sin(ang), np.full(n, 0.1 * r), rad * np.cos(ang), r + 0.1 * (ang + np.pi) / (2 * np.pi)], 1)
        rings.append(p.astype(np.float32))
    pts = np.concatenate(rings)
    o = op.ScanRegistration(orc).process(pts, sizes)
    r = op.RefScanRegistration().process(pts, sizes)
    for name in ("full", "sharp", "less_sharp", "flat", "less_flat"):
        assert np.array_equal(o[name], r[name]), name


@needs_sr
def test_imu_state_machine_equals_the_reference(orc, small_world):
    """updateIMUData, projectPointToStartOfSweep (incl. the monotone history index and the stale-scan-time order of events),
    reset and updateIMUTransform (:55-152, :258-281) by the reference's own code, over sweeps with a wrapping history."""
    from loam_velodyne_amd import synth
    o, r = op.ScanRegistration(orc, imuHistorySize=50), op.RefScanRegistration(imuHistorySize=50)
    rng = np.random.default_rng(9)
    t_imu, k = 0.0, 0
    for sweep in range(4):
        t_scan = 0.125 * (sweep + 1)
        while t_imu < t_scan + 0.11:
            args = (t_imu, 0.02 * np.sin(3 * t_imu), 0.015 * np.cos(2 * t_imu), 0.4 * t_imu + (6.2 if k % 41 == 20 else 0.0),
                    (0.8 * np.sin(5 * t_imu), 0.1, -0.5 * np.cos(4 * t_imu)))
            o.update_imu(*args)
            r.update_imu(*args)
            t_imu += 0.001953125            # 2^-9 s: exact in double, and an integer number of nanoseconds ... / 2 (rounded by the reference's Time)
            k += 1
        sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=40 + sweep, az_steps=300)
        raw = synth.to_raw(sw)
        # the kept points in FIRING order with their exact relTimes: the reference projects them one by one (:231), then
        # they are split into rings (stable) as MultiScanRegistration::process does
        pts_f, ring_f, rel_f = op.multiscan_trace(orc, raw, "VLP-16")
        proj = r.project(pts_f, rel_f)
        idx = np.argsort(ring_f, kind="stable")
        proj_ref, rs = proj[idx], np.bincount(ring_f, minlength=16)
        res_o = o.process_raw(raw, t_scan, "VLP-16")
        res_r = r.process(proj_ref, rs, scan_time=t_scan)
        assert np.array_equal(res_o["full"], proj_ref), sweep                  # bit for bit
        assert np.array_equal(res_o["imu_trans"], res_r["imu_trans"]), sweep
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            assert np.array_equal(res_o[name], res_r[name]), name
    assert np.abs(res_r["imu_trans"]).max() > 1e-3


# ---------------------------------------------------------------------------------------------------------------------------
# Odometry and mapping: the reference's own BasicLaserOdometry.cpp / BasicLaserMapping.cpp (+ its vendored nanoflann) compiled
# against oracle/ref_stubs.  The Eigen stand-in forwards the matrix product, colPivHouseholderQr, the self-adjoint eigen
# solver and the 6x6 inverse to the oracle's restatements and pcl::VoxelGrid to the oracle's voxel grid, so bit-for-bit
# agreement here pins every statement of the two translation units and NOT those five third-party operations.
needs_od = pytest.mark.skipif(not op.RefLaserOdometry.available(), reason="oracle/_ref/libref_odometry.so not built")
needs_mp = pytest.mark.skipif(not op.RefLaserMapping.available(), reason="oracle/_ref/libref_mapping.so not built")


def _sweeps(world, sensor, n, az_steps, step=1.0, yaw=0.5):
    poses = synth.trajectory(n, step=step, yaw_step_deg=yaw)
    return [synth.make_sweep(world, sensor, poses[k], poses[k + 1], seed=100 + k, az_steps=az_steps) for k in range(n)]


@needs_od
@pytest.mark.parametrize("sensor,az,cfg", [("VLP-16", 900, {}), ("HDL-32", 1024, dict(maxIterations=7, deltaTAbort=0.3, deltaRAbort=0.2)),
                                           ("VLP-16", 600, dict(scanPeriod=0.05))])
def test_odometry_equals_the_reference(orc, small_world, sensor, az, cfg):
    sr = op.ScanRegistration(orc, **({"scanPeriod": cfg["scanPeriod"]} if "scanPeriod" in cfg else {}))
    o, r = op.LaserOdometry(orc, **cfg), op.RefLaserOdometry(**cfg)
    rng = np.random.default_rng(3)
    iters = []
    for k, sw in enumerate(_sweeps(small_world, sensor, 6, az, step=1.5, yaw=1.0)):
        f = sr.process(sw.points, sw.ring_sizes)
        if k >= 3:                                                              # the IMU plug-in path (updateIMU, :181-193; pluginIMURotation)
            t12 = np.concatenate([rng.uniform(-0.02, 0.02, 6), rng.uniform(-0.05, 0.05, 6)]).astype(np.float32)
            o.update_imu(t12)
            r.update_imu(t12)
        # the reference scans the PREVIOUS corner cloud up to the CURRENT sharp count (:262) and reads past its end when the
        # previous cloud is the shorter one (undefined behaviour; the oracle stops at the end): keep the inputs where it is defined
        assert k == 0 or len(f["sharp"]) <= len(o.last_corner())
        o.set_features(f)
        r.set_features(f)
        o.process()
        r.process()
        assert np.array_equal(o.transform, r.transform), k
        assert np.array_equal(o.transform_sum, r.transform_sum), k
        assert np.array_equal(o.last_corner(), r.last_corner()) and np.array_equal(o.last_surf(), r.last_surf()), k
        assert np.array_equal(o.full_to_end(), r.full_to_end()), k
        iters.append(o.stats()["iterations"])
    assert max(iters) > 3 and np.abs(o.transform[3:]).max() > 0.5             # the optimisation ran and found the motion


@needs_od
def test_odometry_from_a_poor_seed_equals_the_reference(orc, small_world):
    """a start far from the optimum: many iterations, re-association every fifth (:247), the distance weights after the
    fifth (:346-349, :460-463).  (removeNaNFromPointCloud, :230 / :252, is not restated by the oracle: the scan registration
    never emits non-finite points, MultiScanRegistration.cpp:187-191, and the C-ABI requires finite input.)"""
    sr, o, r = op.ScanRegistration(orc), op.LaserOdometry(orc), op.RefLaserOdometry()
    for k, sw in enumerate(_sweeps(small_world, "VLP-16", 3, 900)):
        f = sr.process(sw.points, sw.ring_sizes)
        for od in (o, r):
            od.set_features(f)
            if k == 2:
                od.set_transform([0.02, -0.03, 0.01, 0.4, -0.2, -0.3])
            od.process()
        assert np.array_equal(o.transform, r.transform), k
        assert np.array_equal(o.transform_sum, r.transform_sum), k
        assert np.array_equal(o.last_corner(), r.last_corner()), k
    assert o.stats()["iterations"] > 10


@needs_od
def test_non_finite_system_takes_the_reset_branch(orc, small_world):
    """BasicLaserOdometry.cpp:606-612.  Duplicated tripod points (the same corner on two adjacent rings of the previous cloud) make
    l12 = 0: those features' rows are NaN and — NaN != 0 — selected (:330-361), the 6x6 system is NaN throughout, Eigen's pivoted
    QR (maxima seeded with the first coefficient: a NaN survives, no pivot is declared negligible) answers NaN, and the reference
    zeroes every non-finite transform component — in each of the first five iterations.  The oracle's
    restatement of the QR follows those semantics; the reference's own translation unit (over the stand-in that forwards to it)
    takes the branch."""
    poses = synth.trajectory(2)
    sws = [synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=100 + k, az_steps=900) for k in range(2)]   # (the input of tests/test_gpu_next.py)
    sr = op.ScanRegistration(orc)
    f0, f1 = sr.process(sws[0].points, sws[0].ring_sizes), sr.process(sws[1].points, sws[1].ring_sizes)
    ls = f0["less_sharp"]
    dup = ls.copy()
    dup[:, 3] += 1.0
    both = np.concatenate([ls, dup])
    f0d = dict(f0)
    f0d["less_sharp"] = both[np.argsort(np.floor(both[:, 3]), kind="stable")]
    seed = np.float32([0.001, 0.002, -0.001, 0.05, 0.0, -0.3])
    o, r, z = op.LaserOdometry(orc), op.RefLaserOdometry(), op.LaserOdometry(orc)
    for od, start in ((o, seed), (r, seed), (z, np.zeros(6, np.float32))):
        od.set_features(f0d)
        od.process()
        od.set_features(f1)
        od.set_transform(start)
        od.process()
    assert np.array_equal(o.transform, r.transform) and np.array_equal(o.transform_sum, r.transform_sum)
    # the seed was wiped in the first iteration: the run is the run that starts from zero.  (The distance weights that set in with
    # the sixth iteration, :346-349, turn the infinite distances of the NaN rows into negative weights and deselect them, so the
    # iterations after the fifth are ordinary ones.)
    assert np.array_equal(o.transform, z.transform) and not np.array_equal(o.transform, seed)
    assert 5 < o.stats()["iterations"] <= 25 and np.all(np.isfinite(o.transform))


@needs_od
def test_reset_branch_through_collinear_tripods(orc, small_world):
    """BasicLaserOdometry.cpp:606-612 once more, without any distance tie (the device can be compared on this one,
    tests/test_gpu_next.py): the previous surface cloud is a straight line, so every plane tripod is collinear, its normal 0 / 0 = NaN
    (:437-470), the rows selected while s = 1 (iterations 0-4), the system non-finite, the transform reset five times; from the sixth
    iteration on the NaN weights deselect those rows and the edge rows alone carry the run."""
    poses = synth.trajectory(2)
    sws = [synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=100 + k, az_steps=900) for k in range(2)]
    sr = op.ScanRegistration(orc)
    f0, f1 = sr.process(sws[0].points, sws[0].ring_sizes), sr.process(sws[1].points, sws[1].ring_sizes)
    f0d = dict(f0)
    f0d["less_flat"] = conftest.collinear_previous_surf(f1)
    seed = np.float32([0.001, 0.002, -0.001, 0.05, 0.0, -0.3])
    o, r, z = op.LaserOdometry(orc), op.RefLaserOdometry(), op.LaserOdometry(orc)
    for od, start in ((o, seed), (r, seed), (z, np.zeros(6, np.float32))):
        od.set_features(f0d)
        od.process()
        od.set_features(f1)
        od.set_transform(start)
        od.process()
    assert np.array_equal(o.transform, r.transform) and np.array_equal(o.transform_sum, r.transform_sum)
    assert np.array_equal(o.transform, z.transform) and not np.array_equal(o.transform, seed)      # the seed was wiped: the reset was taken
    assert o.stats()["iterations"] > 5 and o.stats()["sel"] >= 10 and np.all(np.isfinite(o.transform))


def _run_mapping(orc, world, sensor, az, n, cfg, offset=(0.0, 0.0, 0.0), imu=False):
    sr, od = op.ScanRegistration(orc), op.LaserOdometry(orc)
    o, r = op.LaserMapping(orc, **cfg), op.RefLaserMapping(**cfg)
    rng = np.random.default_rng(9)
    fresh = 0
    for k, sw in enumerate(_sweeps(world, sensor, n, az)):
        od.set_features(sr.process(sw.points, sw.ring_sizes))
        od.process()
        tsum = od.transform_sum + np.array([0, 0, 0, *offset], np.float32)     # the odometry frame is arbitrary: a far-away origin
        args = (od.last_corner(), od.last_surf(), od.full_to_end(), tsum)
        samples = rng.uniform(-0.01, 0.01, (12, 2)).astype(np.float32)        # 100 Hz roll / pitch around the sweep's time stamp
        for mp in (o, r):
            if imu:
                for j in range(12):
                    mp.update_imu(0.1 * k + 0.01 * j - 0.03, samples[j, 0], samples[j, 1])
                mp.set_time(0.1 * k + 0.021)
            mp.set_inputs(*args)
        assert o.process() == r.process(), k
        for which in ("aft", "bef", "tobe", "sum"):
            assert np.array_equal(o.transform(which), r.transform(which)), (k, which)
        for name in o.CLOUDS:
            assert np.array_equal(o.cloud(name), r.cloud(name)), (k, name)
        assert o.has_fresh_map() == r.has_fresh_map(), k
        fresh += o.has_fresh_map()
    return o, r, fresh


@needs_mp
@pytest.mark.parametrize("sensor,az,cfg", [("VLP-16", 900, {}), ("HDL-32", 512, dict(maxIterations=4, deltaTAbort=0.2, deltaRAbort=0.2, cornerLeaf=0.3, surfLeaf=0.6))])
def test_mapping_equals_the_reference(orc, small_world, sensor, az, cfg):
    o, r, fresh = _run_mapping(orc, small_world, sensor, az, 7, cfg)
    assert fresh == 2                                                           # every 5th frame publishes a map (:245-249)
    assert o.stats()["optimized"] == 1 and len(o.cloud("surf_cubes")) > 5000


@needs_mp
def test_mapping_rolling_window_equals_the_reference(orc, small_world):
    """an origin 373 m / -127 m away: the cube window shifts several cubes on the first frame and again when the sensor
    crosses the next boundary (:300-445), with negative coordinates on one axis (:290-292)"""
    o, r, _ = _run_mapping(orc, small_world, "VLP-16", 600, 6, {}, offset=(-127.0, 3.0, 372.0))
    c = r.grid_center()
    assert tuple(c) != (10, 5, 10)
    assert len(o.cloud("corner_cubes")) > 300


@needs_mp
def test_mapping_imu_blend_equals_the_reference(orc, small_world):
    """updateIMU + the roll / pitch blend of transformUpdate (:161-190) interpolated at the sweep's time stamp"""
    o, r, _ = _run_mapping(orc, small_world, "VLP-16", 600, 4, {}, imu=True)
    o2, _, _ = _run_mapping(orc, small_world, "VLP-16", 600, 4, {})
    assert not np.array_equal(o.transform("aft"), o2.transform("aft"))          # the blend took part


# ---------------------------------------------------------------------------------------------------------------------------
# Sweep ingestion: the reference's own MultiScanRegistration.cpp (ring binning, start / end orientation, half-sweep logic,
# relative times, IMU projection) + BasicScanRegistration.cpp behind inert ROS stand-ins.
needs_ms = pytest.mark.skipif(not op.RefMultiScanRegistration.available(), reason="oracle/_ref/libref_multiscan.so not built")


@needs_ms
@pytest.mark.parametrize("sensor,az", [("VLP-16", 600), ("HDL-32", 512), ("HDL-64E", 512)])
def test_raw_ingestion_equals_the_reference(orc, small_world, sensor, az):
    o, r = op.ScanRegistration(orc), op.RefMultiScanRegistration(sensor)
    poses = synth.trajectory(3, yaw_step_deg=3.0)
    for k in range(3):
        sw = synth.make_sweep(small_world, sensor, poses[k], poses[k + 1], seed=60 + k, az_steps=az)
        raw = synth.to_raw(sw, bad_every=53)                                   # NaN, zero and out-of-field returns in the packet
        if k == 2:
            raw = np.roll(raw, 7 * len(sw.ring_sizes) + 3, axis=0)              # the sweep starts mid-firing at another azimuth
        a, b = o.process_raw(raw, 0.1 * k, sensor), r.process_raw(raw, 0.1 * k)
        for name in ("ring_sizes", "full", "sharp", "less_sharp", "flat", "less_flat", "imu_trans"):
            assert np.array_equal(a[name], b[name]), (k, name)
        assert len(a["sharp"]) > 50 and a["ring_sizes"].sum() < len(raw)


@needs_ms
def test_raw_ingestion_with_imu_equals_the_reference(orc, small_world):
    o, r = op.ScanRegistration(orc), op.RefMultiScanRegistration("VLP-16")
    rng = np.random.default_rng(21)
    tick = 1.0 / 512                          # exact in double seconds (the oracle's clock) AND in nanoseconds (the reference's)
    for k in range(3):
        t_scan = 51 * k * tick
        for j in range(14):                                                     # ~100 Hz states overlapping the sweep on both sides
            s = (t_scan + (5 * j - 10) * tick, *rng.uniform(-0.03, 0.03, 3), rng.uniform(-0.5, 0.5, 3))
            o.update_imu(*s)
            r.update_imu(*s)
        sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=80 + k, az_steps=400)
        raw = synth.to_raw(sw, bad_every=41)
        a, b = o.process_raw(raw, t_scan, "VLP-16"), r.process_raw(raw, t_scan)
        for name in ("ring_sizes", "full", "sharp", "less_sharp", "flat", "less_flat", "imu_trans"):
            assert np.array_equal(a[name], b[name]), (k, name)
    assert np.abs(b["imu_trans"]).max() > 1e-3


@needs_ms
def test_ring_mapping_equals_the_reference(orc):
    """getRingForAngle (:64-66) through process(): one return per vertical angle on a dense sweep of angles incl. the
    half-spacing boundaries (and their float neighbours) and angles outside the field of view — kept / dropped and binned alike"""
    for sensor in ("VLP-16", "HDL-32", "HDL-64E"):
        lo, hi, nr = op.MAPPERS[sensor]
        step = (hi - lo) / (nr - 1)
        edges = np.float32(lo + step * (np.arange(-2, nr + 2) + 0.5))
        deg = np.concatenate([np.linspace(lo - 2 * step, hi + 2 * step, 4001), edges, np.nextafter(edges, np.float32(-1e9)),
                              np.nextafter(edges, np.float32(1e9))]).astype(np.float64)
        az = np.linspace(0.0, 2 * np.pi, len(deg), endpoint=False)              # one turn, so that the relative times are defined
        raw = np.stack([np.cos(np.deg2rad(deg)) * np.cos(az) * 10, np.cos(np.deg2rad(deg)) * np.sin(az) * 10, np.sin(np.deg2rad(deg)) * 10], axis=1).astype(np.float32)
        a, b = op.ScanRegistration(orc).process_raw(raw, 0.0, sensor), op.RefMultiScanRegistration(sensor).process_raw(raw, 0.0)
        assert np.array_equal(a["ring_sizes"], b["ring_sizes"])
        assert np.array_equal(a["full"], b["full"])
        assert 0 < a["ring_sizes"].sum() < len(raw) and a["ring_sizes"].min() > 0
        # angles below the lowest ring by less than one spacing still land in ring 0 (int() truncates towards zero)
        assert a["ring_sizes"][0] > 1.3 * a["ring_sizes"][1]


@needs_ms
def test_startup_delay_of_the_message_handler():
    """handleCloudMessage drops the first 20 messages (:143-149, _systemDelay); ros::Time -> Time keeps the nanoseconds"""
    r = op.RefMultiScanRegistration("VLP-16")
    raw = np.array([[10.0, 0.0, 0.0], [0.0, 10.0, 0.0], [-10.0, 0.0, 0.0], [0.0, -10.0, 0.0]], np.float32)
    got = [r.handle_message(raw, 5, 1000 * k) is not None for k in range(23)]
    assert got == [False] * 20 + [True] * 3


# ---------------------------------------------------------------------------------------------------------------------------
# The committed golden fixtures (tests/golden/*.npz, generated from the oracle by make_golden.py and used by the -m gpu parity
# tests on the GPU box, where neither /root/reference nor these libraries' sources exist) reproduced by the REFERENCE'S OWN
# code: what the device path is compared with on the GPU box is what the reference computes, not only what the oracle says.
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_all = pytest.mark.skipif(not (op.RefScanRegistration.available() and op.RefLaserOdometry.available() and op.RefLaserMapping.available()),
                               reason="oracle/_ref libraries not built")


@needs_all
def test_golden_features_are_the_reference_s():
    g = np.load(os.path.join(GOLD, "features_vlp16.npz"))
    f = op.RefScanRegistration().process(g["points"], g["ring_sizes"])
    for name in ("sharp", "less_sharp", "flat", "less_flat"):
        assert np.array_equal(f[name], g[name]), name


@needs_all
def test_golden_pipeline_is_the_reference_s():
    """feature extraction -> odometry -> registration against a frozen map, 2 streams x 4 sweeps"""
    g = np.load(os.path.join(GOLD, "pipeline_vlp16.npz"))
    for s in range(2):
        sr, od, mp = op.RefScanRegistration(), op.RefLaserOdometry(), op.RefLaserMapping()
        mp.set_frozen(g["corner_map"], g["surf_map"])
        mp.set_transform("aft", g[f"start_{s}"])
        for t in range(4):
            od.set_features(sr.process(g[f"points_{s}_{t}"], g[f"rings_{s}_{t}"]))
            od.process()
            if t > 0:
                mp.set_transform("sum", od.transform_sum)
                mp.register_frozen(od.last_corner(), od.last_surf(), mp.associate())
            assert np.array_equal(od.transform_sum, g[f"sum_{s}"][t]), (s, t)
            assert np.array_equal(mp.transform("aft"), g[f"aft_{s}"][t]), (s, t)


@needs_all
def test_golden_live_map_sequence_is_the_reference_s():
    """five sweeps through process() with the rolling live map; the recorded prior and posterior states of the last two"""
    g = np.load(os.path.join(GOLD, "mapping_seq_vlp16.npz"))
    world = synth.World(half_extent=45.0)
    poses = synth.trajectory(5, start=(0.0, 0.0, 0.0))
    sr, od, mp = op.RefScanRegistration(), op.RefLaserOdometry(), op.RefLaserMapping()
    for t in range(5):
        sw = synth.make_sweep(world, "VLP-16", poses[t], poses[t + 1], seed=900 + t, az_steps=600)      # make_golden.py part 3
        od.set_features(sr.process(sw.points, sw.ring_sizes))
        od.process()
        full_end = od.full_to_end()
        lc, ls, ts = od.last_corner(), od.last_surf(), od.transform_sum
        if t >= 3:
            assert np.array_equal(mp.cloud("corner_cubes"), g[f"pre_corner_cubes_{t}"])
            assert np.array_equal(mp.cloud("surf_cubes"), g[f"pre_surf_cubes_{t}"])
            assert np.array_equal(mp.transform("aft"), g[f"pre_aft_{t}"]) and np.array_equal(mp.transform("bef"), g[f"pre_bef_{t}"])
            assert np.array_equal(lc, g[f"corner_last_{t}"]) and np.array_equal(ls, g[f"surf_last_{t}"])
            assert np.array_equal(full_end[::8], g[f"full_{t}"]) and np.array_equal(ts, g[f"sum_{t}"])
        mp.set_inputs(lc, ls, full_end, ts)
        mp.process()
        if t >= 3:
            assert np.array_equal(mp.transform("aft"), g[f"post_aft_{t}"]) and np.array_equal(mp.transform("bef"), g[f"post_bef_{t}"])
            assert np.array_equal(mp.cloud("full_res")[::8], g[f"post_full_{t}"])
            assert len(mp.cloud("corner_cubes")) == int(g[f"post_n_corner_{t}"]) and len(mp.cloud("surf_cubes")) == int(g[f"post_n_surf_{t}"])


@needs_all
def test_frozen_map_registration_equals_the_reference(orc, small_world):
    """the unit of the batched mode — one sweep against a caller-provided sub-map — for a denser sensor and a larger map,
    from a perturbed start (several Gauss-Newton iterations, both residual kinds, the degeneracy check)"""
    corner_map, surf_map = small_world.make_map(120000)
    gt = np.array([0.004, 0.3, -0.002, 2.0, 0.02, -3.0])
    sw = synth.make_sweep(small_world, "HDL-32", gt, gt, seed=33, az_steps=1024)
    f = op.ScanRegistration(orc).process(sw.points, sw.ring_sizes)
    o, r = op.LaserMapping(orc), op.RefLaserMapping()
    o.set_frozen(corner_map, surf_map)
    r.set_frozen(corner_map, surf_map)
    guess = gt + np.array([0.006, -0.005, 0.004, 0.08, -0.06, 0.09])
    po, pr = o.register_frozen(f["less_sharp"], f["less_flat"], guess), r.register_frozen(f["less_sharp"], f["less_flat"], guess)
    assert np.array_equal(po, pr)
    assert np.array_equal(o.transform("aft"), r.transform("aft"))
    for name in ("corner_stack_ds", "surf_stack_ds"):
        assert np.array_equal(o.cloud(name), r.cloud(name)), name
    assert o.stats()["iterations"] >= 3 and np.abs(po[3:] - gt[3:]).max() < 0.03


# ---------------------------------------------------------------------------------------------------------------------------
# The silent guards and the degeneracy handling (SURVEY.md §8c checklist 5-7), oracle vs the compiled reference.
def _corridor():
    """two endless walls along z, ground and ceiling: nothing constrains the motion along the corridor"""
    w = synth.World(half_extent=400.0, pitch=1000.0)
    w.boxes = np.array([[-6.0, -5.0, -399.0, 399.0], [5.0, 6.0, -399.0, 399.0]])
    return w


@needs_all
def test_degenerate_corridor_equals_the_reference(orc):
    """isDegenerate: eigenvalues of AtA below 10 (odometry, :561-597) / 100 (mapping, :869-905) on the first iteration, the
    projector built from the ROW-zeroed eigenvector matrix and reused by the later iterations"""
    w = _corridor()
    poses = synth.trajectory(5, step=1.0, yaw_step_deg=0.0)
    sr, od, ro, mp, rm = op.ScanRegistration(orc), op.LaserOdometry(orc), op.RefLaserOdometry(), op.LaserMapping(orc), op.RefLaserMapping()
    degenerate = 0
    for k in range(5):
        sw = synth.make_sweep(w, "VLP-16", poses[k], poses[k + 1], seed=k, az_steps=900)
        f = sr.process(sw.points, sw.ring_sizes)
        for o in (od, ro):
            o.set_features(f)
            o.process()
        assert np.array_equal(od.transform, ro.transform) and np.array_equal(od.transform_sum, ro.transform_sum), k
        args = (od.last_corner(), od.last_surf(), od.full_to_end(), od.transform_sum)
        for m in (mp, rm):
            m.set_inputs(*args)
            m.process()
        for which in ("aft", "bef", "tobe"):
            assert np.array_equal(mp.transform(which), rm.transform(which)), (k, which)
        degenerate += mp.stats()["degenerate"]
    assert degenerate >= 3
    assert abs(od.transform[5]) < 0.2                                           # 1 m per sweep along the corridor went unobserved


@needs_od
def test_odometry_guards_equal_the_reference(orc, small_world):
    """(a) fewer than 10 selected rows -> the iteration is skipped and the motion estimate stays (:485-488): the previous sweep
    is 60 m away, no neighbour within 5 m; (b) previous clouds of <= 10 corner / <= 100 surface points -> no optimisation at
    all, the pose is still accumulated (:222)"""
    sw = _sweeps(small_world, "VLP-16", 2, 900)
    sr = op.ScanRegistration(orc)
    f0, f1 = sr.process(sw[0].points, sw[0].ring_sizes), sr.process(sw[1].points, sw[1].ring_sizes)
    far = {n: v + np.array([60.0, 0, 60.0, 0], np.float32) for n, v in f1.items()}
    tiny = {n: v[:8] if n in ("sharp", "less_sharp") else v[:90] for n, v in f0.items()}
    for first, second in ((f0, far), (tiny, f1)):
        o, r = op.LaserOdometry(orc), op.RefLaserOdometry()
        for od in (o, r):
            od.set_features(first)
            od.process()
            od.set_features(second)
            od.set_transform([0.001, 0.002, -0.001, 0.05, 0.0, -0.3])
            od.process()
        assert np.array_equal(o.transform, r.transform) and np.array_equal(o.transform_sum, r.transform_sum)
        assert np.array_equal(o.transform, np.float32([0.001, 0.002, -0.001, 0.05, 0.0, -0.3]))      # untouched by the guards
        assert np.abs(o.transform_sum).max() > 0.1                                                    # but accumulated


@needs_mp
def test_mapping_guards_equal_the_reference(orc, small_world):
    """(a) a sub-map of <= 10 corner or <= 100 surface points: optimisation AND transformUpdate are skipped (:628-629), Bef / Aft
    keep their stale values; (b) fewer than 50 selected rows: the iteration is skipped (:826-828) — the sweep sits 40 m above its map"""
    sw = _sweeps(small_world, "VLP-16", 2, 900)
    sr, od = op.ScanRegistration(orc), op.LaserOdometry(orc)
    od.set_features(sr.process(sw[0].points, sw[0].ring_sizes))
    od.process()
    lc, ls, full = od.last_corner(), od.last_surf(), od.full_to_end()
    for case in ("sparse", "far"):
        o, r = op.LaserMapping(orc), op.RefLaserMapping()
        for m in (o, r):
            if case == "sparse":
                m.set_inputs(lc[:9], ls[:95], full, np.zeros(6, np.float32))
                m.process()                                                    # frame 0 only inserts; its map stays below the guard
                m.set_inputs(lc, ls, full, np.float32([0, 0.01, 0, 0.1, 0, 0.5]))
            else:
                m.set_inputs(lc, ls, full, np.zeros(6, np.float32))
                m.process()
                m.set_inputs(lc, ls, full, np.float32([0, 0, 0, 0, 40.0, 0]))
            m.process()
        for which in ("aft", "bef", "tobe", "sum"):
            assert np.array_equal(o.transform(which), r.transform(which)), (case, which)
        for name in ("corner_cubes", "surf_cubes"):
            assert np.array_equal(o.cloud(name), r.cloud(name)), (case, name)
        if case == "sparse":
            assert o.stats()["optimized"] == 0 and np.array_equal(o.transform("aft"), np.zeros(6, np.float32))
        else:
            assert o.stats()["optimized"] == 1 and o.stats()["sel"] < 50


# ---------------------------------------------------------------------------------------------------------------------------
# How much can the five UNPINNED third-party operations matter?  The same reference translation units built over the
# alternative arithmetic of the Eigen stand-in (-DREF_STUB_ALT_ARITH: double accumulation, normal equations / elimination,
# double Jacobi) — if the poses barely move between the two builds, the real Eigen's rounding cannot move them much either.
needs_alt = pytest.mark.skipif(not (op.RefLaserMappingAlt.available() and op.RefNodes.available(os.path.join(os.path.dirname(op.__file__), "_ref", "libref_nodes_alt.so"))),
                               reason="oracle/_ref/*_alt.so not built")


@needs_alt
def test_third_party_arithmetic_sensitivity_frozen_map(orc, small_world):
    """the benchmark's unit of work (one sweep against a frozen sub-map): the pose moves by a few 1e-6 at most"""
    corner_map, surf_map = small_world.make_map(120000)
    worst = 0.0
    for trial in range(3):
        rng = np.random.default_rng(trial)
        gt = np.array([0.004, 0.3, -0.002, 2.0, 0.02, -3.0]) + rng.uniform(-1, 1, 6) * [0.01, 0.2, 0.01, 3, 0.02, 3]
        sw = synth.make_sweep(small_world, "HDL-32", gt, gt, seed=33 + trial, az_steps=1024)
        f = op.ScanRegistration(orc).process(sw.points, sw.ring_sizes)
        guess = gt + rng.uniform(-1, 1, 6) * [0.006, 0.006, 0.006, 0.08, 0.08, 0.08]
        res = []
        for cls in (op.RefLaserMapping, op.RefLaserMappingAlt):
            m = cls()
            m.set_frozen(corner_map, surf_map)
            res.append(m.register_frozen(f["less_sharp"], f["less_flat"], guess))
        worst = max(worst, float(np.abs(res[0] - res[1]).max()))
    print(f"frozen-map registration, pose change under the alternative third-party arithmetic: {worst:.2e}")
    assert worst < 1e-5


@pytest.mark.skipif(not (op.RefLaserOdometry.available() and op.RefLaserOdometryAlt.available()), reason="oracle/_ref/libref_odometry_alt.so not built")
def test_third_party_arithmetic_sensitivity_odometry(orc, small_world):
    """the odometry's unit of work (one sweep against the previous one): under the alternative third-party arithmetic (products accumulated
    in double, the 6x6 solve by elimination) the optimised transform of a sweep moves (3e-7 on these small VLP-16 sweeps, up to 4e-5 on HDL-64E ones) — the kind of difference between
    the device (double row sums) and the oracle (float, row by row) that the long-chain envelope of bench.py is about (round 6)"""
    poses = synth.trajectory(6)
    plain, alt = op.RefLaserOdometry(), op.RefLaserOdometryAlt()
    sr = op.ScanRegistration(orc)
    worst = 0.0
    for k in range(6):
        sw = synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=300 + k, az_steps=900)
        f = sr.process(sw.points, sw.ring_sizes)
        for o in (plain, alt):
            o.set_features(f)
            o.process()
        worst = max(worst, float(np.abs(plain.transform - alt.transform).max()))
    print(f"odometry, per-sweep transform change under the alternative third-party arithmetic: {worst:.2e}")
    assert 0.0 < worst < 1e-4


@needs_alt
def test_third_party_arithmetic_sensitivity_node_graph(small_world):
    """the four-node pipeline: the odometry stays within 1e-4; the LIVE-MAP poses are reported, not bounded tightly — the rolling
    map feeds rounding back through its voxel grids, which is why the device path is compared with the oracle
    from identical prior states and in frozen-map mode"""
    alt = os.path.join(os.path.dirname(op.__file__), "_ref", "libref_nodes_alt.so")
    a, b = op.RefNodes("VLP-16"), op.RefNodes("VLP-16", lib=alt)
    poses = synth.trajectory(9)
    for k in range(9):
        sw = synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=200 + k, az_steps=900)
        raw = synth.to_raw(sw)
        ns = (51 * k + 5) * 1953125
        for n in (a, b):
            n.push_cloud(raw, 1000 + ns // 10**9, ns % 10**9)
    d = {t: float(np.abs(a.odometry(t)[1] - b.odometry(t)[1]).max()) for t in a.TOPICS}
    print("node graph, change under the alternative third-party arithmetic:", d)
    assert d["/laser_odom_to_init"] < 1e-4
    assert d["/aft_mapped_to_init"] < 5e-2 and d["/integrated_to_init"] < 5e-2
