"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle*.so (the CPU restatement of the reference path).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    need = force or not all(os.path.exists(os.path.join(_HERE, f)) for f in ("liboracle.so", "liboracle_fast.so"))
    if need or os.path.exists("/root/reference"):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _pts(a):
    a = _f32(a)
    if a.ndim == 1:
        a = a.reshape(-1, 4)
    assert a.shape[1] == 4
    return a


class Oracle:
    def __init__(self, fast: bool = False):
        path = os.path.join(_HERE, "liboracle_fast.so" if fast else "liboracle.so")
        if not os.path.exists(path):
            build()
        L = self.L = C.CDLL(path)
        vp = C.c_void_p
        L.orc_voxel_grid.restype = C.c_int
        for name in ("orc_scanreg_create", "orc_odom_create", "orc_map_create"):
            getattr(L, name).restype = vp
        for name in ("orc_scanreg_get", "orc_odom_get_cloud", "orc_map_get_cloud", "orc_map_process",
                     "orc_map_has_fresh_map", "orc_map_residual_pass", "orc_inv6", "orc_degeneracy"):
            getattr(L, name).restype = C.c_int

    # ---- primitives
    def voxel_grid(self, pts, leaf):
        pts = _pts(pts)
        out = np.zeros((max(len(pts), 1), 4), np.float32)
        n = self.L.orc_voxel_grid(pts.ctypes.data_as(C.c_void_p), len(pts), C.c_float(leaf),
                                  out.ctypes.data_as(C.c_void_p), len(out))
        return out[:n].copy()

    def knn(self, pts, queries, k, brute=False):
        pts, q = _pts(pts), _pts(queries)
        idx = np.zeros((len(q), k), np.int32)
        d2 = np.zeros((len(q), k), np.float32)
        self.L.orc_knn(pts.ctypes.data_as(C.c_void_p), len(pts), q.ctypes.data_as(C.c_void_p), len(q), k,
                       idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p), 1 if brute else 0)
        return idx, d2

    def eig(self, A):
        A = _f32(A)
        n = A.shape[0]
        w = np.zeros(n, np.float32)
        V = np.zeros((n, n), np.float32)
        fn = self.L.orc_eig3 if n == 3 else self.L.orc_eig6
        fn(A.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
        return w, V

    def qr_solve(self, A, b):
        A, b = _f32(A), _f32(b)
        x = np.zeros(A.shape[1], np.float32)
        fn = self.L.orc_qr53 if A.shape == (5, 3) else self.L.orc_qr66
        fn(A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
        return x

    def inv6(self, A):
        A = _f32(A)
        out = np.zeros((6, 6), np.float32)
        ok = self.L.orc_inv6(A.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out, bool(ok)

    def degeneracy(self, AtA, thr):
        AtA = _f32(AtA)
        P = np.zeros((6, 6), np.float32)
        d = self.L.orc_degeneracy(AtA.ctypes.data_as(C.c_void_p), C.c_float(thr), P.ctypes.data_as(C.c_void_p))
        return bool(d), P

    def rotate_zxy(self, p, rz, rx, ry):
        p = _f32(p).copy()
        self.L.orc_rotate_zxy(p.ctypes.data_as(C.c_void_p), C.c_float(rz), C.c_float(rx), C.c_float(ry))
        return p


def _get_cloud(fn, h, which, cap_hint=0):
    n = fn(h, which, None, 0)
    out = np.zeros((max(n, 1), 4), np.float32)
    fn(h, which, out.ctypes.data_as(C.c_void_p), n)
    return out[:n].copy()


MAPPERS = {"VLP-16": (-15.0, 15.0, 16), "HDL-32": (-30.67, 10.67, 32), "HDL-64E": (-24.9, 2.0, 64)}   # MultiScanRegistration.h:60-75


def multiscan_bin(orc: "Oracle", raw_xyz, mapper="VLP-16", scan_period=0.1):
    """oracle restatement of MultiScanRegistration::process: raw (n,3) sensor-axes points in firing order ->
    (binned points (m,4) with rings concatenated, ring_sizes (n_rings,))."""
    lo, hi, nr = MAPPERS[mapper] if isinstance(mapper, str) else mapper
    raw = np.ascontiguousarray(raw_xyz, np.float32).reshape(-1, 3)
    out = np.zeros((max(len(raw), 1), 4), np.float32)
    rs = np.zeros(nr, np.int32)
    orc.L.orc_multiscan_bin.restype = C.c_int
    m = orc.L.orc_multiscan_bin(raw.ctypes.data_as(C.c_void_p), len(raw), C.c_float(lo), C.c_float(hi), int(nr), C.c_float(scan_period),
                                out.ctypes.data_as(C.c_void_p), len(out), rs.ctypes.data_as(C.c_void_p))
    return out[:m].copy(), rs


def tm_associate(orc: "Oracle", transform_sum, bef_mapped, aft_mapped):
    """oracle restatement of BasicTransformMaintenance::transformAssociateToMap -> transformMapped (6,)."""
    s, b, a = _f32(transform_sum), _f32(bef_mapped), _f32(aft_mapped)
    out = np.zeros(6, np.float32)
    orc.L.orc_tm_associate(s.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def wire_pose_to_quat(orc: "Oracle", rot3):
    r = _f32(rot3)
    q = np.zeros(4, np.float64)
    orc.L.orc_wire_pose_to_quat(r.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p))
    return q


def wire_quat_to_pose(orc: "Oracle", q4):
    q = np.ascontiguousarray(q4, np.float64)
    r = np.zeros(3, np.float32)
    orc.L.orc_wire_quat_to_pose(q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
    return r


def multiscan_trace(orc: "Oracle", raw_xyz, mapper="VLP-16", scan_period=0.1):
    """kept points of a raw sweep in firing order before any IMU projection: (points (m,4), ring (m,), relTime (m,))"""
    lo, hi, nr = MAPPERS[mapper] if isinstance(mapper, str) else mapper
    raw = np.ascontiguousarray(raw_xyz, np.float32).reshape(-1, 3)
    out = np.zeros((max(len(raw), 1), 5), np.float32)
    ring = np.zeros(max(len(raw), 1), np.int32)
    orc.L.orc_multiscan_trace.restype = C.c_int
    m = orc.L.orc_multiscan_trace(raw.ctypes.data_as(C.c_void_p), len(raw), C.c_float(lo), C.c_float(hi), int(nr), C.c_float(scan_period),
                                  out.ctypes.data_as(C.c_void_p), ring.ctypes.data_as(C.c_void_p), len(out))
    return out[:m, :4].copy(), ring[:m].copy(), out[:m, 4].copy()


class ScanRegistration:
    """oracle restatement of BasicScanRegistration (IMU-less)."""
    NAMES = ("full", "sharp", "less_sharp", "flat", "less_flat")

    def __init__(self, orc: Oracle, **cfg):
        self.o = orc
        self.h = C.c_void_p(orc.L.orc_scanreg_create())
        c = dict(scanPeriod=0.1, nFeatureRegions=6, curvatureRegion=5, maxCornerSharp=2, maxSurfaceFlat=4,
                 lessFlatFilterSize=0.2, surfaceCurvatureThreshold=0.1)
        c.update(cfg)
        orc.L.orc_scanreg_config(self.h, C.c_float(c["scanPeriod"]), c["nFeatureRegions"], c["curvatureRegion"],
                                 c["maxCornerSharp"], c["maxSurfaceFlat"], C.c_float(c["lessFlatFilterSize"]),
                                 C.c_float(c["surfaceCurvatureThreshold"]))
        orc.L.orc_scanreg_config2(self.h, int(c.get("maxCornerLessSharp", 10 * c["maxCornerSharp"])), int(c.get("imuHistorySize", 200)))

    def __del__(self):
        self.o.L.orc_scanreg_destroy(self.h)

    def update_imu(self, stamp, roll, pitch, yaw, acc):
        """updateIMUData: stamp in seconds, acc = local acceleration with gravity removed (ScanRegistration.cpp:171-174)"""
        self.o.L.orc_scanreg_update_imu(self.h, C.c_double(stamp), C.c_float(roll), C.c_float(pitch), C.c_float(yaw),
                                        C.c_float(acc[0]), C.c_float(acc[1]), C.c_float(acc[2]))

    def process_raw(self, raw_xyz, scan_time, mapper="VLP-16"):
        """MultiScanRegistration::process(laserCloudIn, scanTime) with this handle's IMU state"""
        lo, hi, nr = MAPPERS[mapper] if isinstance(mapper, str) else mapper
        raw = np.ascontiguousarray(raw_xyz, np.float32).reshape(-1, 3)
        rs = np.zeros(nr, np.int32)
        self.o.L.orc_scanreg_process_raw(self.h, raw.ctypes.data_as(C.c_void_p), len(raw), C.c_double(scan_time), C.c_float(lo), C.c_float(hi),
                                         int(nr), rs.ctypes.data_as(C.c_void_p))
        res = {n: _get_cloud(self.o.L.orc_scanreg_get, self.h, k) for k, n in enumerate(self.NAMES)}
        res["ring_sizes"] = rs
        it = np.zeros(12, np.float32)
        self.o.L.orc_scanreg_get_imu_trans(self.h, it.ctypes.data_as(C.c_void_p))
        res["imu_trans"] = it
        return res

    def process(self, pts, ring_sizes):
        pts = _pts(pts)
        rs = np.ascontiguousarray(ring_sizes, np.int32)
        self.o.L.orc_scanreg_process(self.h, pts.ctypes.data_as(C.c_void_p), rs.ctypes.data_as(C.c_void_p), len(rs))
        return {n: _get_cloud(self.o.L.orc_scanreg_get, self.h, k) for k, n in enumerate(self.NAMES)}


class LaserOdometry:
    def __init__(self, orc: Oracle, scanPeriod=0.1, maxIterations=25, deltaTAbort=0.1, deltaRAbort=0.1):
        self.o = orc
        self.h = C.c_void_p(orc.L.orc_odom_create())
        orc.L.orc_odom_config(self.h, C.c_float(scanPeriod), maxIterations, C.c_float(deltaTAbort), C.c_float(deltaRAbort))

    def __del__(self):
        self.o.L.orc_odom_destroy(self.h)

    def set_features(self, f):
        for k, n in enumerate(ScanRegistration.NAMES):
            p = _pts(f[n])
            self.o.L.orc_odom_set_cloud(self.h, k, p.ctypes.data_as(C.c_void_p), len(p))

    def set_transform(self, t6):
        t6 = _f32(t6)
        self.o.L.orc_odom_set_transform(self.h, t6.ctypes.data_as(C.c_void_p))

    def update_imu(self, t12):
        t12 = _f32(t12)
        self.o.L.orc_odom_update_imu(self.h, t12.ctypes.data_as(C.c_void_p))

    def process(self):
        self.o.L.orc_odom_process(self.h)

    def _t(self, fn):
        t = np.zeros(6, np.float32)
        fn(self.h, t.ctypes.data_as(C.c_void_p))
        return t

    @property
    def transform(self):
        return self._t(self.o.L.orc_odom_get_transform)

    @property
    def transform_sum(self):
        return self._t(self.o.L.orc_odom_get_transform_sum)

    def last_corner(self):
        return _get_cloud(self.o.L.orc_odom_get_cloud, self.h, 0)

    def last_surf(self):
        return _get_cloud(self.o.L.orc_odom_get_cloud, self.h, 1)

    def full_to_end(self):
        self.o.L.orc_odom_transform_full_to_end(self.h)
        return _get_cloud(self.o.L.orc_odom_get_cloud, self.h, 2)

    def stats(self):
        s = np.zeros(3, np.int32)
        self.o.L.orc_odom_stats(self.h, s.ctypes.data_as(C.c_void_p))
        return dict(iterations=int(s[0]), sel=int(s[1]), frame=int(s[2]))


class LaserMapping:
    CLOUDS = ("full_res", "surround_ds", "corner_from_map", "surf_from_map", "corner_stack_ds", "surf_stack_ds",
              "corner_cubes", "surf_cubes")

    def __init__(self, orc: Oracle, scanPeriod=0.1, maxIterations=10, deltaTAbort=0.05, deltaRAbort=0.05,
                 cornerLeaf=0.2, surfLeaf=0.4):
        self.o = orc
        self.h = C.c_void_p(orc.L.orc_map_create())
        orc.L.orc_map_config(self.h, C.c_float(scanPeriod), maxIterations, C.c_float(deltaTAbort),
                             C.c_float(deltaRAbort), C.c_float(cornerLeaf), C.c_float(surfLeaf))

    def __del__(self):
        self.o.L.orc_map_destroy(self.h)

    def set_inputs(self, corner_last, surf_last, full_res, transform_sum):
        for k, p in enumerate((corner_last, surf_last, full_res)):
            p = _pts(p)
            self.o.L.orc_map_set_cloud(self.h, k, p.ctypes.data_as(C.c_void_p), len(p))
        t = _f32(transform_sum)
        self.o.L.orc_map_update_odometry(self.h, t.ctypes.data_as(C.c_void_p))

    def update_imu(self, stamp, roll, pitch):
        """updateIMU(IMUState2): stamp in seconds"""
        self.o.L.orc_map_update_imu(self.h, C.c_double(stamp), C.c_float(roll), C.c_float(pitch))

    def set_time(self, t):
        """the laserOdometryTime argument of process()"""
        self.o.L.orc_map_set_time(self.h, C.c_double(t))

    def process(self):
        return bool(self.o.L.orc_map_process(self.h))

    def transform(self, which="aft"):
        t = np.zeros(6, np.float32)
        self.o.L.orc_map_get_transform(self.h, ("aft", "bef", "tobe", "sum").index(which), t.ctypes.data_as(C.c_void_p))
        return t

    def set_transform(self, which, t6):
        t6 = _f32(t6)
        self.o.L.orc_map_set_transform(self.h, ("aft", "bef", "tobe", "sum").index(which), t6.ctypes.data_as(C.c_void_p))

    def cloud(self, name):
        return _get_cloud(self.o.L.orc_map_get_cloud, self.h, self.CLOUDS.index(name))

    def has_fresh_map(self):
        return bool(self.o.L.orc_map_has_fresh_map(self.h))

    def load_cubes(self, corner, surf):
        c, s = _pts(corner), _pts(surf)
        self.o.L.orc_map_load_cubes(self.h, c.ctypes.data_as(C.c_void_p), len(c), s.ctypes.data_as(C.c_void_p), len(s))

    def set_frozen(self, corner, surf):
        c, s = _pts(corner), _pts(surf)
        self.o.L.orc_map_set_frozen(self.h, c.ctypes.data_as(C.c_void_p), len(c), s.ctypes.data_as(C.c_void_p), len(s))

    def register_frozen(self, corner_last, surf_last, guess6):
        for k, p in enumerate((corner_last, surf_last)):
            p = _pts(p)
            self.o.L.orc_map_set_cloud(self.h, k, p.ctypes.data_as(C.c_void_p), len(p))
        g = _f32(guess6)
        pose = np.zeros(6, np.float32)
        self.o.L.orc_map_register_frozen(self.h, g.ctypes.data_as(C.c_void_p), pose.ctypes.data_as(C.c_void_p))
        return pose

    def associate(self):
        t = np.zeros(6, np.float32)
        self.o.L.orc_map_associate(self.h, t.ctypes.data_as(C.c_void_p))
        return t

    def residual_pass(self, pose6, cap=200000):
        p = _f32(pose6)
        ori = np.zeros((cap, 4), np.float32)
        co = np.zeros((cap, 4), np.float32)
        n = self.o.L.orc_map_residual_pass(self.h, p.ctypes.data_as(C.c_void_p), ori.ctypes.data_as(C.c_void_p),
                                           co.ctypes.data_as(C.c_void_p), cap)
        return ori[:n].copy(), co[:n].copy()

    def stats(self):
        s = np.zeros(8, np.int32)
        self.o.L.orc_map_stats(self.h, s.ctypes.data_as(C.c_void_p))
        keys = ("iterations", "sel", "corner_ds", "surf_ds", "corner_from_map", "surf_from_map", "degenerate", "optimized")
        return dict(zip(keys, (int(v) for v in s)))


def ref_knn(pts, queries, k):
    """kNN through the REFERENCE's own nanoflann.hpp (oracle/_ref); None when the shim is not built."""
    path = os.path.join(_HERE, "_ref", "libref_nanoflann.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    pts, q = _pts(pts), _pts(queries)
    idx = np.zeros((len(q), k), np.int32)
    d2 = np.zeros((len(q), k), np.float32)
    L.ref_knn(pts.ctypes.data_as(C.c_void_p), len(pts), q.ctypes.data_as(C.c_void_p), len(q), k,
              idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p))
    return idx, d2


def ref_small():
    """The REFERENCE's own BasicTransformMaintenance.cpp, math_utils.h, Angle.h, CircularBuffer.h compiled where they lie
    (oracle/_ref/libref_loam_small.so, oracle/Makefile target `ref`); None when the library is not built."""
    path = os.path.join(_HERE, "_ref", "libref_loam_small.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_rad2deg.restype = C.c_float
    L.ref_deg2rad.restype = C.c_float
    L.ref_circular.restype = C.c_int
    return L


class RefScanRegistration:
    """The REFERENCE's own BasicScanRegistration (src/lib/BasicScanRegistration.cpp compiled where it lies against
    oracle/ref_stubs — see oracle/ref_scanreg_shim.cpp for what is and is not the reference's code).  `available()` is False
    when oracle/_ref/libref_scanreg.so is not built (no /root/reference on this machine)."""
    NAMES = ("full", "sharp", "less_sharp", "flat", "less_flat")
    _L = None

    @classmethod
    def available(cls):
        if cls._L is None:
            path = os.path.join(_HERE, "_ref", "libref_scanreg.so")
            if not os.path.exists(path):
                return False
            L = C.CDLL(path)
            L.ref_sr_create.restype = C.c_void_p
            L.ref_sr_get.restype = C.c_int
            cls._L = L
        return True

    def __init__(self, **cfg):
        assert self.available()
        c = dict(scanPeriod=0.1, imuHistorySize=200, nFeatureRegions=6, curvatureRegion=5, maxCornerSharp=2, maxSurfaceFlat=4,
                 lessFlatFilterSize=0.2, surfaceCurvatureThreshold=0.1)
        c.update(cfg)
        self.h = C.c_void_p(self._L.ref_sr_create(C.c_float(c["scanPeriod"]), c["imuHistorySize"], c["nFeatureRegions"], c["curvatureRegion"],
                                                  c["maxCornerSharp"], c["maxSurfaceFlat"], C.c_float(c["lessFlatFilterSize"]),
                                                  C.c_float(c["surfaceCurvatureThreshold"])))
        if "maxCornerLessSharp" in c:      # parsed on its own by the node (ScanRegistration.cpp:100-109)
            self._L.ref_sr_set_less_sharp(self.h, int(c["maxCornerLessSharp"]))

    def __del__(self):
        if getattr(self, "h", None):
            self._L.ref_sr_destroy(self.h)

    def update_imu(self, stamp, roll, pitch, yaw, acc):
        self._L.ref_sr_update_imu(self.h, C.c_double(stamp), C.c_float(roll), C.c_float(pitch), C.c_float(yaw), C.c_float(acc[0]),
                                  C.c_float(acc[1]), C.c_float(acc[2]))

    def project(self, pts, rel_times):
        """projectPointToStartOfSweep on every row of pts (n,4), in order"""
        out = np.ascontiguousarray(pts, np.float32).copy()
        for i in range(len(out)):
            self._L.ref_sr_project(self.h, out[i].ctypes.data_as(C.c_void_p), C.c_float(rel_times[i]))
        return out

    def process(self, pts, ring_sizes, scan_time=0.0):
        pts = _pts(pts)
        rs = np.ascontiguousarray(ring_sizes, np.int32)
        self._L.ref_sr_process(self.h, C.c_double(scan_time), pts.ctypes.data_as(C.c_void_p), rs.ctypes.data_as(C.c_void_p), len(rs))
        res = {}
        for k, n in enumerate(self.NAMES):
            cnt = self._L.ref_sr_get(self.h, k, None, 0)
            out = np.zeros((max(cnt, 1), 4), np.float32)
            self._L.ref_sr_get(self.h, k, out.ctypes.data_as(C.c_void_p), cnt)
            res[n] = out[:cnt].copy()
        it = np.zeros(12, np.float32)
        self._L.ref_sr_imu_trans(self.h, it.ctypes.data_as(C.c_void_p))
        res["imu_trans"] = it
        return res


class RefLaserOdometry:
    """The REFERENCE's own BasicLaserOdometry (src/lib/BasicLaserOdometry.cpp + the vendored nanoflann compiled where they lie
    against oracle/ref_stubs — see oracle/ref_odometry_shim.cpp for what is and is not the reference's code).  Same surface as
    LaserOdometry above.  `available()` is False when oracle/_ref/libref_odometry.so is not built."""
    _L = None
    _SO = "libref_odometry.so"

    @classmethod
    def available(cls):
        if cls._L is None:
            path = os.path.join(_HERE, "_ref", cls._SO)
            if not os.path.exists(path):
                return False
            cls._L = C.CDLL(path)
            cls._L.ref_odom_create.restype = C.c_void_p
            cls._L.ref_odom_frame_count.restype = C.c_long
        return True

    def __init__(self, scanPeriod=0.1, maxIterations=25, deltaTAbort=0.1, deltaRAbort=0.1):
        assert self.available()
        self.h = C.c_void_p(self._L.ref_odom_create(C.c_float(scanPeriod), maxIterations, C.c_float(deltaTAbort), C.c_float(deltaRAbort)))

    def __del__(self):
        self._L.ref_odom_destroy(self.h)

    def set_features(self, f):
        for k, n in enumerate(ScanRegistration.NAMES):
            p = _pts(f[n])
            self._L.ref_odom_set_cloud(self.h, k, p.ctypes.data_as(C.c_void_p), len(p))

    def set_transform(self, t6):
        t6 = _f32(t6)
        self._L.ref_odom_set_transform(self.h, t6.ctypes.data_as(C.c_void_p))

    def update_imu(self, t12):
        t12 = _f32(t12)
        self._L.ref_odom_update_imu(self.h, t12.ctypes.data_as(C.c_void_p))

    def process(self):
        self._L.ref_odom_process(self.h)

    def _t(self, fn):
        t = np.zeros(6, np.float32)
        fn(self.h, t.ctypes.data_as(C.c_void_p))
        return t

    @property
    def transform(self):
        return self._t(self._L.ref_odom_get_transform)

    @property
    def transform_sum(self):
        return self._t(self._L.ref_odom_get_transform_sum)

    def last_corner(self):
        return _get_cloud(self._L.ref_odom_get_cloud, self.h, 0)

    def last_surf(self):
        return _get_cloud(self._L.ref_odom_get_cloud, self.h, 1)

    def full_to_end(self):
        self._L.ref_odom_transform_full_to_end(self.h)
        return _get_cloud(self._L.ref_odom_get_cloud, self.h, 2)


class RefLaserMapping:
    """The REFERENCE's own BasicLaserMapping (src/lib/BasicLaserMapping.cpp + the vendored nanoflann compiled where they lie
    against oracle/ref_stubs — see oracle/ref_mapping_shim.cpp for what is and is not the reference's code).  Same surface as
    the live-map part of LaserMapping above.  `available()` is False when oracle/_ref/libref_mapping.so is not built."""
    CLOUDS = LaserMapping.CLOUDS
    _L = None
    _SO = "libref_mapping.so"

    @classmethod
    def available(cls):
        if cls._L is None:
            path = os.path.join(_HERE, "_ref", cls._SO)
            if not os.path.exists(path):
                return False
            cls._L = C.CDLL(path)
            cls._L.ref_map_create.restype = C.c_void_p
        return True

    def __init__(self, scanPeriod=0.1, maxIterations=10, deltaTAbort=0.05, deltaRAbort=0.05, cornerLeaf=0.2, surfLeaf=0.4):
        assert self.available()
        self.h = C.c_void_p(self._L.ref_map_create(C.c_float(scanPeriod), maxIterations, C.c_float(deltaTAbort), C.c_float(deltaRAbort),
                                                   C.c_float(cornerLeaf), C.c_float(surfLeaf)))
        self.t = 0.0

    def __del__(self):
        self._L.ref_map_destroy(self.h)

    def set_inputs(self, corner_last, surf_last, full_res, transform_sum):
        for k, p in enumerate((corner_last, surf_last, full_res)):
            p = _pts(p)
            self._L.ref_map_set_cloud(self.h, k, p.ctypes.data_as(C.c_void_p), len(p))
        t = _f32(transform_sum)
        self._L.ref_map_update_odometry(self.h, t.ctypes.data_as(C.c_void_p))

    def update_imu(self, stamp, roll, pitch):
        self._L.ref_map_update_imu(self.h, C.c_double(stamp), C.c_float(roll), C.c_float(pitch))

    def set_time(self, t):
        self.t = float(t)

    def process(self):
        return bool(self._L.ref_map_process(self.h, C.c_double(self.t)))

    def transform(self, which="aft"):
        t = np.zeros(6, np.float32)
        self._L.ref_map_get_transform(self.h, ("aft", "bef", "tobe", "sum").index(which), t.ctypes.data_as(C.c_void_p))
        return t

    def set_transform(self, which, t6):
        t6 = _f32(t6)
        self._L.ref_map_set_transform(self.h, ("aft", "bef", "tobe", "sum").index(which), t6.ctypes.data_as(C.c_void_p))

    def cloud(self, name):
        return _get_cloud(self._L.ref_map_get_cloud, self.h, self.CLOUDS.index(name))

    def has_fresh_map(self):
        return bool(self._L.ref_map_has_fresh_map(self.h))

    def grid_center(self):
        c = np.zeros(3, np.int32)
        self._L.ref_map_grid_center(self.h, c.ctypes.data_as(C.c_void_p))
        return c

    def set_frozen(self, corner, surf):
        self._frozen = (_pts(corner), _pts(surf))

    def register_frozen(self, corner_last, surf_last, guess6):
        """one sweep against the caller's sub-map through the reference's own optimizeTransformTobeMapped (ref_mapping_shim.cpp)"""
        for k, p in enumerate((corner_last, surf_last)):
            p = _pts(p)
            self._L.ref_map_set_cloud(self.h, k, p.ctypes.data_as(C.c_void_p), len(p))
        cm, sm = self._frozen
        g, pose = _f32(guess6), np.zeros(6, np.float32)
        self._L.ref_map_register_frozen(self.h, cm.ctypes.data_as(C.c_void_p), len(cm), sm.ctypes.data_as(C.c_void_p), len(sm),
                                        g.ctypes.data_as(C.c_void_p), pose.ctypes.data_as(C.c_void_p))
        return pose

    def associate(self):
        t = np.zeros(6, np.float32)
        self._L.ref_map_associate(self.h, t.ctypes.data_as(C.c_void_p))
        return t


class RefMultiScanRegistration:
    """The REFERENCE's own MultiScanRegistration (src/lib/MultiScanRegistration.cpp + BasicScanRegistration.cpp compiled where
    they lie against oracle/ref_stubs — see oracle/ref_multiscan_shim.cpp for what is and is not the reference's code): the
    sweep ingestion from raw sensor-axes points to the five feature clouds.  `available()` is False when
    oracle/_ref/libref_multiscan.so is not built."""
    NAMES = ScanRegistration.NAMES
    _L = None

    @classmethod
    def available(cls):
        if cls._L is None:
            path = os.path.join(_HERE, "_ref", "libref_multiscan.so")
            if not os.path.exists(path):
                return False
            cls._L = C.CDLL(path)
            cls._L.ref_ms_create.restype = C.c_void_p
        return True

    @classmethod
    def ring_for_angle(cls, mapper, angle_rad):
        lo, hi, nr = MAPPERS[mapper] if isinstance(mapper, str) else mapper
        return int(cls._L.ref_ms_ring_for_angle(C.c_float(lo), C.c_float(hi), int(nr), C.c_float(angle_rad)))

    def __init__(self, mapper="VLP-16", **cfg):
        assert self.available()
        lo, hi, nr = MAPPERS[mapper] if isinstance(mapper, str) else mapper
        self.n_rings = int(nr)
        c = dict(scanPeriod=0.1, imuHistorySize=200, nFeatureRegions=6, curvatureRegion=5, maxCornerSharp=2, maxSurfaceFlat=4,
                 lessFlatFilterSize=0.2, surfaceCurvatureThreshold=0.1)
        c.update(cfg)
        self.h = C.c_void_p(self._L.ref_ms_create(C.c_float(lo), C.c_float(hi), int(nr), C.c_float(c["scanPeriod"]), c["imuHistorySize"],
                                                  c["nFeatureRegions"], c["curvatureRegion"], c["maxCornerSharp"], c["maxSurfaceFlat"],
                                                  C.c_float(c["lessFlatFilterSize"]), C.c_float(c["surfaceCurvatureThreshold"])))

    def __del__(self):
        if getattr(self, "h", None):
            self._L.ref_ms_destroy(self.h)

    def update_imu(self, stamp, roll, pitch, yaw, acc):
        self._L.ref_ms_update_imu(self.h, C.c_double(stamp), C.c_float(roll), C.c_float(pitch), C.c_float(yaw), C.c_float(acc[0]),
                                  C.c_float(acc[1]), C.c_float(acc[2]))

    def _results(self):
        res = {}
        for k, n in enumerate(self.NAMES):
            cnt = self._L.ref_ms_get(self.h, k, None, 0)
            out = np.zeros((max(cnt, 1), 4), np.float32)
            self._L.ref_ms_get(self.h, k, out.ctypes.data_as(C.c_void_p), cnt)
            res[n] = out[:cnt].copy()
        rs = np.zeros(self.n_rings, np.int32)
        self._L.ref_ms_ring_sizes(self.h, rs.ctypes.data_as(C.c_void_p), self.n_rings)
        res["ring_sizes"] = rs
        it = np.zeros(12, np.float32)
        self._L.ref_ms_imu_trans(self.h, it.ctypes.data_as(C.c_void_p))
        res["imu_trans"] = it
        return res

    def process_raw(self, raw_xyz, scan_time):
        """MultiScanRegistration::process(laserCloudIn, scanTime)"""
        raw = np.ascontiguousarray(raw_xyz, np.float32).reshape(-1, 3)
        self._L.ref_ms_process(self.h, raw.ctypes.data_as(C.c_void_p), len(raw), C.c_double(scan_time))
        return self._results()

    def handle_message(self, raw_xyz, sec, nsec):
        """handleCloudMessage with a (sec, nsec) stamp; None while the start-up delay swallows the message"""
        raw = np.ascontiguousarray(raw_xyz, np.float32).reshape(-1, 3)
        if not self._L.ref_ms_handle_message(self.h, raw.ctypes.data_as(C.c_void_p), len(raw), int(sec), int(nsec)):
            return None
        return self._results()


class RefNodes:
    """The reference's FOUR NODES (MultiScanRegistration -> LaserOdometry -> LaserMapping -> TransformMaintenance: its own node
    classes and Basic* classes compiled where they lie) in one process over the in-process topic bus of oracle/ref_stubs — see
    oracle/ref_nodes_shim.cpp.  `lib` selects the build: None = oracle/_ref/libref_nodes.so (the reference, CPU); a path = the same
    node sources linked against the product's adapter (needs a GPU).  One instance at a time per library (the reference keeps
    process-global kd-trees, BasicLaserMapping.cpp:623-624)."""
    TOPICS = ("/laser_odom_to_init", "/aft_mapped_to_init", "/integrated_to_init")
    _libs = {}

    @classmethod
    def _load(cls, lib):
        path = lib or os.path.join(_HERE, "_ref", "libref_nodes.so")
        if path not in cls._libs:
            if not os.path.exists(path):
                return None
            L = C.CDLL(path)
            L.nodes_create.restype = C.c_void_p
            L.nodes_last_error.restype = C.c_char_p
            cls._libs[path] = L
        return cls._libs[path]

    @classmethod
    def available(cls, lib=None):
        return cls._load(lib) is not None

    def __init__(self, lidar="VLP-16", lib=None, **params):
        self.L = self._load(lib)
        assert self.L is not None
        self.L.nodes_reset_bus()
        for k, v in dict(lidar=lidar, **params).items():
            self.L.nodes_set_param(k.encode(), str(v).encode())
        self.h = C.c_void_p(self.L.nodes_create())
        assert self.L.nodes_ok(self.h), "node set-up failed: " + self.L.nodes_last_error(self.h).decode()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.nodes_destroy(self.h)

    def push_imu(self, sec, nsec, quat_xyzw, acc_xyz):
        q = np.ascontiguousarray(quat_xyzw, np.float64)
        a = np.ascontiguousarray(acc_xyz, np.float64)
        if self.L.nodes_push_imu(self.h, int(sec), int(nsec), q.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p)) != 0:
            raise RuntimeError(self.L.nodes_last_error(self.h).decode())

    def push_cloud(self, raw_xyz, sec, nsec):
        raw = np.ascontiguousarray(raw_xyz, np.float32).reshape(-1, 3)
        if self.L.nodes_push_cloud(self.h, raw.ctypes.data_as(C.c_void_p), len(raw), int(sec), int(nsec)) != 0:
            raise RuntimeError(self.L.nodes_last_error(self.h).decode())

    def odometry(self, topic):
        """every message published on `topic` so far: (stamps (m,), values (m,13) = quaternion xyzw, position, twist angular, linear)"""
        w = self.TOPICS.index(topic)
        m = self.L.nodes_odom_count(self.h, w)
        stamps, vals = np.zeros(m), np.zeros((m, 13), np.float32)
        for i in range(m):
            s = C.c_double()
            self.L.nodes_odom_get(self.h, w, i, C.byref(s), vals[i].ctypes.data_as(C.c_void_p))
            stamps[i] = s.value
        return stamps, vals

    def clouds(self, which):
        """the payloads published on /velodyne_cloud_registered (0) or /laser_cloud_surround (1), each (n,4)"""
        out = []
        for i in range(self.L.nodes_cloud_count(self.h, which)):
            n = self.L.nodes_cloud_get(self.h, which, i, None, 0)
            a = np.zeros(max(n, 1), np.float32)
            self.L.nodes_cloud_get(self.h, which, i, a.ctypes.data_as(C.c_void_p), n)
            out.append(a[:n].reshape(-1, 4).copy())
        return out


class RefLaserOdometryAlt(RefLaserOdometry):
    """the reference's BasicLaserOdometry over the ALTERNATIVE third-party arithmetic (-DREF_STUB_ALT_ARITH): products accumulated in double,
    the 6x6 solve by elimination in double — a sensitivity probe (what the unpinned Eigen operations could change), not a pin"""
    _L = None
    _SO = "libref_odometry_alt.so"


class RefLaserMappingAlt(RefLaserMapping):
    """the same reference translation unit over the ALTERNATIVE third-party arithmetic (-DREF_STUB_ALT_ARITH, ref_stubs/Eigen/Core):
    a sensitivity probe for the operations the stand-in forwards, not a pin"""
    _L = None
    _SO = "libref_mapping_alt.so"

