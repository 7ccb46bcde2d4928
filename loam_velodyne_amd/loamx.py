"""ctypes binding of libloamx.so (the C-ABI of include/loamx.h) for the Python harness (tests/, bench.py, smoke()).

This is plumbing over the C-ABI, not a second implementation: every method is one `loamx_*` call.  There is no CPU
fallback — if the shared library is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOAMX_LIB") or os.path.join(_HERE, "libloamx.so")   # LOAMX_LIB: a diagnostic build (time stamps inside kernels)

OK, SKIPPED, E_INVALID, E_CAPACITY, E_HIP, E_NOGPU, E_UNSUPPORTED = 0, 1, -1, -2, -3, -4, -5


class Cloud(C.Structure):
    _fields_ = [("data", C.c_void_p), ("count", C.c_uint32), ("stride", C.c_uint32),
                ("intensity_offset", C.c_uint32), ("reserved", C.c_uint32)]


class ScanRegConfig(C.Structure):
    _fields_ = [("scan_period", C.c_float), ("n_feature_regions", C.c_int), ("curvature_region", C.c_int),
                ("max_corner_sharp", C.c_int), ("max_surface_flat", C.c_int), ("less_flat_filter_size", C.c_float),
                ("surface_curvature_threshold", C.c_float), ("device", C.c_int), ("max_corner_less_sharp", C.c_int),
                ("imu_history_size", C.c_int)]


class OdomConfig(C.Structure):
    _fields_ = [("scan_period", C.c_float), ("max_iterations", C.c_int), ("delta_t_abort", C.c_float),
                ("delta_r_abort", C.c_float), ("device", C.c_int)]


class MapConfig(C.Structure):
    _fields_ = [("scan_period", C.c_float), ("max_iterations", C.c_int), ("delta_t_abort", C.c_float),
                ("delta_r_abort", C.c_float), ("corner_filter_size", C.c_float), ("surf_filter_size", C.c_float),
                ("map_filter_size", C.c_float), ("device", C.c_int)]


class LoamxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"loamx error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libloamx.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        L.loamx_last_error.restype = C.c_char_p
        if hasattr(L, "loamx_build_info"):
            L.loamx_build_info.restype = C.c_char_p
        if hasattr(L, "loamx_dist_pack_clouds"):   # (64-bit sizes and more than six arguments: spelled out rather than left to ctypes' defaults)
            L.loamx_dist_pack_clouds.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
            L.loamx_dist_unpack_clouds_header.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
            L.loamx_dist_unpack_clouds_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
            L.loamx_dist_gatherv.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        for n in ("loamx_scanreg_create", "loamx_odom_create", "loamx_map_create", "loamx_batch_create",
                  "loamx_batch_stream"):
            if hasattr(L, n):
                getattr(L, n).restype = C.c_void_p
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise LoamxError(rc, lib().loamx_last_error().decode())
    return rc


def device_count() -> int:
    return int(lib().loamx_device_count())


def pinned_empty(shape, dtype=np.float32):
    """numpy array in host memory pinned by the library's HIP runtime (loamx_host_alloc): the entry points copy straight from / to such
    arrays (DMA) instead of through their staging blocks.  Freed when the array and every view of it are gone."""
    import weakref
    L = lib()
    L.loamx_host_alloc.restype = C.c_void_p
    L.loamx_host_alloc.argtypes = [C.c_size_t]
    L.loamx_host_free.argtypes = [C.c_void_p]
    shape = tuple(int(x) for x in (shape if hasattr(shape, "__len__") else (shape,)))
    nbytes = max(int(np.prod(shape)) * np.dtype(dtype).itemsize, 1)
    ptr = L.loamx_host_alloc(nbytes)
    if not ptr:
        raise LoamxError(E_HIP, "loamx_host_alloc failed")
    buf = (C.c_char * nbytes).from_address(ptr)
    weakref.finalize(buf, L.loamx_host_free, C.c_void_p(ptr))
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def pinned_copy(a, dtype=np.float32):
    out = pinned_empty(np.shape(a), dtype)
    out[...] = a
    return out


def build_info() -> dict:
    """loamx_build_info() as a dict: abi, diag (1: a diagnostic build that reads the result-changing LOAMX_* switches), rccl, roctx."""
    return dict(kv.split("=", 1) for kv in lib().loamx_build_info().decode().split(";") if kv)


def as_points(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a.reshape(-1, 4)
    assert a.ndim == 2 and a.shape[1] in (4, 8), "points must be (N,4) packed xyzi or (N,8) PCL PointXYZI records"
    return a


def cloud_of(a: np.ndarray) -> Cloud:
    """Describe a (N,4) packed or (N,8) PCL-layout float32 array."""
    stride = a.shape[1] * 4
    return Cloud(a.ctypes.data if a.size else None, a.shape[0], stride, 12 if stride == 16 else 16, 0)


def to_pcl_layout(a: np.ndarray) -> np.ndarray:
    """(N,4) xyzi -> (N,8) pcl::PointXYZI records {x,y,z,1, intensity,0,0,0}."""
    out = np.zeros((len(a), 8), np.float32)
    out[:, :3] = a[:, :3]
    out[:, 3] = 1.0
    out[:, 4] = a[:, 3]
    return out


def _cfg(struct_t, default_fn, **kw):
    c = struct_t()
    getattr(lib(), default_fn)(C.byref(c))
    for k, v in kw.items():
        assert hasattr(c, k), k
        setattr(c, k, v)
    return c


class Batch:
    """loamx_batch_*: B independent sweeps registered against one frozen sub-map."""

    def __init__(self, max_sweeps: int, **cfg):
        self._c = _cfg(MapConfig, "loamx_map_default_config", **cfg)
        self.h = C.c_void_p(lib().loamx_batch_create(C.byref(self._c), max_sweeps))
        if not self.h:
            raise LoamxError(E_INVALID, lib().loamx_last_error().decode())
        self.n = 0
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            lib().loamx_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def set_frozen(self, corner_map, surf_map):
        c, s = as_points(corner_map), as_points(surf_map)
        cc, sc = cloud_of(c), cloud_of(s)
        _check(lib().loamx_batch_set_frozen(self.h, C.byref(cc), C.byref(sc)))

    def set_frozen_device(self, d_corner_ptr: int, n_corner: int, d_surf_ptr: int, n_surf: int):
        _check(lib().loamx_batch_set_frozen_device(self.h, C.c_void_p(d_corner_ptr), n_corner, C.c_void_p(d_surf_ptr), n_surf))

    def stage_frozen_device(self, d_corner_ptr: int, n_corner: int, d_surf_ptr: int, n_surf: int, wait_event: int = 0):
        """index the NEXT epoch's sub-map in the background (double-buffered map epochs); wait_event = raw hipEvent_t"""
        _check(lib().loamx_batch_stage_frozen_device(self.h, C.c_void_p(d_corner_ptr), n_corner, C.c_void_p(d_surf_ptr), n_surf,
                                                     C.c_void_p(wait_event or None)))

    def stage_frozen(self, corner_map, surf_map):
        cm, sm = as_points(corner_map), as_points(surf_map)
        cc, sc = cloud_of(cm), cloud_of(sm)
        _check(lib().loamx_batch_stage_frozen(self.h, C.byref(cc), C.byref(sc)))

    def swap_frozen(self) -> bool:
        return _check(lib().loamx_batch_swap_frozen(self.h)) == OK

    def upload(self, corner_last, surf_last, guesses, full_res=None):
        n = len(corner_last)
        assert len(surf_last) == n and len(guesses) == n
        cl = [as_points(a) for a in corner_last]
        sl = [as_points(a) for a in surf_last]
        fr = [as_points(a) for a in full_res] if full_res is not None else None
        CA = (Cloud * n)(*[cloud_of(a) for a in cl])
        SA = (Cloud * n)(*[cloud_of(a) for a in sl])
        FA = (Cloud * n)(*[cloud_of(a) for a in fr]) if fr is not None else None
        g = np.ascontiguousarray(guesses, np.float32).reshape(n, 6)
        _check(lib().loamx_batch_upload(self.h, n, CA, SA, FA, g.ctypes.data_as(C.c_void_p)))
        self.n = n
        self._full_sizes = [len(a) for a in fr] if fr is not None else None

    def run(self):
        return _check(lib().loamx_batch_run(self.h))

    def run_async(self):
        _check(lib().loamx_batch_run_async(self.h))

    def sync(self):
        _check(lib().loamx_batch_sync(self.h))

    def download(self):
        poses = np.zeros((self.n, 6), np.float32)
        stats = np.zeros((self.n, 4), np.int32)
        _check(lib().loamx_batch_download(self.h, poses.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p)))
        return poses, stats

    def download_full_res(self, sweep: int):
        out = np.zeros((self._full_sizes[sweep], 4), np.float32)
        c = cloud_of(out)
        _check(lib().loamx_batch_download_full_res(self.h, sweep, C.byref(c)))
        return out[:c.count]

    def download_ds(self, sweep: int, cap: int = 1 << 17):
        """the down-sampled stack clouds (corner, surf) of one sweep of the last run — the Gauss-Newton query points"""
        co, so = np.zeros((cap, 4), np.float32), np.zeros((cap, 4), np.float32)
        cc, sc = cloud_of(co), cloud_of(so)
        _check(lib().loamx_batch_download_ds(self.h, sweep, C.byref(cc), C.byref(sc)))
        return co[:cc.count].copy(), so[:sc.count].copy()

    def qr6_probe(self, ata, atb):
        """parity hook: n 6x6 systems through the kernels' wave-cooperative solve and through the scalar routine -> (x_coop, x_scalar)"""
        A = np.ascontiguousarray(ata, np.float32).reshape(-1, 36)
        b = np.ascontiguousarray(atb, np.float32).reshape(-1, 6)
        assert len(A) == len(b)
        xc, xs = np.zeros_like(b), np.zeros_like(b)
        _check(lib().loamx_batch_qr6_probe(self.h, A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), len(A),
                                           xc.ctypes.data_as(C.c_void_p), xs.ctypes.data_as(C.c_void_p)))
        return xc, xs

    def xrec_stress(self, pairs: int = 64, rounds: int = 20000):
        """stress probe of k_odom_lm's tagged-record exchange -> dict(accepted, torn, inconsistent, timed_out)"""
        out = (C.c_uint64 * 4)()
        _check(lib().loamx_batch_xrec_stress(self.h, pairs, rounds, out))
        return dict(accepted=int(out[0]), torn=int(out[1]), inconsistent=int(out[2]), timed_out=int(out[3]))

    def knn_probe(self, which: int, queries_xyz):
        """the library's own 5-NN search for map-frame points: (indices into the cloud given to set_frozen, squared distances)"""
        q = np.ascontiguousarray(np.asarray(queries_xyz, np.float32)[:, :3])
        idx = np.zeros((len(q), 5), np.uint32)
        d2 = np.zeros((len(q), 5), np.float32)
        _check(lib().loamx_batch_knn_probe(self.h, which, q.ctypes.data_as(C.c_void_p), len(q), idx.ctypes.data_as(C.c_void_p),
                                           d2.ctypes.data_as(C.c_void_p)))
        return idx, d2

    def set_timing(self, on: bool):
        _check(lib().loamx_batch_set_timing(self.h, 1 if on else 0))

    def timing(self):
        ms = (C.c_float * 4)()
        cnt = (C.c_uint64 * 4)()
        _check(lib().loamx_batch_get_timing(self.h, ms, cnt))
        return dict(run_ms=ms[0], residual_ms=ms[1], residual_launches=int(cnt[0]), query_iterations=int(cnt[1]),
                    queries=int(cnt[2]))

    @property
    def stream(self) -> int:
        return int(lib().loamx_batch_stream(self.h) or 0)


class TransformMaintenance:
    """loamx_tm_*: BasicTransformMaintenance (host arithmetic, no device)."""

    def __init__(self):
        lib().loamx_tm_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().loamx_tm_create())

    def close(self):
        if getattr(self, "h", None):
            lib().loamx_tm_destroy(self.h)
            self.h = None

    __del__ = close

    def update_odometry(self, transform_sum):
        t = np.ascontiguousarray(transform_sum, np.float32)
        _check(lib().loamx_tm_update_odometry(self.h, t.ctypes.data_as(C.c_void_p)))

    def update_mapping_transform(self, aft_mapped, bef_mapped):
        a, b = np.ascontiguousarray(aft_mapped, np.float32), np.ascontiguousarray(bef_mapped, np.float32)
        _check(lib().loamx_tm_update_mapping_transform(self.h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))

    def associate_to_map(self):
        _check(lib().loamx_tm_associate_to_map(self.h))
        out = np.zeros(6, np.float32)
        _check(lib().loamx_tm_get_mapped(self.h, out.ctypes.data_as(C.c_void_p)))
        return out


def wire_pose_to_quat(rot_xyz):
    r = np.ascontiguousarray(rot_xyz, np.float32)
    q = np.zeros(4, np.float64)
    _check(lib().loamx_wire_pose_to_quat(r.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p)))
    return q


def wire_quat_to_pose(quat_xyzw):
    q = np.ascontiguousarray(quat_xyzw, np.float64)
    r = np.zeros(3, np.float32)
    _check(lib().loamx_wire_quat_to_pose(q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p)))
    return r


class MultiScanMapper(C.Structure):
    """loamx_multiscan_mapper (loam::MultiScanMapper)."""
    _fields_ = [("lower_bound_deg", C.c_float), ("upper_bound_deg", C.c_float), ("n_scan_rings", C.c_uint32)]


class ScanRegistration:
    """loamx_scanreg_*: BasicScanRegistration::processScanlines / extractFeatures on the GPU."""
    NAMES = ("sharp", "less_sharp", "flat", "less_flat")

    def __init__(self, **cfg):
        self._c = _cfg(ScanRegConfig, "loamx_scanreg_default_config", **cfg)
        self.h = C.c_void_p(lib().loamx_scanreg_create(C.byref(self._c)))
        if not self.h:
            raise LoamxError(E_INVALID, lib().loamx_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            lib().loamx_scanreg_destroy(self.h)
            self.h = None

    __del__ = close

    def configure(self, **cfg):
        """loamx_scanreg_configure: new parameters, the handle's IMU history and sweep state are kept"""
        for k, v in cfg.items():
            assert hasattr(self._c, k), k
            setattr(self._c, k, v)
        _check(lib().loamx_scanreg_configure(self.h, C.byref(self._c)))

    def process(self, points, ring_sizes, pcl_layout=False):
        pts = as_points(points)
        rs = np.ascontiguousarray(ring_sizes, np.uint32)
        n = len(pts)
        width = 8 if pcl_layout else 4
        key = (max(n, 1), width)
        if getattr(self, "_out_key", None) != key:   # landing buffers are kept between calls (the results below are copies)
            self._outs = [np.zeros(key, np.float32) for _ in range(4)]
            self._out_key = key
        outs = self._outs
        cl = [cloud_of(o) for o in outs]
        cin = cloud_of(pts)
        _check(lib().loamx_scanreg_process(self.h, C.byref(cin), rs.ctypes.data_as(C.c_void_p), len(rs), C.byref(cl[0]),
                                           C.byref(cl[1]), C.byref(cl[2]), C.byref(cl[3])))
        res = {name: outs[k][:cl[k].count].copy() for k, name in enumerate(self.NAMES)}
        res["full"] = pts   # laserCloud(): THE CALLER'S ARRAY, not a copy (binned rings already): a consumer that registers it in place
        # (LaserMapping.process(inplace=True)) rewrites the caller's sweep — pass a copy where the sweep is needed again
        return res

    def process_linked(self, points, ring_sizes):
        """loamx_scanreg_process_linked: enqueue the extraction and leave the clouds in HBM for LaserOdometry.process_linked"""
        pts = as_points(points)
        rs = np.ascontiguousarray(ring_sizes, np.uint32)
        cin = cloud_of(pts)
        self._linked_n = len(pts)
        return _check(lib().loamx_scanreg_process_linked(self.h, C.byref(cin), rs.ctypes.data_as(C.c_void_p), len(rs)))

    def update_imu(self, stamp, roll, pitch, yaw, acc):
        """loamx_scanreg_update_imu (updateIMUData); stamp in seconds"""
        a = np.ascontiguousarray(acc, np.float32)
        _check(lib().loamx_scanreg_update_imu(self.h, C.c_double(stamp), C.c_float(roll), C.c_float(pitch), C.c_float(yaw),
                                              a.ctypes.data_as(C.c_void_p)))

    def set_time(self, t):
        """the scanTime of the next process call"""
        _check(lib().loamx_scanreg_set_time(self.h, C.c_double(t)))

    def imu_trans(self):
        out = np.zeros(12, np.float32)
        _check(lib().loamx_scanreg_get_imu_trans(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def process_raw(self, raw_xyz, sensor="VLP-16", mapper=None):
        """loamx_scanreg_process_raw: MultiScanRegistration::process on a raw (n,3) firing-order cloud in sensor axes.
        mapper = (lower_deg, upper_deg, n_rings) overrides the sensor preset.  Adds "full" and "ring_sizes"."""
        raw = np.ascontiguousarray(raw_xyz, np.float32).reshape(-1, 3)
        m = MultiScanMapper()
        if mapper is None:
            _check(lib().loamx_multiscan_mapper_preset(sensor.encode(), C.byref(m)))
        else:
            m.lower_bound_deg, m.upper_bound_deg, m.n_scan_rings = float(mapper[0]), float(mapper[1]), int(mapper[2])
        n = len(raw)
        outs = [np.zeros((max(n, 1), 4), np.float32) for _ in range(5)]
        cl = [cloud_of(o) for o in outs]
        rs = np.zeros(max(int(m.n_scan_rings), 1), np.uint32)
        _check(lib().loamx_scanreg_process_raw(self.h, C.byref(m), raw.ctypes.data_as(C.c_void_p), n, 12, C.byref(cl[4]),
                                               rs.ctypes.data_as(C.c_void_p), C.byref(cl[0]), C.byref(cl[1]), C.byref(cl[2]),
                                               C.byref(cl[3])))
        res = {name: outs[k][:cl[k].count].copy() for k, name in enumerate(self.NAMES)}
        res["full"] = outs[4][:cl[4].count].copy()
        res["ring_sizes"] = rs.astype(np.int32)
        return res


class LaserOdometry:
    """loamx_odom_*: BasicLaserOdometry on the GPU."""

    def __init__(self, **cfg):
        self._c = _cfg(OdomConfig, "loamx_odom_default_config", **cfg)
        self.h = C.c_void_p(lib().loamx_odom_create(C.byref(self._c)))
        if not self.h:
            raise LoamxError(E_INVALID, lib().loamx_last_error().decode())
        self._sizes = (0, 0)

    def close(self):
        if getattr(self, "h", None):
            lib().loamx_odom_destroy(self.h)
            self.h = None

    __del__ = close

    def update_imu(self, t12):
        t = np.ascontiguousarray(t12, np.float32)
        _check(lib().loamx_odom_update_imu(self.h, t.ctypes.data_as(C.c_void_p)))

    def process(self, feats):
        arrs = [as_points(feats[n]) for n in ("sharp", "less_sharp", "flat", "less_flat")]
        cl = [cloud_of(a) for a in arrs]
        rc = _check(lib().loamx_odom_process(self.h, C.byref(cl[0]), C.byref(cl[1]), C.byref(cl[2]), C.byref(cl[3])))
        self._sizes = (len(arrs[1]), len(arrs[3]))
        return rc

    def process_linked(self, scanreg):
        """loamx_odom_process_linked: the sweep `scanreg` has just extracted, taken from its device buffers"""
        rc = _check(lib().loamx_odom_process_linked(self.h, scanreg.h))
        self._sizes = (scanreg._linked_n, scanreg._linked_n)   # (capacities for last_clouds(): each cloud is a subset of the sweep)
        return rc

    def link_wait(self):
        return _check(lib().loamx_odom_link_wait(self.h))

    def _t(self, fn):
        t = np.zeros(6, np.float32)
        _check(fn(self.h, t.ctypes.data_as(C.c_void_p)))
        return t

    @property
    def transform(self):
        return self._t(lib().loamx_odom_get_transform)

    @property
    def transform_sum(self):
        return self._t(lib().loamx_odom_get_transform_sum)

    def set_transform(self, t6):
        t = np.ascontiguousarray(t6, np.float32)
        _check(lib().loamx_odom_set_transform(self.h, t.ctypes.data_as(C.c_void_p)))

    def set_transform_sum(self, t6):
        t = np.ascontiguousarray(t6, np.float32)
        _check(lib().loamx_odom_set_transform_sum(self.h, t.ctypes.data_as(C.c_void_p)))

    def last_clouds(self):
        c = np.zeros((max(self._sizes[0], 1), 4), np.float32)
        s = np.zeros((max(self._sizes[1], 1), 4), np.float32)
        cc, sc = cloud_of(c), cloud_of(s)
        _check(lib().loamx_odom_get_last_clouds(self.h, C.byref(cc), C.byref(sc)))
        return c[:cc.count].copy(), s[:sc.count].copy()

    def transform_to_end(self, cloud):
        a = as_points(cloud).copy()
        c = cloud_of(a)
        _check(lib().loamx_odom_transform_to_end(self.h, C.byref(c)))
        return a

    def stats(self):
        s = (C.c_int * 4)()
        _check(lib().loamx_odom_get_stats(self.h, s))
        return dict(iterations=s[0], sel=s[1], frame=s[2], degenerate=s[3])


class LaserMapping:
    """loamx_map_*: BasicLaserMapping (live rolling map) on the GPU."""

    def __init__(self, **cfg):
        self._c = _cfg(MapConfig, "loamx_map_default_config", **cfg)
        self.h = C.c_void_p(lib().loamx_map_create(C.byref(self._c)))
        if not self.h:
            raise LoamxError(E_INVALID, lib().loamx_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            lib().loamx_map_destroy(self.h)
            self.h = None

    __del__ = close

    def update_odometry(self, t6):
        t = np.ascontiguousarray(t6, np.float32)
        _check(lib().loamx_map_update_odometry(self.h, t.ctypes.data_as(C.c_void_p)))

    def update_imu(self, stamp, roll, pitch):
        """updateIMU(IMUState2); stamp in seconds"""
        _check(lib().loamx_map_update_imu(self.h, C.c_double(stamp), C.c_float(roll), C.c_float(pitch)))

    def set_time(self, t):
        """the laserOdometryTime argument of process()"""
        _check(lib().loamx_map_set_time(self.h, C.c_double(t)))

    def process(self, corner_last, surf_last, full_res=None, inplace=False):
        """inplace: register full_res where it lies (what the C entry point does) instead of in a copy"""
        c, s = as_points(corner_last), as_points(surf_last)
        cc, sc = cloud_of(c), cloud_of(s)
        if full_res is not None:
            f = as_points(full_res) if inplace else as_points(full_res).copy()
            fc = cloud_of(f)
            rc = _check(lib().loamx_map_process(self.h, C.byref(cc), C.byref(sc), C.byref(fc)))
            return rc, f
        rc = _check(lib().loamx_map_process(self.h, C.byref(cc), C.byref(sc), None))
        return rc, None

    def process_linked(self, odometry, full_out=None):
        """loamx_map_process_linked: the sweep `odometry` has just handed on (device to device); full_out: an (n, 4) / (n, 8) float32
        array with room for the sweep — receives the registered full-resolution cloud (returned trimmed, a view)"""
        if full_out is None:
            return _check(lib().loamx_map_process_linked(self.h, odometry.h, None)), None
        fc = cloud_of(full_out)
        rc = _check(lib().loamx_map_process_linked(self.h, odometry.h, C.byref(fc)))
        return rc, full_out[:fc.count]

    def speculation(self):
        """(sweeps that adopted the partition prepared for the predicted pose, sweeps whose prediction missed)"""
        c = (C.c_uint64 * 2)()
        _check(lib().loamx_map_get_speculation(self.h, c))
        return int(c[0]), int(c[1])

    def insert(self, corner_last, surf_last, pose6):
        """loamx_map_insert: the epoch merge step — stack, down-size and insert a sweep registered elsewhere with the given pose"""
        c, s = as_points(corner_last), as_points(surf_last)
        cc, sc = cloud_of(c), cloud_of(s)
        p = np.ascontiguousarray(pose6, np.float32)
        assert p.shape == (6,)
        return _check(lib().loamx_map_insert(self.h, C.byref(cc), C.byref(sc), p.ctypes.data_as(C.c_void_p)))

    def transform(self, which="aft"):
        t = np.zeros(6, np.float32)
        _check(lib().loamx_map_get_transform(self.h, ("aft", "bef", "tobe", "sum").index(which), t.ctypes.data_as(C.c_void_p)))
        return t

    def set_transform(self, which, t6):
        t = np.ascontiguousarray(t6, np.float32)
        _check(lib().loamx_map_set_transform(self.h, ("aft", "bef", "tobe", "sum").index(which), t.ctypes.data_as(C.c_void_p)))

    def set_timing(self, on: bool):
        _check(lib().loamx_map_set_timing(self.h, 1 if on else 0))

    def timing(self):
        ms = (C.c_float * 4)()
        cnt = (C.c_uint64 * 4)()
        _check(lib().loamx_map_get_timing(self.h, ms, cnt))
        return dict(run_ms=ms[0], residual_ms=ms[1], residual_launches=int(cnt[0]), query_iterations=int(cnt[1]), queries=int(cnt[2]))

    def has_fresh_map(self):
        return bool(lib().loamx_map_has_fresh_map(self.h))

    def _get(self, fn, *args, cap=1 << 16):
        while True:
            out = np.zeros((cap, 4), np.float32)
            c = cloud_of(out)
            rc = fn(self.h, *args, C.byref(c))
            if rc == E_CAPACITY:
                cap = int(c.count) + 16
                continue
            _check(rc)
            return out[:c.count].copy()

    def save_snapshot(self, path: str):
        _check(lib().loamx_map_save_snapshot(self.h, path.encode()))

    def load_snapshot(self, path: str):
        _check(lib().loamx_map_load_snapshot(self.h, path.encode()))

    def surround(self):
        return self._get(lib().loamx_map_get_surround)

    def cubes(self, which):
        return self._get(lib().loamx_map_get_cubes, 0 if which == "corner" else 1)

    def load_cubes(self, corner, surf):
        c, s = as_points(corner), as_points(surf)
        cc, sc = cloud_of(c), cloud_of(s)
        _check(lib().loamx_map_load_cubes(self.h, C.byref(cc), C.byref(sc)))

    def stats(self):
        s = (C.c_int * 8)()
        _check(lib().loamx_map_get_stats(self.h, s))
        keys = ("iterations", "sel", "corner_ds", "surf_ds", "corner_from_map", "surf_from_map", "degenerate", "optimized")
        return dict(zip(keys, (int(v) for v in s)))


class Pipeline:
    """loamx_pipeline_*: n independent streams, one sweep per stream per step, features -> odometry -> registration
    against a frozen sub-map."""

    def __init__(self, n_streams: int, scanreg=None, odom=None, mapping=None):
        self._f = _cfg(ScanRegConfig, "loamx_scanreg_default_config", **(scanreg or {}))
        self._o = _cfg(OdomConfig, "loamx_odom_default_config", **(odom or {}))
        self._m = _cfg(MapConfig, "loamx_map_default_config", **(mapping or {}))
        L = lib()
        L.loamx_pipeline_create.restype = C.c_void_p
        L.loamx_pipeline_stream.restype = C.c_void_p
        self.h = C.c_void_p(L.loamx_pipeline_create(C.byref(self._f), C.byref(self._o), C.byref(self._m), n_streams))
        if not self.h:
            raise LoamxError(E_INVALID, L.loamx_last_error().decode())
        self.n_streams = n_streams
        self._sizes = None

    def close(self):
        if getattr(self, "h", None):
            lib().loamx_pipeline_destroy(self.h)
            self.h = None

    __del__ = close

    def set_frozen(self, corner_map, surf_map):
        c, s = as_points(corner_map), as_points(surf_map)
        cc, sc = cloud_of(c), cloud_of(s)
        _check(lib().loamx_pipeline_set_frozen(self.h, C.byref(cc), C.byref(sc)))

    def set_frozen_device(self, d_corner_ptr, n_corner, d_surf_ptr, n_surf):
        _check(lib().loamx_pipeline_set_frozen_device(self.h, C.c_void_p(d_corner_ptr), n_corner, C.c_void_p(d_surf_ptr), n_surf))

    def stage_frozen_device(self, d_corner_ptr, n_corner, d_surf_ptr, n_surf, wait_event: int = 0):
        """index the NEXT epoch's sub-map in the background (double-buffered map epochs); wait_event = raw hipEvent_t"""
        _check(lib().loamx_pipeline_stage_frozen_device(self.h, C.c_void_p(d_corner_ptr), n_corner, C.c_void_p(d_surf_ptr), n_surf,
                                                        C.c_void_p(wait_event or None)))

    def stage_frozen(self, corner_map, surf_map):
        cm, sm = as_points(corner_map), as_points(surf_map)
        cc, sc = cloud_of(cm), cloud_of(sm)
        _check(lib().loamx_pipeline_stage_frozen(self.h, C.byref(cc), C.byref(sc)))

    def swap_frozen(self) -> bool:
        return _check(lib().loamx_pipeline_swap_frozen(self.h)) == OK

    def set_state(self, stream, transform=None, transform_sum=None, bef=None, aft=None):
        def p(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, np.float32)
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        keep = []
        _check(lib().loamx_pipeline_set_state(self.h, stream, p(transform), p(transform_sum), p(bef), p(aft)))

    def upload(self, sweeps):
        """sweeps[t][s] = (points (N,4), ring_sizes)"""
        n_steps = len(sweeps)
        ns = self.n_streams
        pts, rings = [], []
        for t in range(n_steps):
            assert len(sweeps[t]) == ns
            for s in range(ns):
                pts.append(as_points(sweeps[t][s][0]))
                rings.append(np.ascontiguousarray(sweeps[t][s][1], np.uint32))
        CA = (Cloud * len(pts))(*[cloud_of(a) for a in pts])
        RP = (C.c_void_p * len(pts))(*[r.ctypes.data for r in rings])
        NR = (C.c_uint32 * len(pts))(*[len(r) for r in rings])
        _check(lib().loamx_pipeline_upload(self.h, n_steps, CA, RP, NR))
        self._sizes = [len(a) for a in pts]

    def stage_step(self, t: int, sweeps_t):
        """streaming input: stage step t (sweeps_t[s] = (points (N,4) float32 C-contiguous, ring_sizes)) without blocking.
        The arrays must stay alive until step(t) has returned (kept referenced here); pinned memory makes the copy a DMA."""
        ns = self.n_streams
        assert len(sweeps_t) == ns
        pts = [as_points(sweeps_t[s][0]) for s in range(ns)]
        rings = [np.ascontiguousarray(sweeps_t[s][1], np.uint32) for s in range(ns)]
        CA = (Cloud * ns)(*[cloud_of(a) for a in pts])
        RP = (C.c_void_p * ns)(*[r.ctypes.data for r in rings])
        NR = (C.c_uint32 * ns)(*[len(r) for r in rings])
        _check(lib().loamx_pipeline_stage_step(self.h, t, CA, RP, NR))
        if not hasattr(self, "_staged"):
            self._staged = {}
        self._staged[t % 8] = (pts, rings, CA, RP, NR)

    def stage_step_raw(self, t: int, raws, sensor="VLP-16", mapper=None, scan_times=None):
        """streaming raw input: raws[s] = (N,3) float32 C-contiguous sensor-frame points in firing order (kept alive here)"""
        ns = self.n_streams
        assert len(raws) == ns
        arrs = [np.ascontiguousarray(r, np.float32).reshape(-1, 3) for r in raws]
        m = MultiScanMapper()
        if mapper is None:
            _check(lib().loamx_multiscan_mapper_preset(sensor.encode(), C.byref(m)))
        else:
            m.lower_bound_deg, m.upper_bound_deg, m.n_scan_rings = float(mapper[0]), float(mapper[1]), int(mapper[2])
        PP = (C.c_void_p * ns)(*[a.ctypes.data if len(a) else None for a in arrs])
        CN = (C.c_uint32 * ns)(*[len(a) for a in arrs])
        st = (C.c_double * ns)(*[float(x) for x in scan_times]) if scan_times is not None else None
        _check(lib().loamx_pipeline_stage_step_raw(self.h, t, PP, CN, 12, C.byref(m), st))
        if not hasattr(self, "_staged"):
            self._staged = {}
        self._staged[t % 8] = (arrs, PP, CN, st)

    def update_imu(self, stream: int, stamp, roll, pitch, yaw, acc):
        a = np.ascontiguousarray(acc, np.float32)
        _check(lib().loamx_pipeline_update_imu(self.h, stream, C.c_double(stamp), C.c_float(roll), C.c_float(pitch), C.c_float(yaw),
                                               a.ctypes.data_as(C.c_void_p)))

    def enable_async_downloads(self):
        _check(lib().loamx_pipeline_enable_async_downloads(self.h))

    def download_step_async(self, outs):
        """outs: list of (N,4) float32 C-contiguous arrays (capacity); returns the point counts (the data is valid after wait_downloads())"""
        CA = (Cloud * len(outs))(*[cloud_of(a) for a in outs])
        _check(lib().loamx_pipeline_download_step_async(self.h, CA, len(outs)))
        self._dl_keep = (outs, CA)
        return [int(CA[k].count) for k in range(len(outs))]

    def wait_downloads(self):
        _check(lib().loamx_pipeline_wait_downloads(self.h))

    def download_counts(self):
        """(downloads handed to the SDMA engine directly, downloads through hipMemcpyAsync)"""
        c = (C.c_uint64 * 2)()
        _check(lib().loamx_pipeline_download_counts(self.h, c))
        return int(c[0]), int(c[1])

    def step(self, t: int):
        return _check(lib().loamx_pipeline_step(self.h, t))

    def get(self, stream: int):
        tr, ts, aft = (np.zeros(6, np.float32) for _ in range(3))
        st = (C.c_int * 8)()
        _check(lib().loamx_pipeline_get(self.h, stream, tr.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p),
                                        aft.ctypes.data_as(C.c_void_p), st))
        keys = ("odom_iterations", "odom_sel", "map_iterations", "map_sel", "corner_ds", "surf_ds", "degenerate", "mapped")
        return tr, ts, aft, dict(zip(keys, (int(v) for v in st)))

    def download_full_res(self, slot: int, n: int):
        out = np.zeros((n, 4), np.float32)
        c = cloud_of(out)
        _check(lib().loamx_pipeline_download_full_res(self.h, slot, C.byref(c)))
        return out[:c.count]

    def last_clouds(self, stream: int, capacity: int):
        """loamx_pipeline_download_last_clouds: (corner_last, surf_last) of the stream's sweep of the last step"""
        c, s = np.zeros((max(capacity, 1), 4), np.float32), np.zeros((max(capacity, 1), 4), np.float32)
        cc, sc = cloud_of(c), cloud_of(s)
        _check(lib().loamx_pipeline_download_last_clouds(self.h, stream, C.byref(cc), C.byref(sc)))
        return c[:cc.count].copy(), s[:sc.count].copy()

    def set_lookahead(self, on: bool):
        _check(lib().loamx_pipeline_set_lookahead(self.h, 1 if on else 0))

    def lookahead_depth(self) -> int:
        """steps the odometry may run ahead of the registration (0: look-ahead off)"""
        return int(lib().loamx_pipeline_lookahead_depth(self.h))

    def drain_lookahead(self):
        """wait until the look-ahead has run as far as it may; returns the last step whose odometry is complete (-1: none)"""
        last = C.c_int(-1)
        _check(lib().loamx_pipeline_drain_lookahead(self.h, C.byref(last)))
        return int(last.value)

    def set_timing(self, on, per_launch: bool = True):
        """on: stage events; per_launch: also an event pair around every Gauss-Newton launch (costs ~3 % of a step)."""
        _check(lib().loamx_pipeline_set_timing(self.h, (1 if per_launch else 2) if on else 0))

    def timing(self):
        ms = (C.c_float * 4)()
        rms = (C.c_float * 4)()
        cnt = (C.c_uint64 * 4)()
        _check(lib().loamx_pipeline_get_timing(self.h, ms, rms, cnt))
        return dict(features_ms=ms[0], odometry_ms=ms[1], registration_ms=ms[2], step_ms=ms[3], reg_run_ms=rms[0],
                    residual_ms=rms[1], residual_launches=int(cnt[0]), query_iterations=int(cnt[1]), queries=int(cnt[2]))

    def odom_launch_timing(self):
        """running totals of the odometry chains' timed launch pairs (loamx_pipeline_get_odom_launch_timing)"""
        ms = (C.c_double * 4)()
        cnt = (C.c_uint64 * 7)()
        _check(lib().loamx_pipeline_get_odom_launch_timing(self.h, ms, cnt))
        return dict(lm_ms=ms[0], lm_noop_ms=ms[1], corr_ms=ms[2], corr_noop_ms=ms[3], lm_launches=int(cnt[0]), lm_noop_launches=int(cnt[1]),
                    lm_iterations=int(cnt[2]), corr_launches=int(cnt[3]), corr_noop_launches=int(cnt[4]), lm_bytes=int(cnt[5]),
                    corr_features=int(cnt[6]))

    @property
    def stream(self) -> int:
        return int(lib().loamx_pipeline_stream(self.h) or 0)


def dist_shard_of(rank: int, world: int, batch: int):
    b, e = C.c_uint32(0), C.c_uint32(0)
    _check(lib().loamx_dist_shard_of(rank, world, batch, C.byref(b), C.byref(e)))
    return int(b.value), int(e.value)


def dist_pack_results(poses6, iters_flags, n_pad: int) -> np.ndarray:
    """a rank's padded send block of the result exchange (n_pad x 8 floats)"""
    p = np.ascontiguousarray(poses6, np.float32).reshape(-1, 6)
    f = np.ascontiguousarray(iters_flags, np.int32).reshape(len(p), 2) if iters_flags is not None else None
    out = np.zeros((n_pad, 8), np.float32)
    _check(lib().loamx_dist_pack_results(p.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p) if f is not None else None, len(p), n_pad,
                                         out.ctypes.data_as(C.c_void_p)))
    return out


def dist_unpack_results(recv, counts, n_pad: int):
    """inverse of dist_pack_results for the gathered blocks recv[world][n_pad][8]"""
    r = np.ascontiguousarray(recv, np.float32)
    c = np.ascontiguousarray(counts, np.uint32)
    tot = int(c.sum())
    pa, fa = np.zeros((max(tot, 1), 6), np.float32), np.zeros((max(tot, 1), 2), np.int32)
    _check(lib().loamx_dist_unpack_results(r.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), len(c), n_pad, pa.ctypes.data_as(C.c_void_p),
                                           fa.ctypes.data_as(C.c_void_p)))
    return pa[:tot], fa[:tot]


def dist_pack_clouds(corners, surfs, poses6) -> np.ndarray:
    """one rank's message of the epoch merge step (loamx_dist_pack_clouds): per stream its re-projected corner / surf clouds and its
    transformAftMapped -> uint32 words"""
    cs = [as_points(c) for c in corners]
    ss = [as_points(c) for c in surfs]
    n = len(cs)
    assert len(ss) == n
    p = np.ascontiguousarray(poses6, np.float32).reshape(n, 6) if n else np.zeros((0, 6), np.float32)
    CA = Cloud * max(n, 1)
    ca, sa = CA(*[cloud_of(c) for c in cs]), CA(*[cloud_of(c) for c in ss])
    nw = C.c_uint64(0)
    _check(lib().loamx_dist_pack_clouds(n, C.addressof(ca), C.addressof(sa), p.ctypes.data, None, 0, C.addressof(nw)))
    out = np.zeros(int(nw.value), np.uint32)
    _check(lib().loamx_dist_pack_clouds(n, C.addressof(ca), C.addressof(sa), p.ctypes.data, out.ctypes.data, len(out), C.addressof(nw)))
    return out


def dist_unpack_clouds(words):
    """inverse of dist_pack_clouds -> [(pose6, corner, surf)] per stream"""
    w = np.ascontiguousarray(words, np.uint32)
    ns = C.c_uint32(0)
    _check(lib().loamx_dist_unpack_clouds_header(w.ctypes.data, len(w), C.addressof(ns), None, None, 0))
    n = int(ns.value)
    nc, nsf = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
    _check(lib().loamx_dist_unpack_clouds_header(w.ctypes.data, len(w), C.addressof(ns), nc.ctypes.data, nsf.ctypes.data, max(n, 1)))
    out = []
    for s in range(n):
        pose = np.zeros(6, np.float32)
        co, so = np.zeros((max(int(nc[s]), 1), 4), np.float32), np.zeros((max(int(nsf[s]), 1), 4), np.float32)
        cc, sc = cloud_of(co), cloud_of(so)
        _check(lib().loamx_dist_unpack_clouds_stream(w.ctypes.data, len(w), s, pose.ctypes.data, C.addressof(cc), C.addressof(sc)))
        out.append((pose, co[:cc.count].copy(), so[:sc.count].copy()))
    return out


def dist_split_messages(words, counts):
    """the root's receive buffer of a gatherv -> one message per rank (rank order; an empty rank -> None)"""
    out, o = [], 0
    for c in counts:
        c = int(c)
        out.append(np.ascontiguousarray(words[o:o + c]) if c else None)
        o += c
    return out


class Dist:
    """loamx_dist_*: the multi-GPU exchanges of the batched mode over RCCL (one process per GPU)."""
    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * Dist.ID_BYTES)()
        _check(lib().loamx_dist_get_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id: bytes, rank: int, world_size: int, device: int = 0):
        assert len(unique_id) == self.ID_BYTES
        L = lib()
        L.loamx_dist_create.restype = C.c_void_p
        L.loamx_dist_stream.restype = C.c_void_p
        buf = (C.c_ubyte * self.ID_BYTES).from_buffer_copy(unique_id)
        self.h = C.c_void_p(L.loamx_dist_create(buf, rank, world_size, device))
        if not self.h:
            raise LoamxError(E_INVALID, L.loamx_last_error().decode())
        self.rank, self.world = rank, world_size

    def close(self):
        if getattr(self, "h", None):
            lib().loamx_dist_destroy(self.h)
            self.h = None

    __del__ = close

    def shard(self, batch: int):
        b, e = C.c_uint32(), C.c_uint32()
        _check(lib().loamx_dist_shard(self.h, batch, C.byref(b), C.byref(e)))
        return int(b.value), int(e.value)

    def broadcast_map(self, d_corner_ptr: int, n_corner: int, d_surf_ptr: int, n_surf: int, root: int = 0, wait_event: int = 0) -> int:
        """asynchronous; returns the raw hipEvent_t recorded behind the broadcast (for stage_frozen_device)"""
        ev = C.c_void_p()
        _check(lib().loamx_dist_broadcast_map(self.h, C.c_void_p(d_corner_ptr), n_corner, C.c_void_p(d_surf_ptr), n_surf, root,
                                              C.c_void_p(wait_event or None), C.byref(ev)))
        return int(ev.value or 0)

    def allgather_results(self, poses6, iters_flags=None, batch=None):
        """every rank's records in rank order; shards may be unequal, also empty.  The counts are gathered first
        (loamx_dist_allgather_counts) and the receive arrays sized by them; batch (optional) = the total the caller expects.  The C call is told the
        capacity either way and answers LOAMX_E_CAPACITY instead of writing beyond it.  Returns (poses, flags, counts per rank)"""
        p = np.ascontiguousarray(poses6, np.float32).reshape(-1, 6)
        n = len(p)
        f = np.ascontiguousarray(iters_flags, np.int32).reshape(n, 2) if iters_flags is not None else None
        cnt = np.zeros(self.world, np.uint32)
        # the counts are ALWAYS gathered: whether a rank passes `batch` must not decide which collectives it takes part in (ranks that
        # disagreed would hang in mismatched collectives); `batch` is only checked against them
        _check(lib().loamx_dist_allgather_counts(self.h, n, cnt.ctypes.data_as(C.c_void_p)))
        cap = int(cnt.sum())
        # (a `batch` that disagrees is reported AFTER the results collective: a rank that left the protocol between the two collectives
        # would leave the others blocked inside the second one)
        pa = np.zeros((max(cap, 1), 6), np.float32)
        fa = np.zeros((max(cap, 1), 2), np.int32)
        _check(lib().loamx_dist_allgather_results_cap(self.h, p.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p) if f is not None else None,
                                                      n, pa.ctypes.data_as(C.c_void_p), fa.ctypes.data_as(C.c_void_p), cap, cnt.ctypes.data_as(C.c_void_p)))
        tot = int(cnt.sum())
        if batch is not None and int(batch) != cap:
            raise ValueError(f"allgather_results: batch = {batch} but the ranks hold {cap} records")
        return pa[:tot], fa[:tot], cnt

    def allgather_counts(self, n: int) -> np.ndarray:
        """every rank's n (loamx_dist_allgather_counts): also the way a root tells the others a size"""
        cnt = np.zeros(self.world, np.uint32)
        _check(lib().loamx_dist_allgather_counts(self.h, int(n), cnt.ctypes.data_as(C.c_void_p)))
        return cnt

    def gatherv(self, words, root: int = 0):
        """variable-size gather of uint32 words to `root` (loamx_dist_gatherv): returns (messages per rank | None off the root, counts)"""
        w = np.ascontiguousarray(words if words is not None else np.zeros(0, np.uint32), np.uint32)
        cnt = np.zeros(self.world, np.uint32)
        # the root cannot size its buffer before it knows the counts: they are gathered by a first call with nothing to receive into
        _check(lib().loamx_dist_allgather_counts(self.h, len(w), cnt.ctypes.data_as(C.c_void_p)))
        tot = int(cnt.sum())
        recv = np.zeros(max(tot, 1), np.uint32) if self.rank == root else None
        _check(lib().loamx_dist_gatherv(self.h, w.ctypes.data, len(w), root, recv.ctypes.data if recv is not None else None,
                                        tot if recv is not None else 0, cnt.ctypes.data))
        return (dist_split_messages(recv[:tot], cnt) if recv is not None else None), cnt

    def comm_count(self) -> int:
        return int(lib().loamx_dist_comm_count(self.h))

    def barrier(self):
        _check(lib().loamx_dist_barrier(self.h))

    @property
    def stream(self) -> int:
        return int(lib().loamx_dist_stream(self.h) or 0)
