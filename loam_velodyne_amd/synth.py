"""Seeded synthetic sweeps and feature maps (harness utility for tests/ and bench.py).

Everything is in the LOAM camera frame the Basic* classes work in (x left, y up, z forward — i.e. after the axis
remap of reference src/lib/MultiScanRegistration.cpp:182-184), so a sweep produced here is exactly what
`BasicScanRegistration::processScanlines` receives: one cloud per scan ring, points in firing order, intensity =
ring id + relative time within the sweep (MultiScanRegistration.cpp:229).

World: a closed hall (ground, ceiling, four outer walls) with a lattice of axis-aligned box pillars that reach
from the ground to the ceiling, so every beam returns.  Pose convention = the reference's
`pointAssociateToMap` (BasicLaserMapping.cpp:207-219): p_map = R_y(ry) R_x(rx) R_z(rz) p_sensor + t.
"""
from __future__ import annotations

import dataclasses
import numpy as np

# (rings, azimuth steps, lowest, highest elevation in degrees) — reference MultiScanRegistration.h:83-89
SENSORS = {
    "VLP-16": (16, 1800, -15.0, 15.0),
    "HDL-32": (32, 2048, -30.67, 10.67),
    "HDL-64E": (64, 2048, -24.9, 2.0),
}


def rot_zxy(rx, ry, rz):
    """3x3 matrix of rotateZXY (reference math_utils.h:212-238): R = Ry @ Rx @ Rz."""
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    Rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    return Ry @ Rx @ Rz


@dataclasses.dataclass
class World:
    half_extent: float = 125.0      # outer walls at x,z = +-half_extent
    ground_y: float = -1.8
    ceil_y: float = 10.2
    pitch: float = 20.0             # pillar lattice pitch
    seed: int = 20240601
    boxes: np.ndarray = None        # (nb, 4): xmin, xmax, zmin, zmax

    def __post_init__(self):
        rng = np.random.default_rng(self.seed)
        n = int(np.floor((self.half_extent - 10.0) / self.pitch))
        cs = (np.arange(-n, n + 1) + 0.5) * self.pitch          # lattice centres, none on x=0 / z=0 corridor
        cx, cz = np.meshgrid(cs, cs, indexing="ij")
        cx, cz = cx.ravel(), cz.ravel()
        hx = rng.uniform(1.5, 3.5, cx.size)
        hz = rng.uniform(1.5, 3.5, cx.size)
        cx = cx + rng.uniform(-1.0, 1.0, cx.size)
        cz = cz + rng.uniform(-1.0, 1.0, cx.size)
        self.boxes = np.stack([cx - hx, cx + hx, cz - hz, cz + hz], axis=1)

    # ---- ray casting -------------------------------------------------------------------------------------------
    def cast(self, o: np.ndarray, d: np.ndarray) -> np.ndarray:
        """o,d: (N,3) float64 world-frame origins / unit directions -> range t (N,)."""
        n = o.shape[0]
        t = np.full(n, np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            # ground / ceiling
            for y in (self.ground_y, self.ceil_y):
                ty = (y - o[:, 1]) / d[:, 1]
                ty[~(ty > 1e-6)] = np.inf
                t = np.minimum(t, ty)
            # outer walls (inside of a box: exit distance)
            E = self.half_extent
            for a in (0, 2):
                tw = np.where(d[:, a] > 0, (E - o[:, a]) / d[:, a], (-E - o[:, a]) / d[:, a])
                tw[~(tw > 1e-6)] = np.inf
                t = np.minimum(t, tw)
            # pillars: 2-D slab test in (x,z); they span the full height
            bx = self.boxes
            chunk = 1024   # (rays x pillars temporaries stay in cache: 2.5 x faster than 16384, same values — every ray is independent)
            for s in range(0, n, chunk):
                oo, dd = o[s:s + chunk], d[s:s + chunk]
                inv_x, inv_z = 1.0 / dd[:, 0:1], 1.0 / dd[:, 2:3]
                t1 = (bx[None, :, 0] - oo[:, 0:1]) * inv_x
                t2 = (bx[None, :, 1] - oo[:, 0:1]) * inv_x
                t3 = (bx[None, :, 2] - oo[:, 2:3]) * inv_z
                t4 = (bx[None, :, 3] - oo[:, 2:3]) * inv_z
                tmin = np.maximum(np.minimum(t1, t2), np.minimum(t3, t4))
                tmax = np.minimum(np.maximum(t1, t2), np.maximum(t3, t4))
                hit = (tmax >= tmin) & (tmin > 1e-6)
                tb = np.where(hit, tmin, np.inf).min(axis=1)
                t[s:s + chunk] = np.minimum(t[s:s + chunk], tb)
        return t

    # ---- analytic feature map ----------------------------------------------------------------------------------
    def make_map(self, n_points: int, corner_fraction: float = 0.1, half_extent: float | None = None,
                 noise: float = 0.01, seed: int = 7):
        """Feature map of exactly n_points in the map (= world) frame: corner points on pillar edges, surface points
        on ground / ceiling / pillar faces, restricted to |x|,|z| <= half_extent.  Returns (corner (Mc,4), surf (Ms,4))
        float32 with intensity 0 (mapping-side intensities are ring ids, irrelevant to registration)."""
        rng = np.random.default_rng(seed)
        E = half_extent or (self.half_extent - 0.5)
        n_corner = int(round(n_points * corner_fraction))
        n_surf = n_points - n_corner
        bx = self.boxes[(np.abs(self.boxes[:, :2]).max(axis=1) < E) & (np.abs(self.boxes[:, 2:]).max(axis=1) < E)]
        H = self.ceil_y - self.ground_y

        # -- corners: 4 vertical edges + 4 bottom + 4 top edges per pillar
        seg = []
        for xa, za in ((0, 2), (0, 3), (1, 2), (1, 3)):
            p0 = np.stack([bx[:, xa], np.full(len(bx), self.ground_y), bx[:, za]], 1)
            p1 = p0.copy(); p1[:, 1] = self.ceil_y
            seg.append((p0, p1))
        for y in (self.ground_y, self.ceil_y):
            for za in (2, 3):
                seg.append((np.stack([bx[:, 0], np.full(len(bx), y), bx[:, za]], 1),
                            np.stack([bx[:, 1], np.full(len(bx), y), bx[:, za]], 1)))
            for xa in (0, 1):
                seg.append((np.stack([bx[:, xa], np.full(len(bx), y), bx[:, 2]], 1),
                            np.stack([bx[:, xa], np.full(len(bx), y), bx[:, 3]], 1)))
        p0 = np.concatenate([s[0] for s in seg]); p1 = np.concatenate([s[1] for s in seg])
        length = np.linalg.norm(p1 - p0, axis=1)
        corner = _sample_weighted(rng, length, n_corner, lambda k, u: p0[k] + (p1[k] - p0[k]) * u[:, :1], 1)
        corner += rng.normal(0, noise, corner.shape)

        # -- surfaces: ground, ceiling (minus pillar footprints), pillar faces
        faces = []   # (origin, edge_u, edge_v)
        for y in (self.ground_y, self.ceil_y):
            faces.append((np.array([[-E, y, -E]]), np.array([[2 * E, 0, 0.0]]), np.array([[0.0, 0, 2 * E]])))
        nb = len(bx)
        for xa in (0, 1):
            faces.append((np.stack([bx[:, xa], np.full(nb, self.ground_y), bx[:, 2]], 1),
                          np.tile([[0.0, H, 0.0]], (nb, 1)), np.stack([np.zeros(nb), np.zeros(nb), bx[:, 3] - bx[:, 2]], 1)))
        for za in (2, 3):
            faces.append((np.stack([bx[:, 0], np.full(nb, self.ground_y), bx[:, za]], 1),
                          np.tile([[0.0, H, 0.0]], (nb, 1)), np.stack([bx[:, 1] - bx[:, 0], np.zeros(nb), np.zeros(nb)], 1)))
        fo = np.concatenate([f[0] for f in faces]); fu = np.concatenate([f[1] for f in faces]); fv = np.concatenate([f[2] for f in faces])
        area = np.linalg.norm(np.cross(fu, fv), axis=1)
        surf = np.zeros((0, 3))
        while len(surf) < n_surf:
            cand = _sample_weighted(rng, area, int((n_surf - len(surf)) * 1.2) + 16,
                                    lambda k, u: fo[k] + fu[k] * u[:, :1] + fv[k] * u[:, 1:2], 2)
            # drop ground/ceiling samples under a pillar footprint
            flat = (np.abs(cand[:, 1] - self.ground_y) < 1e-9) | (np.abs(cand[:, 1] - self.ceil_y) < 1e-9)
            inside = np.zeros(len(cand), bool)
            idx = np.nonzero(flat)[0]
            for s in range(0, len(idx), 65536):
                ii = idx[s:s + 65536]
                c = cand[ii]
                inside[ii] = ((c[:, None, 0] > bx[None, :, 0]) & (c[:, None, 0] < bx[None, :, 1]) &
                              (c[:, None, 2] > bx[None, :, 2]) & (c[:, None, 2] < bx[None, :, 3])).any(axis=1)
            surf = np.concatenate([surf, cand[~inside]])
        surf = surf[:n_surf] + rng.normal(0, noise, (n_surf, 3))

        def pack(p):
            out = np.zeros((len(p), 4), np.float32)
            out[:, :3] = p
            return out
        return pack(corner), pack(surf)


def _sample_weighted(rng, weights, n, fn, nu):
    k = rng.choice(len(weights), size=n, p=weights / weights.sum())
    u = rng.random((n, nu))
    return fn(k, u)


@dataclasses.dataclass
class Sweep:
    points: np.ndarray        # (N,4) float32, rings concatenated in ring order
    ring_sizes: np.ndarray    # (R,) int32
    pose_start: np.ndarray    # (6,) rx ry rz x y z at sweep start (world frame)
    pose_end: np.ndarray


def make_sweep(world: World, sensor: str, pose_start, pose_end, noise: float = 0.01, seed: int = 0,
               scan_period: float = 0.1, az_steps: int | None = None) -> Sweep:
    """One sweep captured while the sensor moves linearly (in the 6 pose parameters) from pose_start to pose_end.
    Every point is expressed in the sensor frame at ITS OWN firing time (raw, motion-distorted), as a real driver
    delivers it; intensity = ring + scan_period * sweep fraction."""
    R, A, lo, hi = SENSORS[sensor]
    if az_steps:
        A = az_steps
    rng = np.random.default_rng(20240601 + seed)
    pose_start = np.asarray(pose_start, np.float64)
    pose_end = np.asarray(pose_end, np.float64)
    elev = np.deg2rad(np.linspace(lo, hi, R))
    frac = np.arange(A) / A
    ori = -np.pi + 2 * np.pi * frac                     # ori = -atan2(x, z), increasing with time
    # sensor-frame unit directions (R, A, 3)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d_s = np.stack([-np.sin(ori)[None, :] * ce, np.broadcast_to(se, (R, A)), np.cos(ori)[None, :] * ce], axis=-1)
    # per-azimuth pose
    poses = pose_start[None, :] + frac[:, None] * (pose_end - pose_start)[None, :]
    Rm = np.stack([rot_zxy(p[0], p[1], p[2]) for p in poses])          # (A,3,3)
    d_w = np.einsum("aij,raj->rai", Rm, d_s)
    o_w = np.broadcast_to(poses[None, :, 3:6], (R, A, 3))
    t = world.cast(o_w.reshape(-1, 3).copy(), d_w.reshape(-1, 3)).reshape(R, A)
    t = t + rng.normal(0, noise, t.shape)
    p_s = d_s * t[..., None]
    inten = np.arange(R)[:, None] + scan_period * frac[None, :]
    pts = np.concatenate([p_s, inten[..., None]], axis=-1).astype(np.float32).reshape(-1, 4)
    return Sweep(pts, np.full(R, A, np.int32), pose_start.astype(np.float32), pose_end.astype(np.float32))


_JOB_WORLDS = {}


def make_sweep_job(job):
    """(half_extent, sensor, pose_start, pose_end, seed) -> (points, ring_sizes): make_sweep for a worker process (bench.py generates
    its sweeps in parallel); the world of a given extent is built once per process."""
    half_extent, sensor, pose_start, pose_end, seed = job
    w = _JOB_WORLDS.get(half_extent)
    if w is None:
        w = _JOB_WORLDS[half_extent] = World(half_extent=half_extent)
    sw = make_sweep(w, sensor, pose_start, pose_end, seed=seed)
    return sw.points, sw.ring_sizes


def to_raw(sweep: Sweep, bad_every: int = 0) -> np.ndarray:
    """The sweep as a Velodyne driver delivers it (what MultiScanRegistration::process consumes): (N,3) float32 in SENSOR
    axes (x forward, y left, z up — the inverse of the remap at MultiScanRegistration.cpp:184-186) in firing order
    (azimuth-major, all lasers of one firing together).  bad_every > 0 plants a NaN return, a zero return and a return far
    outside the vertical field of view at every bad_every-th firing, as real packets contain them."""
    R = len(sweep.ring_sizes)
    A = int(sweep.ring_sizes[0])
    assert np.all(sweep.ring_sizes == A)
    p = sweep.points.reshape(R, A, 4)
    raw = np.stack([p[..., 2], p[..., 0], p[..., 1]], axis=-1).transpose(1, 0, 2).copy()   # (A, R, 3)
    if bad_every:
        for a in range(bad_every // 2, A - 1, bad_every):   # (never the first / last return: they define startOri / endOri)
            raw[a, 0 % R] = np.nan
            raw[a, 1 % R] = 0.0
            raw[a, 2 % R] = (0.1, 0.1, 50.0)
    return raw.reshape(-1, 3).astype(np.float32)


def trajectory(n_sweeps: int, step: float = 1.0, yaw_step_deg: float = 0.5, start=(0.0, 0.0, 0.0), n_static: int = 1):
    """Sweep boundary poses (n_sweeps+1, 6).  The first `n_static` sweeps are taken at rest (the reference keeps the
    very first sweep un-de-skewed, BasicLaserOdometry.cpp:198-211, so a moving first sweep would smear the map);
    afterwards constant forward speed `step` m/sweep along the heading, constant yaw rate, and a small sinusoidal
    pitch / roll / height so all six degrees of freedom are exercised."""
    poses = np.zeros((n_sweeps + 1, 6))
    x, y, z = start
    yaw = 0.0
    for k in range(n_sweeps + 1):
        m = max(0, k - n_static)
        poses[k] = [0.004 * np.sin(0.7 * m), yaw, 0.003 * np.sin(0.5 * m), x, y + 0.02 * np.sin(0.3 * m), z]
        if k >= n_static:
            x += step * np.sin(yaw)
            z += step * np.cos(yaw)
            yaw += np.deg2rad(yaw_step_deg)
    return poses
