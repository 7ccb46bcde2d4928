"""GPU: the kNN contract (SURVEY.md §8 "kNN contract": nanoflann_pcl.h:140-152, nanoflann.hpp:115-139, :372-379) — the library's own
neighbour search (loamx_batch_knn_probe runs knn5_group, the device routine inside the Gauss-Newton kernel k_gn_iter) against
  * the oracle's kd-tree (leaf 10, exact; pinned against the reference's nanoflann in tests/test_ref_pinning.py),
  * the reference's own nanoflann.hpp where oracle/_ref/libref_nanoflann.so was shipped,
  * brute force with the (distance, index) order.
Indices and float distances must be equal, not close.  The one documented difference: among candidates at EXACTLY equal distance
nanoflann keeps the one its tree visits first, the device the one with the lower index — so on the tie-heavy lattice the distances
are compared for all five neighbours and the indices for the neighbours strictly inside the fifth distance."""
import numpy as np
import pytest

import oracle_py as op
from loam_velodyne_amd import loamx, synth

pytestmark = pytest.mark.gpu

COVER2 = np.float32(1.05 * 1.05 * 0.9999)      # the search radius of the device routine (covers the reference's 1 m gate)
MISSING = np.uint32(0xFFFFFFFF)


def _d2_f32(pts, q):
    d = pts[None, :, :3].astype(np.float32) - q[:, None, :3].astype(np.float32)
    d = -d                                     # q - p, as the device forms it (the squares are the same)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def _compare_exact(idx, d2, oidx, od2, what):
    """oracle / reference results (k nearest of the whole cloud) vs the probe (neighbours inside the covered radius)"""
    inside = od2 < np.float32(1.10)
    beyond = od2 > np.float32(1.1026)
    assert np.array_equal(idx[inside], oidx[inside].astype(np.uint32)), what
    assert np.array_equal(d2[inside], od2[inside]), what
    assert np.all(idx[beyond] == MISSING), what
    return int(inside.sum()), int(beyond.sum())


def _queries(orc, world, sensor, n_sweeps, seed):
    """registered feature points: what the Gauss-Newton iterations really look up"""
    rng = np.random.default_rng(seed)
    sr = op.ScanRegistration(orc)
    qc, qs = [], []
    for k in range(n_sweeps):
        gt = np.array([0.01 * rng.normal(), 0.3 * rng.normal(), 0.01 * rng.normal(), 3 * rng.normal(), 0.05 * rng.normal(), 3 * rng.normal()])
        sw = synth.make_sweep(world, sensor, gt, gt, seed=seed * 100 + k)
        f = sr.process(sw.points, sw.ring_sizes)
        R = synth.rot_zxy(*gt[:3])
        for src, dst in ((f["less_sharp"], qc), (f["less_flat"], qs)):
            p = src[:, :3].astype(np.float64) @ R.T + gt[3:] + rng.normal(0, 0.02, (len(src), 3))
            dst.append(p.astype(np.float32))
    return np.concatenate(qc), np.concatenate(qs)[::3]


@pytest.mark.parametrize("sensor,map_points,half", [("HDL-64E", 1_000_000, 125.0), ("VLP-16", 100_000, 65.0)])
def test_neighbour_indices_equal_the_kd_tree(orc, sensor, map_points, half):
    world = synth.World(half_extent=half)
    cm, sm = world.make_map(map_points)
    b = loamx.Batch(1)
    b.set_frozen(cm, sm)
    qc, qs = _queries(orc, world, sensor, 2, seed=11)
    far = np.float32([[1e4, 0, 0], [-300.0, 5.0, 7.0], [0.0, 80.0, 0.0]])          # outside the grid: nothing within the gate
    for which, pts, q in ((0, cm, np.concatenate([qc, far])), (1, sm, np.concatenate([qs, far]))):
        idx, d2 = b.knn_probe(which, q)
        oidx, od2 = orc.knn(pts, np.c_[q, np.zeros(len(q), np.float32)], 5)
        n_in, n_out = _compare_exact(idx, d2, oidx, od2, (sensor, which, "oracle kd-tree"))
        assert n_in > 4 * len(q) * 0.8                                              # the comparison is not vacuous
        assert np.all(idx[-3:] == MISSING) and np.all(d2[-3:] == np.finfo(np.float32).max)
        ref = op.ref_knn(pts, np.c_[q, np.zeros(len(q), np.float32)], 5)
        if ref is not None:
            _compare_exact(idx, d2, ref[0], ref[1], (sensor, which, "reference nanoflann"))
        # the gate of BasicLaserMapping.cpp:671 / :760 decided from the probe == decided from the kd-tree
        assert np.array_equal(d2[:, 4] < 1.0, od2[:, 4] < 1.0)


def test_tie_heavy_lattice(orc):
    """a 0.25 m lattice: every query has many neighbours at exactly equal distances"""
    g = np.arange(-6.0, 6.0, 0.25, dtype=np.float32)
    X, Y, Z = np.meshgrid(g, g[:24], g, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel(), np.zeros(X.size, np.float32)], 1).astype(np.float32)
    rng = np.random.default_rng(3)
    pts = pts[rng.permutation(len(pts))]                                            # index order unrelated to position
    base = pts[rng.integers(0, len(pts), 1500), :3]
    q = np.concatenate([base, base + np.float32(0.125), base + np.float32([0.125, 0, 0]), base + np.float32([0.125, 0.125, 0])]).astype(np.float32)
    b = loamx.Batch(1)
    b.set_frozen(pts[:2000], pts)                                                   # (the corner sub-map is not used here)
    idx, d2 = b.knn_probe(1, q)
    # brute force, (distance, index) order: exact equality
    for lo in range(0, len(q), 500):
        D = _d2_f32(pts, q[lo:lo + 500])
        order = np.lexsort((np.broadcast_to(np.arange(len(pts)), D.shape), D), axis=1)[:, :5]
        assert np.array_equal(idx[lo:lo + 500], order.astype(np.uint32))
        assert np.array_equal(d2[lo:lo + 500], np.take_along_axis(D, order, 1))
    # kd-trees: same distances; same indices strictly inside the fifth distance (tie order among equals is the visiting order there)
    q4 = np.c_[q, np.zeros(len(q), np.float32)]
    for name, res in (("oracle", orc.knn(pts, q4, 5)), ("reference", op.ref_knn(pts, q4, 5))):
        if res is None:
            continue
        oidx, od2 = res
        assert np.array_equal(d2, od2), name
        strict = od2 < od2[:, 4:5]
        for i in np.flatnonzero(strict.any(1))[:2000]:
            assert set(idx[i][strict[i]].tolist()) == set(oidx[i][strict[i]].tolist()), (name, i)
