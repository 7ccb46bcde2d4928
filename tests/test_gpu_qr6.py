"""GPU: the 6x6 column-pivoted Householder QR solve of the update steps (colPivHouseholderQr().solve — BasicLaserMapping.cpp:867,
BasicLaserOdometry.cpp:559).  The kernels run it spread over the lanes of a wave (qr_solve6_coop: columns stay in their lanes, a pivot
swap is a swap of lane numbers); it must equal the scalar routine — the reference's order of operations on every element — bit for
bit, on well-conditioned, rank-deficient and badly scaled systems alike, and the scalar routine must equal the oracle's."""
import numpy as np
import pytest

from loam_velodyne_amd import loamx

pytestmark = pytest.mark.gpu


def _systems(seed):
    rng = np.random.default_rng(seed)
    A, b = [], []
    for k in range(600):   # normal equations of random Jacobians (what the update steps solve), rotations small, translations large
        J = rng.normal(size=(int(rng.integers(6, 400)), 6)) * np.array([30, 30, 30, 1, 1, 1]) * 10.0 ** rng.integers(-3, 3)
        r = rng.normal(size=len(J))
        A.append((J.T @ J).astype(np.float32)); b.append((J.T @ r).astype(np.float32))
    for k in range(200):   # rank-deficient: a repeated column, a zero column, a planar scene (no constraint along one axis)
        J = rng.normal(size=(50, 6))
        mode = k % 4
        if mode == 0: J[:, 3] = J[:, 1]
        elif mode == 1: J[:, int(rng.integers(0, 6))] = 0.0
        elif mode == 2: J[:, 4] = 1e-9 * J[:, 4]
        else: J[:, 2] = J[:, 0] + J[:, 5]
        r = rng.normal(size=50)
        A.append((J.T @ J).astype(np.float32)); b.append((J.T @ r).astype(np.float32))
    for k in range(100):   # general (non-symmetric) matrices, ties between column norms, tiny and huge entries
        M = rng.normal(size=(6, 6)) * 10.0 ** rng.integers(-8, 8)
        if k % 5 == 0: M = np.round(M / np.abs(M).max() * 2)          # many equal norms: the FIRST largest column must win
        A.append(M.astype(np.float32)); b.append(rng.normal(size=6).astype(np.float32))
    A.append(np.zeros((6, 6), np.float32)); b.append(np.ones(6, np.float32))
    A.append(np.eye(6, dtype=np.float32)); b.append(np.arange(6, dtype=np.float32))
    return np.stack(A), np.stack(b)


def test_wave_cooperative_solve_equals_the_scalar_routine_bit_for_bit(orc):
    A, b = _systems(5)
    h = loamx.Batch(1)
    xc, xs = h.qr6_probe(A, b)
    h.close()
    same = (xc.view(np.uint32) == xs.view(np.uint32)) | (np.isnan(xc) & np.isnan(xs))   # (the all-zero system: 0 / 0, a NaN either way)
    assert same.all(), f"{(~same).any(axis=1).sum()} of {len(A)} systems differ, first {np.argwhere(~same)[0]}"
    assert np.isfinite(xs[:800]).all()
    n_def = int((xs[600:800] == 0).any(axis=1).sum())
    assert n_def >= 100, n_def            # rank-deficient systems really took the truncated back substitution
    for i in list(range(0, 900, 7)) + [900, 901]:   # ... and the scalar routine is the oracle's
        xo = orc.qr_solve(A[i].astype(np.float32), b[i].astype(np.float32))
        xo = np.asarray(xo, np.float32)
        assert np.all((xo.view(np.uint32) == xs[i].view(np.uint32)) | (np.isnan(xo) & np.isnan(xs[i]))), i
