#!/usr/bin/env python3
"""Executable specification (NumPy, no GPU) of the bucketed-cell 5-NN search proposed in DESIGN.md §8 for k_knn5, with a
self-test of its EXACTNESS and a count of the memory lines a query touches.

Layout: cell edge h; every occupied cell owns one 128-byte line of up to 8 inline points (x, y, z, original index); a fuller
cell chains overflow lines.  The address of a cell's line follows from the cell coordinates (dense table of line ids).
Search (mapping's contract: the five nearest by squared distance, accepted only if the fifth is closer than 1 m):
  phase 1  the 2x2x2 block of cells nearest to the query (per axis: its own cell and the neighbour across the nearer face) —
           it contains the ball of radius h/2 around the query; all 8 lines can be requested at once;
  phase 2  with r^2 = min(1, sixth-best d^2 so far): every other cell whose box lies within r of the query (<=, so that ties
           on the distance are still ranked by original index); again independent lines;
  result   the six smallest (d^2, original index) keys — five neighbours + the sixth distance that re-validation needs.
Run:  python scripts/knn_bucket_model.py [n_map_points] [cell_edge]"""
import sys

import numpy as np


class BucketGrid:
    LINE = 8

    def __init__(self, pts, h):
        self.p = np.ascontiguousarray(pts[:, :3], np.float32)
        self.h = np.float32(h)
        self.inv = np.float32(1.0) / self.h
        self.lo = np.floor(self.p.min(0) * self.inv).astype(np.int64)              # cell coordinate of the bbox corner
        c = np.floor(self.p * self.inv).astype(np.int64) - self.lo                    # float32 multiply, as the device would bin
        self.dims = c.max(0) + 1
        key = (c[:, 2] * self.dims[1] + c[:, 1]) * self.dims[0] + c[:, 0]
        order = np.argsort(key, kind="stable")                                        # points of a cell keep their input order
        self.sorted_idx = order
        ks = key[order]
        self.cell_first = np.full(int(self.dims.prod()) + 1, -1, np.int64)
        start = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        self.cells = ks[start]
        self.cell_beg = dict(zip(self.cells.tolist(), start.tolist()))
        self.cell_end = dict(zip(self.cells.tolist(), np.r_[start[1:], len(ks)].tolist()))

    def cell_points(self, cx, cy, cz):
        """original indices of the points of one cell and the number of 128-byte lines they occupy"""
        if not (0 <= cx < self.dims[0] and 0 <= cy < self.dims[1] and 0 <= cz < self.dims[2]):
            return np.empty(0, np.int64), 0
        k = int((cz * self.dims[1] + cy) * self.dims[0] + cx)
        if k not in self.cell_beg:
            return np.empty(0, np.int64), 0
        idx = self.sorted_idx[self.cell_beg[k]:self.cell_end[k]]
        return idx, -(-len(idx) // self.LINE)

    def knn6(self, q):
        q = np.asarray(q, np.float32)
        g = q * self.inv
        c = np.floor(g).astype(np.int64) - self.lo
        frac = g - np.floor(g)
        side = np.where(frac < 0.5, -1, 1)
        seen, cand, lines = set(), [], [0, 0]

        def visit(cx, cy, cz, phase):
            if (cx, cy, cz) in seen:
                return
            seen.add((cx, cy, cz))
            idx, n = self.cell_points(cx, cy, cz)
            lines[phase] += n
            if len(idx):
                d = self.p[idx] - q                                                  # float32, (dx^2 + dy^2) + dz^2 as nanoflann sums
                cand.extend(zip(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).tolist(), idx.tolist()))

        for dz in (0, side[2]):
            for dy in (0, side[1]):
                for dx in (0, side[0]):
                    visit(int(c[0] + dx), int(c[1] + dy), int(c[2] + dz), 0)
        cand.sort()
        r2 = min(1.0, cand[5][0]) if len(cand) >= 6 else 1.0
        r = float(np.sqrt(np.float32(r2))) * (1 + 1e-6) + 1e-6                        # margins err towards visiting
        lo_c = np.floor((q - r) * self.inv).astype(np.int64) - self.lo
        hi_c = np.floor((q + r) * self.inv).astype(np.int64) - self.lo
        for cz in range(lo_c[2], hi_c[2] + 1):
            for cy in range(lo_c[1], hi_c[1] + 1):
                for cx in range(lo_c[0], hi_c[0] + 1):
                    if (cx, cy, cz) in seen:
                        continue
                    bmin = (np.array([cx, cy, cz]) + self.lo) * float(self.h)
                    gap = np.maximum(np.maximum(bmin - q, 0), q - (bmin + float(self.h)))
                    if float((gap * gap).sum()) <= r2 * (1 + 1e-5) + 1e-9:
                        visit(cx, cy, cz, 1)
        cand.sort()
        return cand[:6], lines


def self_test(n_map=200000, h=0.8, n_query=1500, seed=0):
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from loam_velodyne_amd import synth
    w = synth.World(half_extent=125.0)
    scale = (n_map / 1e6) ** (1 / 3)                                                 # keep the bench's point density in a smaller hall
    cm, sm = w.make_map(n_map, half_extent=124.5 * scale)
    pts = sm[:, :3].astype(np.float32)
    rng = np.random.default_rng(seed)
    base = pts[rng.integers(0, len(pts), n_query)]
    queries = (base + rng.normal(0, 0.25, base.shape)).astype(np.float32)            # near surfaces, like registered features
    queries[::50] += 30.0                                                            # and some far from everything
    grid = BucketGrid(pts, h)
    occ = np.array([grid.cell_end[k] - grid.cell_beg[k] for k in grid.cells.tolist()])
    print(f"map {len(pts)} pts, cell {h} m: {len(occ)} occupied cells, mean {occ.mean():.1f} pts/cell, "
          f"{(occ > BucketGrid.LINE).mean() * 100:.1f} % of cells overflow one line, table {int(grid.dims.prod())} cells")
    lines1, lines2, accepted = [], [], 0
    for q in queries:
        got, lines = grid.knn6(q)
        d = pts - q
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        order = np.lexsort((np.arange(len(pts)), d2))[:6]
        want = list(zip(d2[order].tolist(), order.tolist()))
        if want[4][0] < 1.0:                      # the reference only uses the neighbours when the fifth is closer than 1 m
            accepted += 1
            assert [g[1] for g in got[:5]] == [x[1] for x in want[:5]], (q, got, want)
            assert np.float32(got[4][0]) == np.float32(want[4][0])
            if want[5][0] < 1.0:
                assert got[5] == want[5]           # the sixth distance (re-validation bound) is exact too
        else:
            assert len(got) < 5 or got[4][0] >= 1.0
        lines1.append(lines[0]); lines2.append(lines[1])
    print(f"{n_query} queries: exact for all {accepted} accepted ones; lines per query  phase 1 mean {np.mean(lines1):.1f} (max {max(lines1)})"
          f"  phase 2 mean {np.mean(lines2):.1f} (max {max(lines2)})")


if __name__ == "__main__":
    self_test(int(sys.argv[1]) if len(sys.argv) > 1 else 200000, float(sys.argv[2]) if len(sys.argv) > 2 else 0.8)
