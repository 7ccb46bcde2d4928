// Sweep-to-sweep odometry for gfx950 — reference src/lib/BasicLaserOdometry.cpp.
//
// The problem is small (<= 768 sharp + 1536 flat features against the previous sweep's <= ~60 k feature points) and
// strictly iterative (<= 25 dependent Gauss-Newton steps), i.e. latency-bound.  Per group of 5 iterations, for all
// streams of a batch at once:
//   k_odom_corr    (:250-302, :368-435)  one wave per feature: transformToStart, exact 1-NN (5 m gate) in the previous cloud
//                  through a uniform grid (3x3x3 block of cells with all lanes striding its nine runs, then — rarely — expanding
//                  shells; replaces the kd-tree of :203-204 / :662-663), then the +-2.5-ring windows as a second, predicated
//                  search of the same cells (LOAMX_ODOM_SCAN: the reference's point-by-point window loops instead)
//                  (a one-thread-per-feature variant was built and measured: 2.6 ms per launch against ~50 us — every lane streaming
//                  through its own cells defeats the memory pipeline; the wave-cooperative, coalesced walk stays)
//   k_odom_lm      (:304-361, :437-481, :497-622)  one PERSISTENT workgroup per stream runs the 5 iterations without a
//                  kernel boundary: thread per feature point-to-line / point-to-plane coefficients and Jacobian row with
//                  the de-skew chain rule, J^T J / J^T r reduced with wave shuffles (double accumulators), thread 0:
//                  6x6 pivoted QR solve, degeneracy projector, update, convergence test.
#include "odometry.hpp"
#include "pinned_copy.hpp"
#include <chrono>
#include <cstring>
#include <type_traits>

namespace loamx {

constexpr int OD_THREADS = 256;
constexpr int OD_TR_STRIDE = OD_THREADS + 2;        // k_odom_lm LDS transpose: row stride in doubles (bank spread)
constexpr int OD_PART_STRIDE = 2 * (2 * 16 * LX_NSUM) + 16;   // per stream, in doubles: 2 x 16 workgroups' partial sums as 16-byte tagged records (+ 16 slots for LOAMX_PROF_LM timestamps)
[[maybe_unused]] constexpr int OD_PART_TS = 2 * (2 * 16 * LX_NSUM);
// k_odom_lm: up to 16 workgroups x 256 threads x 2 features per thread kept in registers = 8192 features per sweep

// per-point de-skew angles (|a| << 1): float sincos (<= 2 ulp); the once-per-iteration pose trig stays in double
__device__ inline void sincos_f(float a, float& s, float& c) { sincosf(a, &s, &c); }

// transformToStart (:40-53)
__device__ inline void transform_to_start(const float* T, float scan_period, const float4 pi, float& x, float& y, float& z) {
  const float s = (1.f / scan_period) * (pi.w - (float)(int)pi.w);
  x = pi.x - s * T[3];
  y = pi.y - s * T[4];
  z = pi.z - s * T[5];
  float sx, cx, sy, cy, sz, cz;
  sincos_f(-s * T[0], sx, cx);
  sincos_f(-s * T[1], sy, cy);
  sincos_f(-s * T[2], sz, cz);
  rot_z(x, y, cz, sz);
  rot_x(y, z, cx, sx);
  rot_y(x, z, cy, sy);
}

// cell-sorted copies of the previous clouds carry (ring << 24) | index inside the cloud in .w (k_bb_scatter with pack_ring); ring 255 = unknown
constexpr uint32_t OD_IDX_MASK = 0xffffffu;

__device__ inline float sqd(const float4& a, float x, float y, float z) {
  const float dx = a.x - x, dy = a.y - y, dz = a.z - z;
  return dx * dx + dy * dy + dz * dz;
}

// wave-level arg-min of (d, order); every lane gets the winner
__device__ inline void wave_argmin(float& d, int& j, int& order) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float od = __shfl_xor(d, off, 64);
    const int oj = __shfl_xor(j, off, 64);
    const int oo = __shfl_xor(order, off, 64);
    if (od < d || (od == d && oo < order)) { d = od; j = oj; order = oo; }
  }
}

// exact nearest neighbour (d2 < 25) searched by a WHOLE WAVE.  Per shell of grid cells: every lane first fetches the
// cell range of "its" (y,z) row (all cell_start loads of the shell are in flight together), then the rows are walked
// with the lanes striding over each row's contiguous candidate run; after every shell the lane minima are combined so
// that pruning and termination stay wave-uniform.  A row is skipped when its (y,z) slab is already farther than the
// best distance, and its x-run is clipped to the cells the best-distance ball can reach (bounds shrunk by a relative
// 1e-4 so float rounding can only make the search visit MORE cells, never fewer).
// L0 / best / best_id: the shells below L0 have been searched already (nn1_wave_flat's 27-cell block) with this wave-uniform result
__device__ inline int nn1_shells(const GridDesc& g, const float4* __restrict__ sorted, const uint32_t* __restrict__ cell_start, float qx,
                                 float qy, float qz, int lane, int L0, float best, int best_id) {
  float lbest = best;           // this lane's best (best: wave-uniform bound; only candidates with d2 < 25 are admissible)
  uint32_t lid = best_id >= 0 ? (uint32_t)best_id : 0xffffffffu;
  const float h = 1.0f / g.inv_h;
  const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  for (int L = L0;; L++) {
    const int side = 2 * L + 1, nrows = side * side;
    for (int r0 = 0; r0 < nrows; r0 += 64) {
      // ---- lane r: ranges of row r0 + lane
      uint32_t b0 = 0, e0 = 0, b1 = 0, e1 = 0;
      {
        const int rr = r0 + lane;
        if (rr < nrows) {
          const int dz = rr / side - L, dy = rr % side - L;
          const int z = cz + dz, y = cy + dy;
          if (z >= 0 && z < g.nz && y >= 0 && y < g.ny) {
            const float gz = dz == 0 ? 0.f : (dz > 0 ? ((float)z - fz) : (fz - (float)(z + 1))) * h;
            const float gy = dy == 0 ? 0.f : (dy > 0 ? ((float)y - fy) : (fy - (float)(y + 1))) * h;
            const float gyz = (gy * gy + gz * gz) * 0.9999f;
            if (gyz < best) {
              const float rx = sqrtf(best - gyz) * g.inv_h * 1.0001f + 1e-3f;
              const int xlo = (int)floorf(fx - rx), xhi = (int)floorf(fx + rx);
              const bool face = (dz == -L || dz == L || dy == -L || dy == L);
              const uint32_t row = ((uint32_t)z * g.ny + y) * g.nx;
              // on a face row the whole x-run [cx-L, cx+L] is new; otherwise only its two end cells
              int xa = cx - L, xb = face ? cx + L : cx - L;
              if (xa < xlo) xa = xlo;
              if (xb > xhi) xb = xhi;
              if (xa < 0) xa = 0;
              if (xb > g.nx - 1) xb = g.nx - 1;
              if (xa <= xb) { b0 = cell_start[row + xa]; e0 = cell_start[row + xb + 1]; }
              if (!face && L > 0) {
                const int xc = cx + L;
                if (xc >= xlo && xc <= xhi && xc >= 0 && xc <= g.nx - 1) { b1 = cell_start[row + xc]; e1 = cell_start[row + xc + 1]; }
              }
            }
          }
        }
      }
      // ---- walk the non-empty runs
      unsigned long long todo = __ballot(e0 > b0 || e1 > b1);
      while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t rb0 = __shfl(b0, src, 64), re0 = __shfl(e0, src, 64), rb1 = __shfl(b1, src, 64), re1 = __shfl(e1, src, 64);
        for (int part = 0; part < 2; part++) {
          const uint32_t beg = part ? rb1 : rb0, end = part ? re1 : re0;
          for (uint32_t k = beg + lane; k < end; k += 64) {
            const float4 p = sorted[k];
            const float dx = qx - p.x, dy2 = qy - p.y, dz2 = qz - p.z;
            const float d2 = dx * dx + dy2 * dy2 + dz2 * dz2;
            const uint32_t id = __float_as_uint(p.w) & OD_IDX_MASK;   // (the top byte carries the point's ring: SubMapIndexBatch::pack_ring)
            if (d2 < lbest || (d2 == lbest && lid != 0xffffffffu && id < lid)) { lbest = d2; lid = id; }
          }
        }
      }
    }
    // combine: smallest distance, ties to the lowest original index
    float d = lbest;
    int j = (int)lid, o_ = (int)lid;   // order key = original index (< 2^31)
    if (lid == 0xffffffffu) { j = -1; o_ = 0x7fffffff; }
    wave_argmin(d, j, o_);
    if (j >= 0 && d < 25.0f) { best = d; best_id = j; }
    const float cover = (float)L * h;
    if (best <= cover * cover) break;
    if (cover * cover >= 25.0f) break;
  }
  return best_id;
}
// ---- the 3x3x3 block of cells around a query, enumerated flat (round 4).  nn1_wave walks the rows of a shell one after the other, a
// load latency each — ~10 of them before the 2.1 m shell is done.  Here: trip 1, the 18 boundaries of the block's 9 rows (a row's
// three cells are one contiguous run of the cell-sorted array), one per lane; then the block's candidates numbered 0 .. total-1 across
// the runs and dealt over the 64 lanes, 4 independent 16-byte loads in flight per lane.  The block contains the ball of radius h
// around the query: a minimum below h^2 found in it is the minimum over the whole cloud.
#ifdef OD_CORR_WAVES
#define OD_CORR_ATTR __attribute__((amdgpu_waves_per_eu(OD_CORR_WAVES, OD_CORR_WAVES)))
#else
#define OD_CORR_ATTR
#endif
struct Block27 {
  uint32_t E[9], O[9];   // cumulative candidate count after run r; position of candidate cc of run r = cc + O[r]
  uint32_t total;
};
__device__ inline void block27_build(const GridDesc& g, const uint32_t* __restrict__ cell_start, float qx, float qy, float qz, int lane, Block27& B) {
  const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  uint32_t v = 0u;
  {
    const int r = lane >> 1, z = cz + r / 3 - 1, y = cy + r % 3 - 1;
    if (lane < 18 && z >= 0 && z < g.nz && y >= 0 && y < g.ny) {
      int xx = (lane & 1) ? cx + 2 : cx - 1;   // run [cx - 1, cx + 1] = cell_start[cx - 1] .. cell_start[cx + 2], clamped onto the row
      xx = xx < 0 ? 0 : (xx > g.nx ? g.nx : xx);
      v = cell_start[((uint32_t)z * g.ny + y) * g.nx + xx];
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const uint32_t b = (uint32_t)__shfl((int)v, 2 * r, 64), e = (uint32_t)__shfl((int)v, 2 * r + 1, 64);
    B.O[r] = b - acc;
    acc += e > b ? e - b : 0u;
    B.E[r] = acc;
  }
  B.total = acc;
}
// visit(point, valid): called for 4 candidates per lane and trip, every lane the same number of times (valid = false past the end).
// The FIRST trip's first OD_BLK_CACHE points (candidates lane, lane + 64 — the whole block when it holds <= 128 points, the
// usual case) are handed in and out through `first`: the nearest-neighbour pass loads them, the ring-window pass that follows takes them
// from registers instead of going to the cache again (one dependent round trip less on every feature's chain).
#ifndef OD_BLK_CACHE
#define OD_BLK_CACHE 2   // points per lane kept from the first trip (0: none — every pass loads): 2 = blocks of <= 128 points, 8 VGPRs; with 4 the kernel drops from 6 to 5 waves per SIMD
#endif
struct Block27First { float4 p[OD_BLK_CACHE > 0 ? OD_BLK_CACHE : 1]; };
template <bool LOAD_FIRST, class F>
__device__ inline void block27_scan(const Block27& B, const float4* __restrict__ sorted, int lane, Block27First& first, F&& visit) {
  for (uint32_t c0 = (uint32_t)lane; c0 < B.total; c0 += 4 * 64) {
    const bool first_trip = c0 == (uint32_t)lane;
    uint32_t pos[4];
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t cc = c0 + 64u * u;
      uint32_t o = B.O[8];
#pragma unroll
      for (int k = 7; k >= 0; k--) o = cc < B.E[k] ? B.O[k] : o;
      pos[u] = cc < B.total ? cc + o : B.O[0];   // (O[0] = the first run's first slot: a valid address whenever total > 0)
    }
    if (LOAD_FIRST || !first_trip) {
#pragma unroll
      for (int u = 0; u < 4; u++) p[u] = sorted[pos[u]];
      if (LOAD_FIRST && first_trip) {
#pragma unroll
        for (int u = 0; u < OD_BLK_CACHE; u++) first.p[u] = p[u];
      }
    } else {
#pragma unroll
      for (int u = OD_BLK_CACHE; u < 4; u++) p[u] = sorted[pos[u]];
#pragma unroll
      for (int u = 0; u < OD_BLK_CACHE; u++) p[u] = first.p[u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) visit(p[u], (c0 + 64u * u) < B.total);
  }
}
// exact nearest neighbour (d2 < 25) = nn1_wave's result: the block first, and only when nothing lies within h of the query (rare) the
// shells from 2 on.  w_out: the winner's packed .w (ring << 24 | index)
__device__ inline int nn1_block(const GridDesc& g, const Block27& B, const float4* __restrict__ sorted, const uint32_t* __restrict__ cell_start,
                                float qx, float qy, float qz, int lane, uint32_t& w_out, Block27First& first) {
  float lbest = 25.0f;
  uint32_t lw = 0xffffffffu;   // packed .w of this lane's best
  block27_scan<true>(B, sorted, lane, first, [&](const float4& p, bool valid) {
    const float dx = qx - p.x, dy2 = qy - p.y, dz2 = qz - p.z;
    const float d2 = dx * dx + dy2 * dy2 + dz2 * dz2;
    const uint32_t w = __float_as_uint(p.w);
    if (valid && (d2 < lbest || (d2 == lbest && lw != 0xffffffffu && (w & OD_IDX_MASK) < (lw & OD_IDX_MASK)))) { lbest = d2; lw = w; }
  });
  float d = lbest;
  int j = lw == 0xffffffffu ? -1 : (int)(lw & OD_IDX_MASK), o_ = lw == 0xffffffffu ? 0x7fffffff : (int)(lw & OD_IDX_MASK);
  int ring = (int)(lw >> 24);
  {   // wave arg-min of (d, index) carrying the ring along
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_xor(d, off, 64);
      const int oj = __shfl_xor(j, off, 64), oo = __shfl_xor(o_, off, 64), orr = __shfl_xor(ring, off, 64);
      if (od < d || (od == d && oo < o_)) { d = od; j = oj; o_ = oo; ring = orr; }
    }
  }
  const float h = 1.0f / g.inv_h;
  if (j >= 0 && d < 25.0f && d <= h * h) { w_out = ((uint32_t)ring << 24) | (uint32_t)j; return j; }
  const int id = nn1_shells(g, sorted, cell_start, qx, qy, qz, lane, 2, (j >= 0 && d < 25.0f) ? d : 25.0f, (j >= 0 && d < 25.0f) ? j : -1);
  w_out = id == j ? (((uint32_t)ring << 24) | (uint32_t)j) : (0xff000000u | (uint32_t)(id < 0 ? 0 : id));   // (a shell winner's ring: unknown here)
  return id;
}

// ---- correspondences, round 4: k_odom_corr's structure (one wave per feature, no LDS, XCD-aware feature order) with
//   * the nearest neighbour from the flat 27-cell block (nn1_block: 3 dependent trips instead of ~10), its ring from the packed .w
//     (no dependent load of last[closest].w), and
//   * the +-2.5-ring windows searched THROUGH THE SAME BLOCK when that is the shorter way.  The reference walks every point of up to
//     five rings (:262-297 / :378-429); the upper rings of an HDL-64E less-flat cloud hold ~2,000 points each (far walls: the 0.2 m
//     voxel filter merges nothing there), so their features scanned ~6,000 points each and were the kernel's long pole (24 us against
//     12 for a mid ring, in-kernel time stamps).  For a ring-ORDERED previous cloud — verified point by point by the re-projection
//     kernel, not assumed — the walk's candidate set is {index < closest, ring >= cscan - 2} u {closest < index < bound, ring <= cscan + 2},
//     so the nearest qualifying point can be looked up in the grid instead: the block's candidates (already in cache from the
//     nearest-neighbour pass) are filtered by ring and index and ranked by (distance, scan order) exactly like the walk.  A slot's block
//     minimum is final when it lies strictly inside the ball the block contains (d2 < h^2); otherwise — sparse ground rings 20-50 m out,
//     whose neighbours are metres apart — the feature takes the walk, which is short exactly there.  The choice (walk length from the
//     ring-first table against the block's candidate count) only decides speed; both ways give the same ind[] triple.
// Measured and dropped on the way (profiles/r04_corr.md): the windows staged in LDS by 32- and 8-feature workgroups (5x less L2
// traffic, no faster: the searches are latency chains, not bandwidth).
constexpr int OD_RF_N = 320;           // ring-first table entries per cloud: rings 0..318 (ring ids < 256: loamx.h); entry 319 = epoch of the last UNORDERED version of the cloud
#ifdef LOAMX_PROF_CORR
__device__ unsigned long long g_corr_ts[3][16];   // [corner feature / flat mid / last flat][stamp]
#define OC_TS(k) do { if (blockIdx.y == 0 && lane == 0 && ts_slot >= 0) g_corr_ts[ts_slot][k] = wall_clock64(); } while (0)
#else
#define OC_TS(k) do { } while (0)
#endif
// patch_dst / patch (optional): three words this launch writes for the kernels BEHIND it on the stream — the cloud offsets of a
// single-stream sweep whose less-flat cloud arrived late (OdometryBatch::process, resolve_late): no copy command between two launches
__global__ __launch_bounds__(256) OD_CORR_ATTR void k_odom_corr_grid(OdomProblem* __restrict__ probs, OdomParams P, uint32_t* __restrict__ patch_dst, uint4 patch) {
  if (patch_dst && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { patch_dst[0] = patch.x; patch_dst[1] = patch.y; patch_dst[2] = patch.z; }
  OdomProblem& pb = probs[blockIdx.y];
  if (pb.done) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nSharp = (int)pb.n_sharp, nFlat = (int)pb.n_flat;
  int f;
  {   // XCD-aware order as in k_odom_corr: XCD x takes the x-th eighth of every stream's sharp and flat lists
    const int x = (int)(blockIdx.x % 8), j = (int)(blockIdx.x / 8);
    const int sS = (int)((long long)x * nSharp / 8), nS = (int)((long long)(x + 1) * nSharp / 8) - sS;
    const int sF = (int)((long long)x * nFlat / 8), nF = (int)((long long)(x + 1) * nFlat / 8) - sF;
    const int fl = 4 * j + wid;
    if (fl >= nS + nF) return;
    f = fl < nS ? sS + fl : nSharp + sF + (fl - nS);
  }
#ifdef LOAMX_PROF_CORR
  const int ts_slot = f == nSharp / 2 ? 0 : f == nSharp + nFlat / 2 ? 1 : f == nSharp + nFlat - 1 ? 2 : -1;
#endif
  OC_TS(0);
  float T[6];
#pragma unroll
  for (int k = 0; k < 6; k++) T[k] = pb.transform[k];
  const bool corner = f < nSharp;
  const float4 pi = corner ? pb.sharp[f] : pb.flat[f - nSharp];
  const uint32_t* __restrict__ tbl = corner ? pb.rf_corner : pb.rf_surf;
  // the ring-first entries around the feature's own ring (one per lane: rings rf - 8 .. rf + 55), asked for before the search needs them
  const int e0 = max((int)pi.w - 8, 0);
  uint32_t tv = 0u, unordered_epoch = 0u;
  if (tbl) {
    if (e0 + lane < OD_RF_N - 1) tv = tbl[e0 + lane];
    unordered_epoch = tbl[OD_RF_N - 1];
  }
  float x, y, z;
  transform_to_start(T, P.scan_period, pi, x, y, z);
  const GridDescB gd = corner ? *pb.lc_desc : *pb.ls_desc;
  const float4* __restrict__ sorted = pb.sorted;
  const uint32_t* __restrict__ cell_start = pb.cell_table + gd.cell_base;
  OC_TS(1);
  Block27 B;
  block27_build(gd.g, cell_start, x, y, z, lane, B);
  OC_TS(2);
  uint32_t wbest = 0u;
  Block27First first;
  const int closest = nn1_block(gd.g, B, sorted, cell_start, x, y, z, lane, wbest, first);
  OC_TS(3);
  if (closest < 0) {   // wave-uniform
    if (lane == 0) { pb.ind[5 * f] = -1; pb.ind[5 * f + 1] = -1; pb.ind[5 * f + 2] = -1; }
    return;
  }
  const float4* last = corner ? pb.last_corner : pb.last_surf;
  const int nLast = corner ? (int)pb.n_last_corner : (int)pb.n_last_surf;
  const int nCur = corner ? nSharp : nFlat;
  const int bound = nCur < nLast ? nCur : nLast;   // forward scans are bounded by the CURRENT feature count (:262, :378)
  const int cscan = (wbest >> 24) != 255u ? (int)(wbest >> 24) : (int)last[closest].w;
  float d2 = 25.f, d3 = 25.f;
  int j2 = -1, j3 = -1, o2 = 0x7fffffff, o3 = 0x7fffffff;
  auto better = [](float d, int order, float dbest, int obest) { return d < dbest || (d == dbest && obest != 0x7fffffff && order < obest); };   // scan order decides ties (never admits d == 25)

  // ---- through the block?  Only for a cloud verified ring-ordered with all its rings < 255, and only when the walk would be longer
  bool via_block = false;
  if (tbl && unordered_epoch != pb.rf_epoch && cscan < 255 && B.total > 0u) {
    const unsigned long long valid = __ballot(e0 + lane < OD_RF_N - 1 && (tv >> 24) == pb.rf_epoch);
    auto first_from = [&](int ring) -> int {   // first index of the first ring >= `ring` with an entry (nLast: none among the 64 read)
      const int k = ring - e0;
      if (k >= 64) return nLast;
      const unsigned long long m = k <= 0 ? valid : (valid >> k) << k;
      if (!m) return nLast;
      const int w = (int)((uint32_t)__shfl((int)tv, __builtin_ctzll(m), 64) & OD_IDX_MASK);
      return w < nLast ? w : nLast;
    };
    if (cscan - 2 >= e0 || e0 == 0) {   // (the entries read cover the walk's rings)
      const int wlo = min(first_from(cscan - 2), closest);
      const int whi = closest + 1 < bound ? max(min(bound, first_from(cscan + 3)), closest + 1) : closest + 1;
      const unsigned walk = (unsigned)(closest - wlo) + (unsigned)(whi - (closest + 1));
      via_block = walk > B.total + 128u;
    }
  }
  OC_TS(4);
  if (via_block) {
    block27_scan<false>(B, sorted, lane, first, [&](const float4& p, bool ok) {
      const uint32_t w = __float_as_uint(p.w);
      const int j = (int)(w & OD_IDX_MASK), ring = (int)(w >> 24);
      const float d = sqd(p, x, y, z);
      if (ok && j > closest && j < bound && ring <= cscan + 2) {          // what the forward walk would examine
        const int order = j - (closest + 1);
        if (corner) {
          if (ring > cscan && better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; }
        } else {
          if (ring <= cscan) { if (better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; } }
          else { if (better(d, order, d3, o3)) { d3 = d; j3 = j; o3 = order; } }
        }
      } else if (ok && j < closest && ring >= cscan - 2) {                 // ... and the backward walk
        const int order = 0x40000000 + (closest - 1 - j);
        if (corner) {
          if (ring < cscan && better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; }
        } else {
          if (ring >= cscan) { if (better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; } }
          else { if (better(d, order, d3, o3)) { d3 = d; j3 = j; o3 = order; } }
        }
      }
    });
    wave_argmin(d2, j2, o2);
    if (!corner) wave_argmin(d3, j3, o3);
    const float h = 1.0f / gd.g.inv_h, h2 = h * h;
    // final only strictly inside the ball the block contains; a slot without such a candidate sends the feature on the walk
    via_block = d2 < h2 && (corner || d3 < h2);
    if (!via_block) { d2 = d3 = 25.f; j2 = j3 = -1; o2 = o3 = 0x7fffffff; }
  }
  OC_TS(5);
  if (!via_block) {
    // forward window (:262-279 / :378-403) and backward window (:280-297 / :404-429), walked TOGETHER as in k_odom_corr
    int baseF = closest + 1, baseB = closest - 1;
    bool stopF = baseF >= bound, stopB = baseB < 0;
    while (!stopF || !stopB) {
      float4 qf[4], qb[4];
      if (!stopF) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = baseF + 64 * u + lane;
          qf[u] = j < bound ? last[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      if (!stopB) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int j = baseB - 64 * u - lane;
          qb[u] = j >= 0 ? last[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      if (!stopF) {
        bool stop = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (stop) continue;
          const int j = baseF + 64 * u + lane;
          const bool in = j < bound;
          const int ring = (int)qf[u].w;
          const bool brk = in && ((double)ring > (double)cscan + 2.5);
          const unsigned long long mb = __ballot(brk);
          const int fb = mb ? __builtin_ctzll(mb) : 64;
          if (in && lane < fb) {
            const float d = sqd(qf[u], x, y, z);
            const int order = j - (closest + 1);
            if (corner) {
              if (ring > cscan && better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; }
            } else {
              if (ring <= cscan) { if (better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; } }
              else { if (better(d, order, d3, o3)) { d3 = d; j3 = j; o3 = order; } }
            }
          }
          if (mb) stop = true;
        }
        baseF += 256;
        stopF = stop || baseF >= bound;
      }
      if (!stopB) {
        bool stop = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (stop) continue;
          const int j = baseB - 64 * u - lane;
          const bool in = j >= 0;
          const int ring = (int)qb[u].w;
          const bool brk = in && ((double)ring < (double)cscan - 2.5);
          const unsigned long long mb = __ballot(brk);
          const int fb = mb ? __builtin_ctzll(mb) : 64;
          if (in && lane < fb) {
            const float d = sqd(qb[u], x, y, z);
            const int order = 0x40000000 + (closest - 1 - j);   // backward candidates come after all forward ones
            if (corner) {
              if (ring < cscan && better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; }
            } else {
              if (ring >= cscan) { if (better(d, order, d2, o2)) { d2 = d; j2 = j; o2 = order; } }
              else { if (better(d, order, d3, o3)) { d3 = d; j3 = j; o3 = order; } }
            }
          }
          if (mb) stop = true;
        }
        baseB -= 256;
        stopB = stop || baseB < 0;
      }
    }
    wave_argmin(d2, j2, o2);
    if (!corner) wave_argmin(d3, j3, o3);
  }
  OC_TS(6);
  if (lane == 0) {
    pb.ind[5 * f] = closest;
    pb.ind[5 * f + 1] = j2;
    pb.ind[5 * f + 2] = corner ? -1 : j3;
  }
#ifdef LOAMX_PROF_CORR
  if (blockIdx.y == 0 && lane == 0 && ts_slot >= 0) {
    g_corr_ts[ts_slot][7] = wall_clock64();
    g_corr_ts[ts_slot][8] = (unsigned long long)B.total; g_corr_ts[ts_slot][9] = (unsigned long long)via_block;
    g_corr_ts[ts_slot][10] = (unsigned long long)closest; g_corr_ts[ts_slot][11] = (unsigned long long)cscan;
  }
#endif
}

// ---- phase C: iterations [iter0, iter0 + n_iters) of one stream in gridDim.x PERSISTENT workgroups (grid = NB x streams).
// The features of a stream are dealt out over its NB workgroups (one CU cannot evaluate 2304 rows in less than ~16 us,
// nine can).  Per iteration every workgroup reduces its rows to 28 double sums, publishes them (double-buffered by
// iteration parity) and counts itself in on the stream's arrival counter; once all NB have arrived EVERY workgroup adds
// the NB partial sums in workgroup order and solves — the same deterministic arithmetic everywhere, so all of them hold
// the same new pose without a second exchange.  Workgroup 0 records the results.  The spin needs every workgroup of the
// launch resident: the host cuts a batch into launches of at most half the device's occupancy-derived capacity
// (OdometryBatch::process).
#ifdef LOAMX_PROF_LM
#define LM_TS(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && iter == it_begin + 1) pb.part[OD_PART_TS + (k)] = (double)wall_clock64(); } while (0)
#else
#define LM_TS(k) do { } while (0)
#endif
// 3 waves per SIMD (<= 168 VGPRs) and 36 KB of LDS: the persistent workgroups of a stream have to find room next to the wide
// registration / feature kernels of the other HIP streams — at 226 VGPRs + 61 KB they waited for half-empty CUs.  Round 6, one box,
// three interleaved rounds (profiles/r06_ab.md): 128 VGPRs / 41 spilled (160 B of scratch per thread, 1.2 MB of scratch writes per
// launch in the counters) 16,828 / 16,805 / 16,725 sweeps/s; 168 VGPRs / 5 spilled (24 B) 16,576 / 16,708 / 17,002; 230 VGPRs / none
// (2 waves) 16,658 / 16,540 / 16,511 — the middle one: a sixth of the scratch at the speed of the first.
#ifndef OD_LM_WAVES
#define OD_LM_WAVES 3
#endif
#ifndef OD_LM_WAVES_MIN
#define OD_LM_WAVES_MIN 3
#endif
#define OD_LM_ATTR __attribute__((amdgpu_waves_per_eu(OD_LM_WAVES_MIN, OD_LM_WAVES)))
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"   // (the two-features-per-thread variant keeps 64 KB of LDS and cannot reach that occupancy)
template <int OD_FPT>
__global__ __launch_bounds__(OD_THREADS) OD_LM_ATTR void k_odom_lm(OdomProblem* __restrict__ probs, OdomParams P, int iter0, int n_iters) {
  OdomProblem& pb = probs[blockIdx.y];
  if (pb.done) return;
  const unsigned NB = gridDim.x;
  const int tid = threadIdx.x;
  const int nSharp = (int)pb.n_sharp, nFlat = (int)pb.n_flat, nFeat = nSharp + nFlat;
  __shared__ float T[6];
  __shared__ float trig[6];
  __shared__ double red[8][LX_NSUM];
  // one feature per thread (the usual case): the 28 products of a row are formed in float and go to LDS as floats — half the
  // LDS and no 56-register double accumulator; the column sums convert to double exactly as before ((double)product, same order).
  // Two features per thread: their two products are added in double first, so the table stays double.
  using TrT = typename std::conditional<OD_FPT == 1, float, double>::type;
  __shared__ TrT tr[LX_NSUM * OD_TR_STRIDE];
  __shared__ int sh_done, sh_degen, sh_abort;
  __shared__ float matP[36];
  __shared__ float ws[216];
  __shared__ double parts[16 * LX_NSUM];   // the partial sums of the stream's (<= 16) workgroups
  __shared__ float AtA[36], AtB[6];
  if (tid < 6) {
    T[tid] = pb.transform[tid];
    // sin/cos of the three angles, one per lane, double then rounded (see pose_set_angles); later iterations get theirs from the update step
    const double ang = (double)pb.transform[tid >> 1];
    trig[tid] = (float)((tid & 1) ? cos(ang) : sin(ang));
  }
  const unsigned xtag = pb.xchg_epoch << 8;   // this sweep's number in the upper 24 bits of every record's tags
  if (tid == 0) { sh_done = 0; sh_abort = 0; sh_degen = pb.stats.degenerate; }
  if (tid < 36) matP[tid] = pb.matP[tid];   // set at iteration 0 (an earlier launch when iter0 > 0)
  __syncthreads();

  // the correspondences are fixed for the iterations of this launch: keep each thread's features (raw point + tripod
  // points) in registers instead of re-gathering them from HBM every iteration
  float4 fpo[OD_FPT], ft1[OD_FPT], ft2[OD_FPT], ft3[OD_FPT];
  bool fvalid[OD_FPT], fcorner[OD_FPT];
#pragma unroll
  for (int u = 0; u < OD_FPT; u++) {
    const int f = (u * (int)NB + (int)blockIdx.x) * OD_THREADS + tid;
    fvalid[u] = false;
    fcorner[u] = f < nSharp;
    fpo[u] = ft1[u] = ft2[u] = ft3[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < nFeat) {
      const int i1 = pb.ind[5 * f], i2 = pb.ind[5 * f + 1], i3 = pb.ind[5 * f + 2];
      if (fcorner[u] ? (i2 >= 0) : (i2 >= 0 && i3 >= 0)) {
        fvalid[u] = true;
        fpo[u] = fcorner[u] ? pb.sharp[f] : pb.flat[f - nSharp];
        const float4* last = fcorner[u] ? pb.last_corner : pb.last_surf;
        ft1[u] = last[i1];
        ft2[u] = last[i2];
        if (!fcorner[u]) ft3[u] = last[i3];
      }
    }
  }

  const int it_begin = iter0, it_end = iter0 + n_iters;
  for (int iter = it_begin; iter < it_end; iter++) {
    // ---- phase C: residual rows + normal equations
    LM_TS(0);
    LM_TS(1);
    double v[OD_FPT == 1 ? 1 : LX_NSUM];
    float a1[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bb1 = 0.f;   // OD_FPT == 1: the row of this thread's feature (zeros when not selected)
    bool sel1 = false;
#pragma unroll
    for (int k = 0; k < (OD_FPT == 1 ? 1 : LX_NSUM); k++) v[k] = 0.0;
#pragma unroll
    for (int u = 0; u < OD_FPT; u++) {
      if (!fvalid[u]) continue;
      const bool corner = fcorner[u];
      const float4 po = fpo[u];
      float cx = 0.f, cy = 0.f, cz = 0.f, ci = 0.f;
      bool sel = false;
      {
        float x0, y0, z0;
        transform_to_start(T, P.scan_period, po, x0, y0, z0);
        if (corner) {
          const float4 t1 = ft1[u], t2 = ft2[u];
          const float x1 = t1.x, y1 = t1.y, z1 = t1.z, x2 = t2.x, y2 = t2.y, z2 = t2.z;
          const float a012 = sqrtf(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                                   ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                                   ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
          const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
          const float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                            (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) / a012 / l12;
          const float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) -
                             (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
          const float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                             (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
          const float ld2 = a012 / l12;
          float s = 1;
          if (iter >= 5) s = 1 - 1.8f * fabsf(ld2);
          cx = s * la; cy = s * lb; cz = s * lc; ci = s * ld2;
          sel = ((double)s > 0.1) && (ld2 != 0);
        } else {
          const float4 t1 = ft1[u], t2 = ft2[u], t3 = ft3[u];
          float pa = (t2.y - t1.y) * (t3.z - t1.z) - (t3.y - t1.y) * (t2.z - t1.z);
          float pbb = (t2.z - t1.z) * (t3.x - t1.x) - (t3.z - t1.z) * (t2.x - t1.x);
          float pc = (t2.x - t1.x) * (t3.y - t1.y) - (t3.x - t1.x) * (t2.y - t1.y);
          float pd = -(pa * t1.x + pbb * t1.y + pc * t1.z);
          const float ps = sqrtf(pa * pa + pbb * pbb + pc * pc);
          pa /= ps; pbb /= ps; pc /= ps; pd /= ps;
          const float pd2 = pa * x0 + pbb * y0 + pc * z0 + pd;
          float s = 1;
          if (iter >= 5) s = 1 - 1.8f * fabsf(pd2) / sqrtf(sqrtf(x0 * x0 + y0 * y0 + z0 * z0));
          cx = s * pa; cy = s * pbb; cz = s * pc; ci = s * pd2;
          sel = ((double)s > 0.1) && (pd2 != 0);
        }
      }
      if (sel) {
        // Jacobian row (:502-553) with s = 1; pointOri is the RAW (not de-skewed) point (:358, :478)
        const float srx = trig[0], crx = trig[1], sry = trig[2], cry = trig[3], srz = trig[4], crz = trig[5];
        const float tx = T[3], ty = T[4], tz = T[5];
        const float px = po.x, py = po.y, pz = po.z;
        float a[6];
        a[0] = (-crx * sry * srz * px + crx * crz * sry * py + srx * sry * pz + tx * crx * sry * srz - ty * crx * crz * sry - tz * srx * sry) * cx +
               (srx * srz * px - crz * srx * py + crx * pz + ty * crz * srx - tz * crx - tx * srx * srz) * cy +
               (crx * cry * srz * px - crx * cry * crz * py - cry * srx * pz + tz * cry * srx + ty * crx * cry * crz - tx * crx * cry * srz) * cz;
        a[1] = ((-crz * sry - cry * srx * srz) * px + (cry * crz * srx - sry * srz) * py - crx * cry * pz + tx * (crz * sry + cry * srx * srz) +
                ty * (sry * srz - cry * crz * srx) + tz * crx * cry) * cx +
               ((cry * crz - srx * sry * srz) * px + (cry * srz + crz * srx * sry) * py - crx * sry * pz + tz * crx * sry -
                ty * (cry * srz + crz * srx * sry) - tx * (cry * crz - srx * sry * srz)) * cz;
        a[2] = ((-cry * srz - crz * srx * sry) * px + (cry * crz - srx * sry * srz) * py + tx * (cry * srz + crz * srx * sry) -
                ty * (cry * crz - srx * sry * srz)) * cx +
               (-crx * crz * px - crx * srz * py + ty * crx * srz + tx * crx * crz) * cy +
               ((cry * crz * srx - sry * srz) * px + (crz * sry + cry * srx * srz) * py + tx * (sry * srz - cry * crz * srx) -
                ty * (crz * sry + cry * srx * srz)) * cz;
        a[3] = -(cry * crz - srx * sry * srz) * cx + crx * srz * cy - (crz * sry + cry * srx * srz) * cz;
        a[4] = -(cry * srz + crz * srx * sry) * cx - crx * crz * cy - (sry * srz - cry * crz * srx) * cz;
        a[5] = crx * sry * cx - srx * cy - crx * cry * cz;
        const float bb = (float)(-0.05 * (double)ci);
        if (OD_FPT == 1) {
#pragma unroll
          for (int i = 0; i < 6; i++) a1[i] = a[i];
          bb1 = bb;
          sel1 = true;
        } else {
          int k = 0;
#pragma unroll
          for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) v[k++] += (double)(a[i] * a[j]);
#pragma unroll
          for (int i = 0; i < 6; i++) v[k++] += (double)(a[i] * bb);
          v[k] += 1.0;
        }
      }
    }
    LM_TS(2);
    // transposed reduction through LDS (28 dependent 64-bit shuffle chains cost ~6 us; this is < 1 us): column c of the
    // 256 x 28 table is summed by 8 threads (rows g, g+8, ...), then the 8 strands in order — a fixed order, so the
    // sums are deterministic
    if (OD_FPT == 1) {
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) tr[(k++) * OD_TR_STRIDE + tid] = (TrT)(a1[i] * a1[j]);
#pragma unroll
      for (int i = 0; i < 6; i++) tr[(k++) * OD_TR_STRIDE + tid] = (TrT)(a1[i] * bb1);
      tr[k * OD_TR_STRIDE + tid] = (TrT)(sel1 ? 1.f : 0.f);
    } else {
#pragma unroll
      for (int t = 0; t < LX_NSUM; t++) tr[t * OD_TR_STRIDE + tid] = (TrT)v[OD_FPT == 1 ? 0 : t];
    }
    __syncthreads();
    if (tid < 8 * LX_NSUM) {
      const int c = tid % LX_NSUM, g = tid / LX_NSUM;
      const TrT* col = tr + c * OD_TR_STRIDE + g;
      double x = 0.0;
#pragma unroll 8
      for (int j = 0; j < OD_THREADS / 8; j++) x += (double)col[8 * j];
      red[g][c] = x;
    }
    __syncthreads();
    LM_TS(3);
    double xs = 0.0;   // thread t < 28: sum t of the stream's normal equations (this workgroup's share first)
    if (tid < LX_NSUM) {
#pragma unroll
      for (int w = 0; w < 8; w++) xs += red[w][tid];
      if (NB > 1) {
        // one tagged 16-byte record per sum, fire and forget: no wait for the store, no arrival counter (dev_math.hpp: xrec_store)
        xrec_store(reinterpret_cast<xrec_t*>(pb.part) + (((unsigned)iter & 1u) * 16u + blockIdx.x) * LX_NSUM + tid, xs, xtag | (unsigned)(iter + 1));
      }
    }
    if (NB > 1) {
      LM_TS(4);
      // Every thread polls the records it is going to add up — all NB workgroups' sums of this iteration (the buffer alternates with the
      // iteration's parity: a workgroup overwrites a record only after every workgroup has published the NEXT iteration, i.e. has read
      // this one).  A record counts when both of its tags name this sweep and this iteration (xtag: the sweep's number — records of
      // earlier sweeps are still lying in the buffer).
      // The exchange needs every workgroup of the stream resident (the host's chunking arithmetic, OdometryBatch::process).  Should that
      // ever not hold — a device shared with another process, more handles than the arithmetic knows of — the wait gives up after
      // ~2 s of wall clock (100 MHz counter) and raises a host-visible error word instead of hanging the GPU (as VoxelPipeline's waits do)
      {
        const unsigned want = xtag | (unsigned)(iter + 1);
        const xrec_t* rec = reinterpret_cast<const xrec_t*>(pb.part) + ((unsigned)iter & 1u) * 16u * LX_NSUM;
        const unsigned long long t_in = wall_clock64();
        for (unsigned e = tid; e < NB * LX_NSUM; e += OD_THREADS) {
          unsigned spins = 0;
          xrec_t r = xrec_load(rec + e);
          while (r.x != want || r.w != want) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0u && wall_clock64() - t_in > 200000000ull) { sh_abort = 1; break; }
            r = xrec_load(rec + e);
          }
          parts[e] = xrec_value(r);
        }
      }
      __syncthreads();
      if (sh_abort) {   // block-uniform: this stream's launch is abandoned, later launches see `done`
        if (tid == 0) {
          pb.done = 1;
          if (pb.err_word) __hip_atomic_store(pb.err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (pb.host_mirror) pb.host_mirror->done = -1;
        }
        return;
      }
      LM_TS(5);
    }
    // ---- wave 0 alone from here to the end of the iteration (no block-wide barrier in between): the NB partial sums in workgroup order
    // -> normal equations -> 6x6 pivoted QR -> update, stop test and the NEXT iteration's sin/cos.  The update (:485-488, :561-612) runs on
    // lanes 0..5, one pose component each, with the operations and their order as the reference's scalar loop has them.
    if (tid < 64) {
      if (NB > 1 && tid < LX_NSUM) {
        xs = 0.0;
        for (unsigned b = 0; b < NB; b++) xs += parts[b * LX_NSUM + tid];   // workgroup order: deterministic
      }
      // scatter straight into the symmetric 6x6 / right-hand side (sum index t -> (i, j) of the upper triangle)
      if (tid < 21) {
        int i = 0, rem = tid;
        while (rem >= 6 - i) { rem -= 6 - i; i++; }
        const int j = i + rem;
        AtA[i * 6 + j] = AtA[j * 6 + i] = (float)xs;
      } else if (tid < 27) {
        AtB[tid - 21] = (float)xs;
      }
      const int sel = (int)__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xs), 27), __builtin_amdgcn_readlane(__double2loint(xs), 27));   // wave-uniform
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (the solve's lanes read what other lanes of this wave have just written to LDS)
      LM_TS(7);
      float xl = 0.f;      // lane r < 6: X[r]
      int done_now = 0;
      if (sel >= 10) {     // :485-488
        xl = qr_solve6_lanes(AtA, AtB);   // all lanes (:559)
        LM_TS(8);
        if (iter == 0) {
          if (tid == 0) sh_degen = degeneracy_projector(AtA, 10.f, matP, ws) ? 1 : 0;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          if (blockIdx.x == 0 && tid < 36) pb.matP[tid] = matP[tid];
        }
        const int degen = sh_degen;
        if (degen) {       // X = matP * X2, row r on lane r (:591-593)
          const int r = tid < 6 ? tid : 5;
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < 6; c++) acc += matP[r * 6 + c] * lane_get(xl, c);
          xl = acc;
        }
        float x6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) x6[k] = lane_get(xl, k);
        const float d0 = (float)(x6[0] * 180.0 / M_PI), d1 = (float)(x6[1] * 180.0 / M_PI), d2 = (float)(x6[2] * 180.0 / M_PI);
        const float deltaR = (float)sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);
        const float t0 = x6[3] * 100, t1 = x6[4] * 100, t2 = x6[5] * 100;
        const float deltaT = (float)sqrt((double)t0 * t0 + (double)t1 * t1 + (double)t2 * t2);
        done_now = (deltaR < P.delta_r_abort && deltaT < P.delta_t_abort) ? 1 : 0;
      }
      // the new pose component of lane r < 6 (unchanged when the solve was skipped) and, on lanes 0..5, sin / cos of the three angles
      // for the next iteration's rows — and for the re-projection parameters, should this be the sweep's last iteration
      float tl = T[tid < 6 ? tid : 5];
      if (sel >= 10) {
        float nv = tl + xl;
        if (!isfinite(nv)) nv = 0.f;   // :606-612
        tl = nv;
      }
      const float a0 = lane_get(tl, 0), a1 = lane_get(tl, 1), a2 = lane_get(tl, 2);
      const double ang = (double)((tid >> 1) == 0 ? a0 : ((tid >> 1) == 1 ? a1 : a2));
      const float tg = (float)((tid & 1) ? cos(ang) : sin(ang));
      if (tid < 6) { T[tid] = tl; trig[tid] = tg; }
      if (tid == 0 && done_now) sh_done = 1;
      if (blockIdx.x == 0) {
        const bool final_now = done_now || iter == it_end - 1;
        if (tid < 6) pb.transform[tid] = tl;
        if (tid == 0) {
          pb.stats.iterations = iter + 1;
          pb.stats.sel = sel;
          if (iter == 0 && sel >= 10) pb.stats.degenerate = sh_degen;
          if (done_now) pb.done = 1;
        }
        // results also go to the host-visible mirror once they are final for this launch: the host then needs no copy on the stream,
        // only the event behind the last launch — and into the re-projection parameters of the sweep's tail (transformToEnd with the
        // optimised transform): the tail is enqueued right behind the last launch, no host round trip and no extra kernel
        if (final_now) {
          ToEndParams& Pe = *pb.te_out;
          if (tid < 6) {
            Pe.T[tid] = tl;
            // (the host caches sin/cos of a float angle (Angle.h); double-then-round is within an ulp of it)
            if (tid & 1) Pe.cT[tid >> 1] = tg; else Pe.sT[tid >> 1] = tg;
          }
          if (pb.host_mirror) {
            OdomProblem* hm = pb.host_mirror;
            if (tid < 6) hm->transform[tid] = tl;
            if (tid == 0) {
              OdomStats stt;
              stt.iterations = iter + 1; stt.sel = sel; stt.frame = pb.stats.frame; stt.degenerate = sh_degen;
              hm->stats = stt;
              hm->done = done_now;
            }
          }
        }
      }
      LM_TS(9);
    }
    __syncthreads();
    if (sh_done) break;
  }
}

#pragma clang diagnostic pop

// transformToEnd (:57-87) of one point
__device__ inline float4 to_end_point(float4 p, const ToEndParams& P) {
  const float s = (1.f / P.scan_period) * (p.w - (float)(int)p.w);
  float x = p.x - s * P.T[3], y = p.y - s * P.T[4], z = p.z - s * P.T[5];
  float sx, cx, sy, cy, sz, cz;
  sincos_f(-s * P.T[0], sx, cx);
  sincos_f(-s * P.T[1], sy, cy);
  sincos_f(-s * P.T[2], sz, cz);
  rot_z(x, y, cz, sz); rot_x(y, z, cx, sx); rot_y(x, z, cy, sy);                            // rotateZXY(rz, rx, ry)
  rot_y(x, z, P.cT[1], P.sT[1]); rot_x(y, z, P.cT[0], P.sT[0]); rot_z(x, y, P.cT[2], P.sT[2]);   // rotateYXZ(T)
  x += P.T[3] - P.shift[0];
  y += P.T[4] - P.shift[1];
  z += P.T[5] - P.shift[2];
  rot_z(x, y, P.c_start[2], P.s_start[2]); rot_x(y, z, P.c_start[0], P.s_start[0]); rot_y(x, z, P.c_start[1], P.s_start[1]);
  rot_y(x, z, P.c_end[1], -P.s_end[1]); rot_x(y, z, P.c_end[0], -P.s_end[0]); rot_z(x, y, P.c_end[2], -P.s_end[2]);
  return make_float4(x, y, z, (float)(int)p.w);
}
__global__ __launch_bounds__(256) void k_transform_to_end_copy(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n, ToEndParams P) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dst[i] = to_end_point(src[i], P);
}
__global__ __launch_bounds__(256) void k_transform_to_end(float4* __restrict__ pts, uint32_t n, ToEndParams P) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pts[i] = to_end_point(pts[i], P);
}
// the optimised transform of every active stream goes into its re-projection parameters on the device, so that the
// tail of the sweep (re-projection, index build) is enqueued behind the iterations without a host round trip
__global__ void k_te_patch(const OdomProblem* __restrict__ probs, uint32_t na, ToEndParams* __restrict__ te) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= na) return;
  const OdomProblem& pb = probs[a];
  ToEndParams& P = te[pb.stream_id];
  for (int k = 0; k < 6; k++) P.T[k] = pb.transform[k];
  for (int k = 0; k < 3; k++) {   // the host caches sin/cos of a float angle (Angle.h); double-then-round is within an ulp of it
    P.sT[k] = (float)sin((double)pb.transform[k]);
    P.cT[k] = (float)cos((double)pb.transform[k]);
  }
}

// all clouds of a batch: cloud k belongs to stream k % ns
// src_c / src_s: when given, point i is read from the (contiguous) source clouds instead of pts — the staging copy of the
// current clouds is fused into the re-projection
__global__ __launch_bounds__(256) void k_transform_to_end_batch(float4* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ off,
                                                                uint32_t K, uint32_t ns, const ToEndParams* __restrict__ params,
                                                                const float4* __restrict__ src_c, const float4* __restrict__ src_s,
                                                                uint32_t n_corner_all, uint32_t* __restrict__ bounds,
                                                                uint32_t* __restrict__ ring_first, uint32_t rf_epoch) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  uint32_t lo = 0;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    uint32_t hi = K;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (off[mid] <= i) lo = mid; else hi = mid;
    }
    const ToEndParams P = params[lo % ns];
    if (src_c) {
      const float4 p = i < n_corner_all ? src_c[i] : src_s[i - n_corner_all];
      q = P.enabled ? to_end_point(p, P) : p;
      pts[i] = q;
    } else {
      q = pts[i];
      if (P.enabled) { q = to_end_point(q, P); pts[i] = q; }
    }
  }
  // ring-first table of every cloud (k_odom_corr_grid sizes a feature's ring-window walk by it — a hint): a point whose ring differs
  // from its predecessor's inside the cloud is the first of its ring.  Entries carry the epoch, so the table is never cleared.  The
  // same comparison VERIFIES that the cloud is ring-ordered with ring ids that fit the packed byte (what lets k_odom_corr_grid look a
  // window up in the grid): one violation anywhere stamps the cloud's last entry with the epoch.
  if (ring_first) {
    const int ring = (int)q.w;
    int prev = __shfl_up(ring, 1, 64);
    const uint32_t plo = (uint32_t)__shfl_up((int)lo, 1, 64);
    const bool first_in_cloud = active && i == off[lo];
    if (active && !first_in_cloud && (threadIdx.x & 63) == 0) {   // the predecessor sits in another wave: read its ring from the source
      const uint32_t ip = i - 1;
      prev = (int)(src_c ? (ip < n_corner_all ? src_c[ip] : src_s[ip - n_corner_all]) : pts[ip]).w;   // (.w's integer part survives the re-projection)
    } else if (active && plo != lo) {
      prev = -1;
    }
    if (active && ring >= 0 && ring < OD_RF_N - 1 && (first_in_cloud || prev != ring))
      ring_first[(size_t)lo * OD_RF_N + ring] = (rf_epoch << 24) | ((i - off[lo]) & OD_IDX_MASK);
    if (active && (ring < 0 || ring >= 255 || (i - off[lo]) > OD_IDX_MASK || (!first_in_cloud && prev > ring)))
      ring_first[(size_t)lo * OD_RF_N + OD_RF_N - 1] = rf_epoch;
  }
  // the clouds are indexed next (SubMapIndexBatch): their bounds are gathered here, one launch and one pass over the points less
  if (bounds) cloud_bounds_update(bounds, active, lo, q.x, q.y, q.z);
}

// segment k of dst = re-projected copy of src[k] with the parameters of stream sid[k]
__global__ __launch_bounds__(256) void k_to_end_gather(float4* __restrict__ dst, uint32_t n, const uint32_t* __restrict__ off, uint32_t K,
                                                       const float4* const* __restrict__ src, const uint32_t* __restrict__ sid,
                                                       const ToEndParams* __restrict__ params) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = K;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  dst[i] = to_end_point(src[lo][i - off[lo]], params[sid[lo]]);
}

// ----------------------------------------------------------------------------------------------------------------
OdometryBatch::OdometryBatch(int device, uint32_t n_streams, hipStream_t shared_stream) : device_(device) {
  select_device(device);
  LX_REQUIRE(n_streams >= 1 && n_streams <= 1024, "n_streams must be in [1, 1024]");
  if (shared_stream) {
    st_ = shared_stream;
  } else {
    st_ = create_stream(env_priority("LOAMX_PRIO_ODOM", +1), 0, /*part=*/0);
    own_stream_ = true;
  }
  for (uint32_t s = 0; s < n_streams; s++) streams_.push_back(new OdomStream());
  // nn1_wave is exact for any cell size; coarser cells than the map index keep the cell tables (rebuilt every sweep) small
  index_.cell_size = 2.1f;
  if (const char* e = diag_env("LOAMX_ODOM_CELL")) { const float v = (float)atof(e); if (v >= 0.25f && v <= 16.f) index_.cell_size = v; }
  index_.pack_ring = true;   // k_odom_corr_grid filters a cell's points by ring
  index_.init(st_);
  h_mirror_.reserve(n_streams);
  h_err_.reserve(16);
  rf_.reserve((size_t)2 * n_streams * OD_RF_N);
  LX_HIP(hipMemsetAsync(rf_.p, 0, sizeof(uint32_t) * 2 * n_streams * OD_RF_N, st_));   // epoch 0 = no entry
  memset(h_err_.p, 0, 16 * sizeof(uint32_t));
  part_.reserve((size_t)n_streams * OD_PART_STRIDE);
  LX_HIP(hipMemset(part_.p, 0, sizeof(double) * part_.cap));   // (tag 0 names no sweep)
  {
    auto up256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_te = up256(sizeof(OdomProblem) * n_streams), o_off = o_te + up256(sizeof(ToEndParams) * n_streams);
    up_bytes_ = o_off + up256(sizeof(uint32_t) * (2 * (size_t)n_streams + 2));
    up_dev_.reserve(up_bytes_);
    up_host_.reserve(up_bytes_);
    memset(up_host_.p, 0, up_bytes_);
    prob_.p = (OdomProblem*)up_dev_.p; h_prob_.p = (OdomProblem*)up_host_.p;
    te_.p = (ToEndParams*)(up_dev_.p + o_te); h_te_.p = (ToEndParams*)(up_host_.p + o_te);
    d_cur_off_.p = (uint32_t*)(up_dev_.p + o_off); h_off_pin_.p = (uint32_t*)(up_host_.p + o_off);
  }
  h_cur_off_.assign(2 * n_streams + 1, 0);
  h_last_off_.assign(2 * n_streams + 1, 0);
}

OdometryBatch::~OdometryBatch() {
  if (ev_tail_) (void)hipEventDestroy(ev_tail_);
  if (ev_up_) (void)hipEventDestroy(ev_up_);
  if (ev_pose_) (void)hipEventDestroy(ev_pose_);
  for (auto* p : streams_) delete p;
  for (auto& c : lt_) for (auto& e : c.ev) if (e) (void)hipEventDestroy(e);
  if (own_stream_ && st_) (void)hipStreamDestroy(st_);
}

void OdometryBatch::lt_resolve_() {
  for (auto& c : lt_) {
    if (!c.pending || !c.pairs) continue;
    if (hipEventQuery(c.ev[3 * (c.pairs - 1) + 2]) != hipSuccess) { (void)hipGetLastError(); continue; }
    for (int k = 0; k < c.pairs; k++) {
      float a = 0.f, b = 0.f;
      if (hipEventElapsedTime(&a, c.ev[3 * k], c.ev[3 * k + 1]) != hipSuccess || hipEventElapsedTime(&b, c.ev[3 * k + 1], c.ev[3 * k + 2]) != hipSuccess) { (void)hipGetLastError(); continue; }
      if (c.iters[k] > 0) {
        lt_tot_.corr_ms += a; lt_tot_.corr_launches++; lt_tot_.corr_features += c.feats[k];
        lt_tot_.lm_ms += b; lt_tot_.lm_launches++; lt_tot_.lm_iterations += (uint64_t)c.iters[k]; lt_tot_.lm_bytes += c.bytes[k];
      } else {
        lt_tot_.corr_noop_ms += a; lt_tot_.corr_noop_launches++;
        lt_tot_.lm_noop_ms += b; lt_tot_.lm_noop_launches++;
      }
    }
    c.pending = false;
  }
}
OdometryBatch::LaunchTotals OdometryBatch::launch_totals() {
  std::lock_guard<std::mutex> lk(lt_mu_);
  lt_resolve_();
  return lt_tot_;
}

void odom_set_imu(OdomStream& S, const float* t);
ToEndParams odom_to_end_params(const OdomStream& S, float scan_period, bool enabled);
void odom_integrate_pose(OdomStream& S);

void OdometryBatch::update_imu(uint32_t s, const float* t) { odom_set_imu(*streams_[s], t); }

ToEndParams OdometryBatch::to_end_params(uint32_t s, bool enabled) const { return odom_to_end_params(*streams_[s], params.scan_period, enabled); }

void OdometryBatch::to_end_device(uint32_t s, float4* pts, uint32_t n) {
  if (!n) return;
  hipLaunchKernelGGL(k_transform_to_end, dim3((n + 255) / 256), dim3(256), 0, st_, pts, n, to_end_params(s, true));
}

// pose integration of one sweep (:626-649): transformSum <- transformSum (+) the sweep's optimised transform, with the IMU plug-in
void odom_integrate_pose(OdomStream& S) {
  HAngle rx, ry, rz;
  accumulate_rotation(S.transform_sum.rot_x, S.transform_sum.rot_y, S.transform_sum.rot_z, -S.transform.rot_x,
                      HAngle((float)(-S.transform.rot_y.r * 1.05)), -S.transform.rot_z, rx, ry, rz);
  HVec3 v{S.transform.pos.x - S.imu_shift.x, S.transform.pos.y - S.imu_shift.y, (float)(S.transform.pos.z * 1.05 - S.imu_shift.z)};
  h_rot_zxy(v, rz, rx, ry);
  HVec3 trans{S.transform_sum.pos.x - v.x, S.transform_sum.pos.y - v.y, S.transform_sum.pos.z - v.z};
  plugin_imu_rotation(rx, ry, rz, S.imu_pitch_start, S.imu_yaw_start, S.imu_roll_start, S.imu_pitch_end, S.imu_yaw_end,
                      S.imu_roll_end, rx, ry, rz);
  S.transform_sum.rot_x = rx; S.transform_sum.rot_y = ry; S.transform_sum.rot_z = rz;
  S.transform_sum.pos = trans;
}
// the IMU quantities of a sweep (updateIMU, :182-197): 4 x (x, y, z) = pitch / yaw / roll start, end, shift from start, velocity from start
void odom_set_imu(OdomStream& S, const float* t) {
  S.imu_pitch_start = HAngle(t[0]); S.imu_yaw_start = HAngle(t[1]); S.imu_roll_start = HAngle(t[2]);
  S.imu_pitch_end = HAngle(t[3]); S.imu_yaw_end = HAngle(t[4]); S.imu_roll_end = HAngle(t[5]);
  S.imu_shift = {t[6], t[7], t[8]};
  S.imu_velo = {t[9], t[10], t[11]};
}
// transformToEnd's parameters from a stream's host state (the transform's sin / cos as the host caches them, Angle.h)
ToEndParams odom_to_end_params(const OdomStream& S, float scan_period, bool enabled) {
  ToEndParams P;
  S.transform.get(P.T);
  const HAngle* ta[3] = {&S.transform.rot_x, &S.transform.rot_y, &S.transform.rot_z};
  const HAngle* sa[3] = {&S.imu_pitch_start, &S.imu_yaw_start, &S.imu_roll_start};
  const HAngle* ea[3] = {&S.imu_pitch_end, &S.imu_yaw_end, &S.imu_roll_end};
  for (int k = 0; k < 3; k++) {
    P.sT[k] = ta[k]->s; P.cT[k] = ta[k]->c;
    P.s_start[k] = sa[k]->s; P.c_start[k] = sa[k]->c;
    P.s_end[k] = ea[k]->s; P.c_end[k] = ea[k]->c;
  }
  P.shift[0] = S.imu_shift.x; P.shift[1] = S.imu_shift.y; P.shift[2] = S.imu_shift.z;
  P.scan_period = scan_period;
  P.enabled = enabled ? 1 : 0;
  return P;
}

void OdometryBatch::process(const OdomInput* in, int* rc, bool defer_tail) {
  TraceRange trace_range("loamx:odometry");
  LX_HIP(hipSetDevice(device_));
  const uint32_t ns = n_streams(), K = 2 * ns;
  if (up_pending_) {   // the previous call's uploads read the pinned staging buffers this call is about to rewrite (they finished long ago:
    LX_HIP(hipEventSynchronize(ev_up_));   // they precede its iterations); the device-side tail of that call needs no waiting for — same stream
    up_pending_ = false;
  }
  // A single-stream caller may hand the less-flat cloud over LATE (late_less_flat, process_linked): the iterations read the sharp / flat
  // features and the previous sweep's clouds only; the less-flat cloud of THIS sweep is first read by the tail (re-projection, index of
  // the "last" clouds).  Its producer — the per-ring voxel grid of the extraction, ~50 us on a stream of its own — then runs beside the
  // first launch pair instead of in front of it.  The callback is called once, before the tail is enqueued, and blocks until the cloud
  // and its size are known; until then the cloud counts as empty (offsets, staging) and the buffers are sized with late_bound.
  const bool late = ns == 1 && (bool)late_less_flat;
  std::vector<OdomInput> in_late;
  if (late) {
    in_late.assign(in, in + ns);
    in_late[0].less_flat = nullptr; in_late[0].n_less_flat = 0;
    in = in_late.data();
  }
  // ---- stage the current less-sharp / less-flat clouds of all streams (they are re-projected in place later)
  for (uint32_t s = 0; s < ns; s++) {
    h_cur_off_[s + 1] = h_cur_off_[s] + in[s].n_less_sharp;
    if (s > 0) LX_REQUIRE(in[s].less_sharp == in[s - 1].less_sharp + in[s - 1].n_less_sharp || in[s].n_less_sharp == 0 || in[s - 1].n_less_sharp == 0,
                          "less_sharp clouds of the streams must be contiguous");
  }
  for (uint32_t s = 0; s < ns; s++) h_cur_off_[ns + s + 1] = h_cur_off_[ns + s] + in[s].n_less_flat;
  const uint32_t n_corner_all = h_cur_off_[ns];
  uint32_t n_all = h_cur_off_[K];   // (late: without the less-flat cloud until resolve_late())
  cur_.reserve((size_t)n_all + (late ? late_bound : 0u) + 1);
  // per-type bulk copies when the inputs are contiguous, else per stream
  bool contig_c = true, contig_s = true;
  for (uint32_t s = 1; s < ns; s++) {
    contig_c = contig_c && in[s].less_sharp == in[s - 1].less_sharp + in[s - 1].n_less_sharp;
    contig_s = contig_s && in[s].less_flat == in[s - 1].less_flat + in[s - 1].n_less_flat;
  }
  // contiguous inputs (the feature extractor's layout) are not staged at all: the re-projection at the tail reads them
  const bool fused_stage = contig_c && contig_s;
  const float4* src_c = fused_stage ? in[0].less_sharp : nullptr;
  const float4* src_s = fused_stage ? in[0].less_flat : nullptr;
  bool late_pending = late, patch_pending = false;
  auto resolve_late = [&]() {
    if (!late_pending) return;
    late_pending = false;
    const float4* p = nullptr;
    uint32_t n = 0;
    late_less_flat(p, n);
    LX_REQUIRE(n <= late_bound, "internal: the late less-flat cloud is larger than its announced bound");
    in_late[0].less_flat = p; in_late[0].n_less_flat = n;
    h_cur_off_[ns + 1] = h_cur_off_[ns] + n;
    n_all = h_cur_off_[K];
    src_s = p;   // (ns == 1: contiguous by construction, the re-projection reads the cloud where it lies)
    patch_pending = true;   // the offsets again, now complete: with the next correspondence launch, or by a copy in front of the tail
  };
  auto flush_patch = [&]() {
    if (!patch_pending) return;
    patch_pending = false;
    h_off_late_.reserve(K + 1);
    memcpy(h_off_late_.p, h_cur_off_.data(), sizeof(uint32_t) * (K + 1));
    LX_HIP(hipMemcpyAsync(d_cur_off_.p, h_off_late_.p, sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice, st_));
    LX_HIP(hipEventRecord(ev_up_, st_));   // (the next call waits for this copy too before it rewrites the pinned blocks)
  };
  if (fused_stage) {
  } else if (contig_c) {
    if (n_corner_all) LX_HIP(hipMemcpyAsync(cur_.p, in[0].less_sharp, sizeof(float4) * n_corner_all, hipMemcpyDeviceToDevice, st_));
  } else {
    for (uint32_t s = 0; s < ns; s++)
      if (in[s].n_less_sharp) LX_HIP(hipMemcpyAsync(cur_.p + h_cur_off_[s], in[s].less_sharp, sizeof(float4) * in[s].n_less_sharp, hipMemcpyDeviceToDevice, st_));
  }
  if (fused_stage) {
  } else if (contig_s) {
    if (n_all > n_corner_all) LX_HIP(hipMemcpyAsync(cur_.p + n_corner_all, in[0].less_flat, sizeof(float4) * (n_all - n_corner_all), hipMemcpyDeviceToDevice, st_));
  } else {
    for (uint32_t s = 0; s < ns; s++)
      if (in[s].n_less_flat) LX_HIP(hipMemcpyAsync(cur_.p + h_cur_off_[ns + s], in[s].less_flat, sizeof(float4) * in[s].n_less_flat, hipMemcpyDeviceToDevice, st_));
  }

  // ---- problems of the streams that optimise this sweep
  xchg_epoch_ = xchg_epoch_ % 0xffffffu + 1u;   // (the records of a stream keep its position among the active ones only by accident: the tag, not the place, says whose they are)
  std::vector<uint32_t> active;
  uint32_t max_feat = 0, max_sharp = 0, max_flat = 0, ind_total = 0;
  std::vector<uint32_t> ind_off(ns + 1, 0);
  for (uint32_t s = 0; s < ns; s++) ind_off[s + 1] = ind_off[s] + 5 * (in[s].n_sharp + in[s].n_flat) + 5;
  ind_total = ind_off[ns];
  ind_.reserve(ind_total + 8);
  for (uint32_t s = 0; s < ns; s++) {
    OdomStream& S = *streams_[s];
    const OdomInput& I = in[s];
    if (!S.inited) {   // :198-211: only stash the clouds and seed transformSum with the IMU start angles
      S.transform_sum.rot_x = HAngle(S.transform_sum.rot_x.r + S.imu_pitch_start.r);
      S.transform_sum.rot_z = HAngle(S.transform_sum.rot_z.r + S.imu_roll_start.r);
      rc[s] = LOAMX_SKIPPED;
      continue;
    }
    rc[s] = LOAMX_OK;
    S.frame++;
    S.transform.pos.x -= S.imu_velo.x * params.scan_period;
    S.transform.pos.y -= S.imu_velo.y * params.scan_period;
    S.transform.pos.z -= S.imu_velo.z * params.scan_period;
    S.stats = {0, 0, (int)S.frame, 0};
    if (S.n_last_corner > 10 && S.n_last_surf > 100) {
      OdomProblem& pb = h_prob_.p[active.size()];
      pb.sharp = I.sharp; pb.n_sharp = I.n_sharp;
      pb.flat = I.flat; pb.n_flat = I.n_flat;
      pb.last_corner = last_.p + h_last_off_[s]; pb.n_last_corner = S.n_last_corner;
      pb.last_surf = last_.p + h_last_off_[ns + s]; pb.n_last_surf = S.n_last_surf;
      pb.sorted = index_.sorted();
      pb.cell_table = index_.cell_table();
      pb.lc_desc = index_.desc(s);
      pb.ls_desc = index_.desc(ns + s);
      pb.ind = ind_.p + ind_off[s];
      S.transform.get(pb.transform);
      pb.stats = {0, 0, 0, 0};
      pb.done = 0;
      pb.ticket = 0;
      pb.xchg_epoch = xchg_epoch_;
      pb.stream_id = (int)s;
      pb.host_mirror = h_mirror_.p + active.size();
      pb.te_out = te_.p + s;
      pb.part = part_.p + (size_t)active.size() * OD_PART_STRIDE;
      pb.err_word = h_err_.p;
      pb.rf_corner = rf_.p + (size_t)s * OD_RF_N;
      pb.rf_surf = rf_.p + (size_t)(ns + s) * OD_RF_N;
      pb.rf_epoch = rf_epoch_;   // the table the previous call's re-projection left behind
      max_feat = std::max(max_feat, I.n_sharp + I.n_flat);
      max_sharp = std::max(max_sharp, I.n_sharp);
      max_flat = std::max(max_flat, I.n_flat);
      active.push_back(s);
    }
  }
  const uint32_t na = (uint32_t)active.size();
  if (max_feat > 32 * OD_THREADS) throw Error(LOAMX_E_CAPACITY, "more than 8192 sharp+flat features in one sweep");
  // ---- everything the device needs for the whole sweep goes up first: problems, cloud offsets, re-projection
  // parameters (those of the optimising streams are completed on the device, k_te_patch)
  for (uint32_t s = 0; s < ns; s++) h_te_.p[s] = to_end_params(s, rc[s] == LOAMX_OK);   // a first sweep is stored as it came (:200-201)
  memcpy(h_off_pin_.p, h_cur_off_.data(), sizeof(uint32_t) * (K + 1));
  // (measured, round 6: fetching this block by kernel instead — pinned_copy.hpp — is no faster: 16.80 k against 16.93 k sweeps/s, profiles/r06_ab.md)
  LX_HIP(hipMemcpyAsync(up_dev_.p, up_host_.p, up_bytes_, hipMemcpyHostToDevice, st_));   // problems + re-projection parameters + offsets
  index_.prepare(K);   // bounding-box accumulators of the NEXT index (a launch only the first time: every build leaves them reset)
  const bool index_prepared = true;
  if (!ev_up_) LX_HIP(hipEventCreateWithFlags(&ev_up_, hipEventDisableTiming));
  LX_HIP(hipEventRecord(ev_up_, st_));
  up_pending_ = true;
  LtCall* lt = nullptr;
  if (launch_timing_.load(std::memory_order_relaxed) && na && max_feat) {
    std::lock_guard<std::mutex> lk(lt_mu_);
    lt_resolve_();
    LtCall& c = lt_[lt_next_ % LT_RING];
    if (!c.pending) { lt = &c; lt_next_++; c.pairs = 0; }   // (a call whose slot is still in flight goes untimed)
  }
  if (na) {
    if (max_feat) {
      // Launch pairs (correspondences + up to five iterations).  The reference leaves its loop when the stop test fires
      // (BasicLaserOdometry.cpp:613-620); launches enqueued behind a sweep that has converged cost ~10 us each, so they are enqueued as
      // they turn out to be needed: the host reads from the pinned mirror (k_odom_lm writes a stream's state there at the end of every
      // launch) whether a pair left any stream unconverged.  Round 5 enqueued all five pairs up front (1.6 empty pairs per pass on
      // average).  Result-neutral by construction: a launch that is not enqueued would have returned at its first instruction (pb.done /
      // the iteration bound).  LOAMX_ODOM_PAIRS: `lag` (default) speculates on ONE launch (the next pair's correspondences), `lag2` on one
      // PAIR (round 6's first form), `exact` on nothing (a host round trip in front of every further pair and of the tail), `all` = round 5.
      const int pair_mode = [] { const char* e = getenv("LOAMX_ODOM_PAIRS"); return !e ? 1 : !strcmp(e, "all") ? 0 : !strcmp(e, "exact") ? 2 : !strcmp(e, "lag2") ? 3 : 1; }();
      const int maxp = (params.max_iterations + 4) / 5;
      for (uint32_t a = 0; a < na; a++) { h_mirror_.p[a].done = 0; h_mirror_.p[a].stats.iterations = 0; }   // (the previous sweep's launches finished before its poses were read)
      // the two halves of a pair, enqueued apart: `lag` speculates on the correspondence launch only
      auto enqueue_corr = [&](int k) {
        const int lk = lt && k < LT_MAXP ? k : -1;
        if (lk >= 0) {
          for (int e = 0; e < 3; e++) if (!lt->ev[3 * lk + e]) LX_HIP(hipEventCreate(&lt->ev[3 * lk + e]));
          LX_HIP(hipEventRecord(lt->ev[3 * lk], st_));
        }
        uint32_t* pd = nullptr;
        uint4 pv = make_uint4(0u, 0u, 0u, 0u);
        if (patch_pending) { patch_pending = false; pd = d_cur_off_.p; pv = make_uint4(h_cur_off_[0], h_cur_off_[1], h_cur_off_[2], 0u); }   // (ns == 1: K + 1 = 3 words)
        hipLaunchKernelGGL(k_odom_corr_grid, dim3(8 * (((max_sharp + 7) / 8 + (max_flat + 7) / 8 + 3) / 4), na), dim3(256), 0, st_, prob_.p, params, pd, pv);
        if (lk >= 0) LX_HIP(hipEventRecord(lt->ev[3 * lk + 1], st_));
      };
      auto enqueue_lm = [&](int k) {
        const int it0 = 5 * k;
        const int nit = std::min(5, params.max_iterations - it0);
        const int lk = lt && k < LT_MAXP ? k : -1;
#ifdef OD_LM_HALF_WGS   // (measurement: half as many workgroups per stream, two features per thread — profiles/r05_ab.md section 2)
        const uint32_t nb = std::min<uint32_t>(16u, std::max<uint32_t>(1u, (max_feat + 2 * OD_THREADS - 1) / (2 * OD_THREADS)));
#else
        const uint32_t nb = std::min<uint32_t>(16u, (max_feat + OD_THREADS - 1) / OD_THREADS);
#endif
        // k_odom_lm's workgroups of one stream spin on each other: everything a launch puts on the device must be resident
        // at once.  The launch is cut into chunks of streams that fill at most half of what the device can hold (occupancy x
        // CUs, queried once) — the registration and feature kernels of the other HIP streams share the CUs.
        const bool two = max_feat > nb * OD_THREADS;
        if (!lm_slots_[two]) {
          int per_cu = 0;
          if (two) LX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_odom_lm<2>, OD_THREADS, 0));
          else LX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_odom_lm<1>, OD_THREADS, 0));
          hipDeviceProp_t prop;
          LX_HIP(hipGetDeviceProperties(&prop, device_));
          lm_slots_[two] = (uint32_t)std::max(per_cu, 1) * (uint32_t)std::max(prop.multiProcessorCount, 1);
        }
        const uint32_t chunk = std::max<uint32_t>(1u, (lm_slots_[two] / 2) / nb);
        for (uint32_t a0 = 0; a0 < na; a0 += chunk) {
          const uint32_t nc = std::min(chunk, na - a0);
          if (two) hipLaunchKernelGGL(k_odom_lm<2>, dim3(nb, nc), dim3(OD_THREADS), 0, st_, prob_.p + a0, params, it0, nit);
          else hipLaunchKernelGGL(k_odom_lm<1>, dim3(nb, nc), dim3(OD_THREADS), 0, st_, prob_.p + a0, params, it0, nit);
        }
        if (lk >= 0) { LX_HIP(hipEventRecord(lt->ev[3 * lk + 2], st_)); lt->pairs = lk + 1; }
      };
      auto enqueue_pair = [&](int it0) { enqueue_corr(it0 / 5); enqueue_lm(it0 / 5); };
      const volatile OdomProblem* hm = h_mirror_.p;
      auto settled = [&](int k) {     // every stream is through launch pair k (or had converged before it)
        const int want = std::min(5 * (k + 1), params.max_iterations);
        for (uint32_t a = 0; a < na; a++) if (hm[a].done == 0 && hm[a].stats.iterations < want) return false;
        return true;
      };
      auto converged = [&]() {        // no stream has anything left to iterate
        for (uint32_t a = 0; a < na; a++) if (hm[a].done == 0 && hm[a].stats.iterations < params.max_iterations) return false;
        return true;
      };
      auto wait_settled = [&](int k) {   // false: the mirror did not answer in time (the caller then enqueues the rest unconditionally)
        const auto t_in = std::chrono::steady_clock::now();
        for (unsigned spins = 0; !settled(k);) {
          if ((++spins & 255u) == 0u && std::chrono::steady_clock::now() - t_in > std::chrono::milliseconds(20)) return false;
          __builtin_ia32_pause();
        }
        return true;
      };
      int enq = 0;                    // launch pairs whose iterations have been enqueued
      if (pair_mode == 1) {
        // `lag` (default): pair 0, then only the NEXT pair's correspondence launch ahead of need.  When pair k's iterations have ended the
        // host knows whether pair k + 1 is needed: if so it enqueues its iterations (and the correspondence launch of pair k + 2) while
        // the correspondence launch of k + 1 runs — the queue never runs dry; if not, that one launch is the pass's only empty one.
        enqueue_corr(0);
        enqueue_lm(0);
        enq = 1;
        if (maxp > 1) enqueue_corr(1);
        resolve_late();   // (the device has a launch pair and a half in its queue: the wait for the less-flat cloud costs the chain nothing)
        while (enq < maxp) {
          if (!wait_settled(enq - 1)) {   // blind: everything that is left, unconditionally (always correct)
            enqueue_lm(enq);
            for (enq++; enq < maxp; enq++) enqueue_pair(5 * enq);
            break;
          }
          if (converged()) break;
          enqueue_lm(enq);
          enq++;
          if (enq < maxp) enqueue_corr(enq);
        }
      } else {
      const int first = pair_mode == 0 ? maxp : std::min(maxp, pair_mode == 2 ? std::max(1, pred_pairs_) : 2);
      for (; enq < first; enq++) enqueue_pair(5 * enq);
      resolve_late();
      bool blind = false;             // the mirror did not answer in time: enqueue the rest unconditionally (always correct)
      while (enq < maxp) {
        const int watch = pair_mode == 2 ? enq - 1 : enq - 2;
        if (!blind && watch >= 0) {
          if (!wait_settled(watch)) blind = true;
          if (!blind && converged()) break;
        }
        enqueue_pair(5 * enq);
        enq++;
      }
      }
      pairs_enqueued_ += (uint64_t)enq;
      pair_calls_++;
    }
    if (!max_feat) LX_HIP(hipMemcpyAsync(h_mirror_.p, prob_.p, sizeof(OdomProblem) * na, hipMemcpyDeviceToHost, st_));   // (no launch wrote the mirror)
    if (!ev_pose_) LX_HIP(hipEventCreateWithFlags(&ev_pose_, hipEventDisableTiming));
    LX_HIP(hipEventRecord(ev_pose_, st_));
    if (!max_feat) hipLaunchKernelGGL(k_te_patch, dim3((na + 63) / 64), dim3(64), 0, st_, prob_.p, na, te_.p);   // (otherwise k_odom_lm did it)
  }
  // ---- re-project to the sweep end (:651-652), hand over as "last" clouds and rebuild their index (:654-664): enqueued
  // right behind the iterations; the host only waits for the poses
  static const bool fuse_bounds = !(diag_env("LOAMX_BB_FUSED") && atoi(diag_env("LOAMX_BB_FUSED")) == 0);   // diagnostic: 0 = separate k_bb_bbox launch
  resolve_late();   // (a sweep without iterations: the tail is the first reader)
  flush_patch();    // (no correspondence launch took the completed offsets along)
  if (n_all)
  {
    if (++rf_epoch_ > 255u) rf_epoch_ = 1u;   // entries of this re-projection carry the new epoch; the problems of the NEXT call name it
    hipLaunchKernelGGL(k_transform_to_end_batch, dim3((n_all + 255) / 256), dim3(256), 0, st_, cur_.p, n_all, d_cur_off_.p, K, ns, te_.p, src_c,
                       src_s, n_corner_all, fuse_bounds ? index_.d_bounds() : nullptr, rf_.p, rf_epoch_);
  }
  // rotating buffers: the clouds this call hands on stay untouched during the next keep_ calls (a registration reads them while the
  // odometry chain is already that many sweeps ahead — Pipeline)
  {
    float4* p = cur_.p; size_t c = cur_.cap;
    DevBuf<float4>& oldest = older_[keep_ - 2];
    cur_.p = oldest.p; cur_.cap = oldest.cap;
    for (int k = keep_ - 2; k > 0; k--) { older_[k].p = older_[k - 1].p; older_[k].cap = older_[k - 1].cap; }
    older_[0].p = last_.p; older_[0].cap = last_.cap;
    last_.p = p; last_.cap = c;
  }
  h_last_off_ = h_cur_off_;
  index_.build(last_.p, h_last_off_.data(), K, d_cur_off_.p, index_prepared, /*bounds_done=*/fuse_bounds && n_all > 0);
  if (!ev_tail_) LX_HIP(hipEventCreateWithFlags(&ev_tail_, hipEventDisableTiming));
  LX_HIP(hipEventRecord(ev_tail_, st_));
  tail_pending_ = true;
  if (na) {
    wait_event(ev_pose_);
    if (*(volatile uint32_t*)h_err_.p) {   // k_odom_lm's exchange timed out (its workgroups were not all resident)
      *h_err_.p = 0u;
      throw Error(LOAMX_E_HIP, "odometry: the exchange between a stream's k_odom_lm workgroups timed out (not all of them resident)");
    }
#ifdef LOAMX_PROF_CORR
    {
      unsigned long long ts[3][16];
      LX_HIP(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_corr_ts), sizeof(ts)));
      for (int q = 0; q < 3; q++)
        fprintf(stderr, "[corr ts %s, us since entry] feature+trig %.2f block table %.2f nn1 %.2f decision %.2f block windows %.2f walk %.2f end %.2f | block %llu pts, via block %llu, closest %lld cscan %lld\n",
                q == 0 ? "corner" : q == 1 ? "flat mid" : "flat last", (ts[q][1] - ts[q][0]) * 0.01, (ts[q][2] - ts[q][0]) * 0.01, (ts[q][3] - ts[q][0]) * 0.01,
                (ts[q][4] - ts[q][0]) * 0.01, (ts[q][5] - ts[q][0]) * 0.01, (ts[q][6] - ts[q][0]) * 0.01, (ts[q][7] - ts[q][0]) * 0.01, ts[q][8], ts[q][9],
                (long long)ts[q][10], (long long)ts[q][11]);
    }
#endif
#ifdef LOAMX_PROF_LM
    {
      double ts[16];
      LX_HIP(hipMemcpy(ts, part_.p + OD_PART_TS, sizeof(ts), hipMemcpyDeviceToHost));
      fprintf(stderr, "[lm ts, 10ns ticks]");
      for (int k = 1; k < 10; k++) fprintf(stderr, " %d:%+.0f", k, ts[k] - ts[0]);
      fprintf(stderr, "\n");
    }
#endif
    {
      int need = 1;
      for (uint32_t a = 0; a < na; a++) need = std::max(need, (h_mirror_.p[a].stats.iterations + 4) / 5);
      pred_pairs_ = need;   // (LOAMX_ODOM_PAIRS=exact: a stream that needs every iteration needs them for several sweeps in a row)
    }
    for (uint32_t a = 0; a < na; a++) {
      OdomStream& S = *streams_[active[a]];
      // _transform.rot_* = rad + x re-derives the cached sin/cos (:599-601)
      S.transform.set(h_mirror_.p[a].transform);
      S.stats.iterations = h_mirror_.p[a].stats.iterations;
      S.stats.sel = h_mirror_.p[a].stats.sel;
      S.stats.degenerate = h_mirror_.p[a].stats.degenerate;
    }
    if (lt) {   // what every timed launch did, now that the iteration counts are known
      std::lock_guard<std::mutex> lk(lt_mu_);
      for (int k = 0; k < lt->pairs; k++) {
        lt->iters[k] = 0; lt->bytes[k] = 0; lt->feats[k] = 0;
        for (uint32_t a = 0; a < na; a++) {
          const int it = h_mirror_.p[a].stats.iterations - 5 * k;
          if (it <= 0) continue;
          const uint64_t nf = (uint64_t)in[active[a]].n_sharp + in[active[a]].n_flat;
          lt->iters[k] = std::max(lt->iters[k], std::min(it, 5));
          lt->bytes[k] += 48ull * nf;
          lt->feats[k] += nf;
        }
      }
      lt->pending = true;
    }
  }
  // ---- pose integration (:626-649)
  for (uint32_t s = 0; s < ns; s++) {
    OdomStream& S = *streams_[s];
    if (rc[s] == LOAMX_OK) odom_integrate_pose(S);
    S.inited = true;
    S.n_last_corner = in[s].n_less_sharp;
    S.n_last_surf = in[s].n_less_flat;
  }
  if (!defer_tail) {   // (a deferring caller orders its consumers behind tail_event() instead of blocking the host here)
    LX_HIP(hipStreamSynchronize(st_));
    tail_pending_ = false;
  }
}

void OdometryBatch::to_end_gather(float4* dst, const uint32_t* h_off, const float4* const* src, const ToEndParams* seg_params, uint32_t K,
                                  hipStream_t stream) {
  const uint32_t n = h_off[K];
  if (!n) return;
  // staging layout in pinned memory: params[K] | src ptrs[K] | off[K+1] | sid[K] (sid[k] = k)
  const uint32_t ns = K;
  const size_t bytes = sizeof(ToEndParams) * ns + sizeof(float4*) * K + sizeof(uint32_t) * (2 * (size_t)K + 2) + 64;
  h_gather_.reserve(bytes);
  d_gather_.reserve(bytes);
  char* h = h_gather_.p;
  ToEndParams* hp = (ToEndParams*)h;
  for (uint32_t s = 0; s < ns; s++) hp[s] = seg_params[s];
  const size_t o_src = (sizeof(ToEndParams) * ns + 15) & ~(size_t)15;
  const float4** hs = (const float4**)(h + o_src);
  for (uint32_t k = 0; k < K; k++) hs[k] = src[k];
  const size_t o_off = o_src + sizeof(float4*) * K;
  uint32_t* ho = (uint32_t*)(h + o_off);
  memcpy(ho, h_off, sizeof(uint32_t) * (K + 1));
  uint32_t* hsid = ho + K + 1;
  for (uint32_t k = 0; k < K; k++) hsid[k] = k;
  LX_HIP(hipMemcpyAsync(d_gather_.p, h, o_off + sizeof(uint32_t) * (2 * (size_t)K + 1), hipMemcpyHostToDevice, stream));
  char* d = d_gather_.p;
  hipLaunchKernelGGL(k_to_end_gather, dim3((n + 255) / 256), dim3(256), 0, stream, dst, n, (const uint32_t*)(d + o_off), K,
                     (const float4* const*)(d + o_src), (const uint32_t*)(d + o_off) + K + 1, (const ToEndParams*)d);
}

int OdometryBatch::process_host(const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat,
                                const loamx_cloud* less_flat) {
  LX_REQUIRE(n_streams() == 1, "process_host is the single-stream entry point");
  LX_HIP(hipSetDevice(device_));
  const loamx_cloud* cl[4] = {sharp, less_sharp, flat, less_flat};
  // the four clouds go up back to back: one pinned block, one copy, no wait (process() orders everything behind it on the stream;
  // the staging block is rewritten by the next call only, which starts after this one has synchronised)
  size_t off[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < 4; k++) {
    check_cloud(cl[k], false);
    off[k + 1] = off[k] + cl[k]->count;
  }
  h_stage_.reserve(off[4] + 1);
  up_[0].reserve(off[4] + 1);
  for (int k = 0; k < 4; k++) pack_cloud(cl[k], h_stage_.p + off[k]);
  // feature clouds are finite by contract (BasicLaserOdometry.cpp:230, :252 strip NaN points as a safeguard; the registration stage
  // never produces them): a caller that hands over NaN / Inf coordinates is told instead of getting a pose through NaN arithmetic
  if (!packed_all_finite(h_stage_.p, off[4])) throw Error(LOAMX_E_INVALID, "a feature cloud holds non-finite coordinates");
  fetch_from_pinned(up_[0].p, h_stage_.p, off[4], st_);   // (by kernel: pinned_copy.hpp)
  OdomInput in{up_[0].p, sharp->count, up_[0].p + off[1], less_sharp->count, up_[0].p + off[2], flat->count, up_[0].p + off[3], less_flat->count};
  int rc = LOAMX_OK;
  last_dl_valid_ = false;
  process(&in, &rc, /*defer_tail=*/true);   // returns with the pose; the re-projection / index build go on behind it
  // what get_last_clouds() hands out is asked for now: the copy lands in pinned memory behind the tail while the caller is busy
  // with the pose (ns = 1: the corner and the surf cloud lie back to back)
  OdomStream& S = *streams_[0];
  const uint32_t n = S.n_last_corner + S.n_last_surf;
  h_last_dl_.reserve((size_t)n + 1);
  if (n) LX_HIP(hipMemcpyAsync(h_last_dl_.p, d_last_corner(0), sizeof(float4) * n, hipMemcpyDeviceToHost, st_));
  last_dl_valid_ = d_last_surf(0) == d_last_corner(0) + S.n_last_corner;
  return rc;
}

int OdometryBatch::process_linked(const float4* const feat[4], const uint32_t n_feat[4], const float4* d_full, uint32_t n_full,
                                  const std::function<void(const float4*&, uint32_t&)>& less_flat_late) {
  LX_REQUIRE(n_streams() == 1, "process_linked is a single-stream entry point");
  LX_HIP(hipSetDevice(device_));
  OdomInput in{feat[0], n_feat[0], feat[1], n_feat[1], feat[2], n_feat[2], feat[3], n_feat[3]};
  int rc = LOAMX_OK;
  last_dl_valid_ = false;
  link_valid_ = false;
  struct Late {   // feat[3] / n_feat[3] are ignored when the cloud comes late (a subset of the sweep: at most n_full points)
    OdometryBatch& o;
    ~Late() { o.late_less_flat = nullptr; o.late_bound = 0; }
  } late_guard{*this};
  if (less_flat_late) { late_less_flat = less_flat_late; late_bound = n_full; }
  process(&in, &rc, /*defer_tail=*/true);
  link_full_.reserve((size_t)n_full + 1);
  if (n_full)
    hipLaunchKernelGGL(k_transform_to_end_copy, dim3((n_full + 255) / 256), dim3(256), 0, st_, link_full_.p, d_full, n_full, to_end_params(0, true));
  n_link_full_ = n_full;
  if (!ev_link_) LX_HIP(hipEventCreateWithFlags(&ev_link_, hipEventDisableTiming));
  LX_HIP(hipEventRecord(ev_link_, st_));
  link_valid_ = true;
  return rc;
}

int OdometryBatch::get_last_clouds(uint32_t s, loamx_cloud* corner, loamx_cloud* surf) {
  LX_HIP(hipSetDevice(device_));
  OdomStream& S = *streams_[s];
  int rc = LOAMX_OK;
  if (corner) check_cloud(corner, false);
  if (surf) check_cloud(surf, false);
  if (last_dl_valid_ && s == 0) {   // process_host() asked for them already
    LX_HIP(hipStreamSynchronize(st_));
    if (corner) {
      const int r = unpack_cloud(h_last_dl_.p, S.n_last_corner, corner);
      if (r != LOAMX_OK) rc = r;
    }
    if (surf) {
      const int r = unpack_cloud(h_last_dl_.p + S.n_last_corner, S.n_last_surf, surf);
      if (r != LOAMX_OK) rc = r;
    }
    return rc;
  }
  // both clouds through pinned memory, one wait (copies into pageable memory are staged by the runtime, each with a wait of its own)
  const uint32_t nc = corner ? S.n_last_corner : 0u, nsf = surf ? S.n_last_surf : 0u;
  h_stage_.reserve((size_t)nc + nsf + 1);
  if (nc) LX_HIP(hipMemcpyAsync(h_stage_.p, d_last_corner(s), sizeof(float4) * nc, hipMemcpyDeviceToHost, st_));
  if (nsf) LX_HIP(hipMemcpyAsync(h_stage_.p + nc, d_last_surf(s), sizeof(float4) * nsf, hipMemcpyDeviceToHost, st_));
  if (nc + nsf) LX_HIP(hipStreamSynchronize(st_));
  if (corner) {
    const int r = unpack_cloud(h_stage_.p, nc, corner);
    if (r != LOAMX_OK) rc = r;
  }
  if (surf) {
    const int r = unpack_cloud(h_stage_.p + nc, nsf, surf);
    if (r != LOAMX_OK) rc = r;
  }
  return rc;
}

int OdometryBatch::transform_to_end_host(uint32_t s, loamx_cloud* cloud) {
  LX_HIP(hipSetDevice(device_));
  check_cloud(cloud, false);
  const uint32_t n = cloud->count;
  h_stage_.reserve(n + 1);
  tmp_cloud_.reserve(n + 1);
  pack_cloud(cloud, h_stage_.p);
  fetch_from_pinned(tmp_cloud_.p, h_stage_.p, n, st_);
  to_end_device(s, tmp_cloud_.p, n);
  if (n) LX_HIP(hipMemcpyAsync(h_stage_.p, tmp_cloud_.p, sizeof(float4) * n, hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
  return unpack_cloud(h_stage_.p, n, cloud);
}

}  // namespace loamx
