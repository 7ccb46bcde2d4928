// Scan-to-map registration kernels for gfx950 (MI355X).
//
// What runs here, per batch of independent sweeps (reference: src/lib/BasicLaserMapping.cpp):
//   k_stack          stack round trip  pointAssociateToMap -> pointAssociateTobeMapped        (:282-292, :512-516)
//   k_keys/sort/k_voxel_reduce   pcl::VoxelGrid on the stack clouds (corner 0.2 m / surf 0.4 m) (:519-527)
//   SubMapIndex      replaces the two kd-tree rebuilds (:636-637) by a counting-sorted uniform grid
//   k_gn_iter        one Gauss-Newton iteration, fused: per query (4 lanes each) pointAssociateToMap + exact 5-NN within the
//                    1 m gate (:668-671, :757-760); per query (one lane) 3x3 eigen edge fit or 5x3 QR plane fit, residual +
//                    weight, Jacobian row (:673-861); per tile J^T J / J^T r (:864-866)
//   solve_sweep      (last workgroup of a sweep in k_gn_iter) 6x6 column-pivoted QR, degeneracy projector, pose update,
//                    convergence test   (:867-922)
//   k_transform_full transformFullResToMap                                                         (:235-240)
// HBM-bound gather work: no MFMA (the only dense contraction is 6x6).  Points are packed float4 so a neighbour is one
// 16-byte load; queries are processed in voxel order, so the lanes of a wave walk the same few grid cells.
#include "registration.hpp"
#include "pinned_copy.hpp"
#include <atomic>
#include <chrono>
#include "scan.hpp"
#include "voxel.hpp"

namespace loamx {

// ----------------------------------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------------------------------

__global__ void k_init_bbox(uint32_t* scratch) {
  const uint32_t t = threadIdx.x;
  if (t < 3) scratch[t] = 0xffffffffu;
  else if (t < 16) scratch[t] = 0u;
}
__global__ void k_zero_u32_dn(uint32_t* p, const uint32_t* d_n) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}

// ----------------------------------------------------------------------------------------------------------------
// SubMapIndex: bounding box -> grid descriptor -> cell histogram -> exclusive scan -> scatter
// scratch layout (uint32): [0..5] encoded min xyz / max xyz, [6] ncell+1, [7] scan total, [8] ncell
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bbox(const float4* __restrict__ pts, uint32_t n, uint32_t* __restrict__ enc) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pts[i];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
    }
  }
  __shared__ float red[4][6];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { red[wid][a] = mn[a]; red[wid][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = red[0][a];
    for (int w = 1; w < 4; w++) v = a < 3 ? fminf(v, red[w][a]) : fmaxf(v, red[w][a]);
    if (a < 3) atomicMin(&enc[a], enc_f32(v)); else atomicMax(&enc[a], enc_f32(v));
  }
}

// grid descriptor from the bounds (k_bbox, or the producer of the points through SubMapIndex::d_bounds()): the cell edge starts at
// 1.05 m and grows by 1.25x while the table would not fit (a coarser grid is still exact: the 27-cell neighbourhood only grows)
__device__ inline GridDesc grid_from_bounds(const uint32_t* __restrict__ scratch, uint32_t max_cells) {
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = dec_f32(scratch[a]); mx[a] = dec_f32(scratch[3 + a]); }
  float h = 1.05f;
  GridDesc g;
  for (;;) {
    g.inv_h = 1.0f / h;
    g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2];
    g.nx = (int)floorf((mx[0] - mn[0]) * g.inv_h) + 1;
    g.ny = (int)floorf((mx[1] - mn[1]) * g.inv_h) + 1;
    g.nz = (int)floorf((mx[2] - mn[2]) * g.inv_h) + 1;
    unsigned long long nc = (unsigned long long)g.nx * g.ny * g.nz;
    if (nc <= max_cells) { g.ncell = (uint32_t)nc; break; }
    h *= 1.25f;
  }
  return g;
}

// count: every workgroup derives the descriptor itself (the same arithmetic everywhere; workgroup 0 records it and the scan's count);
// a point's rank inside its cell is the counter's value before its run's bump, so the scatter needs no atomics and the counters can
// be cleared behind the scan (rounds 1-4: k_init_bbox, k_grid_setup and k_zero_u32_dn were three launches of their own)
__global__ __launch_bounds__(256) void k_cell_count(const float4* __restrict__ pts, uint32_t n, uint32_t* __restrict__ scratch, GridDesc* __restrict__ desc,
                                                    uint32_t max_cells, uint32_t* __restrict__ cell_of, uint32_t* __restrict__ counts,
                                                    uint32_t* __restrict__ rank_of) {
  __shared__ GridDesc s_g;
  if (threadIdx.x == 0) {
    s_g = grid_from_bounds(scratch, max_cells);
    if (blockIdx.x == 0) {
      *desc = s_g;
      scratch[6] = s_g.ncell + 1;
      scratch[8] = s_g.ncell;
    }
  }
  __syncthreads();
  const GridDesc g = s_g;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  uint32_t c = 0;
  if (active) {
    const float4 p = pts[i];
    int cx, cy, cz;
    cell_coords(g, p.x, p.y, p.z, cx, cy, cz);
    c = ((uint32_t)cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
  }
  int head, len;
  wave_runs(c, active, head, len);
  uint32_t base = 0;
  if (active && head == (int)__lane_id()) base = atomicAdd(&counts[c], (uint32_t)len);
  base = __shfl(base, head, 64);
  if (active) rank_of[i] = base + (uint32_t)((int)__lane_id() - head);
}

// scatter; its first thread leaves the bounds accumulators reset for the next build
__global__ __launch_bounds__(256) void k_cell_scatter(const float4* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ cell_of,
                                                      const uint32_t* __restrict__ rank_of, const uint32_t* __restrict__ cell_start,
                                                      float4* __restrict__ sorted, uint32_t* __restrict__ scratch) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
#pragma unroll
    for (int a = 0; a < 6; a++) scratch[a] = a < 3 ? 0xffffffffu : 0u;
  }
  if (i >= n) return;
  float4 p = pts[i];
  p.w = __uint_as_float(i);   // original index: kNN ties are broken on it, so the slot order inside a cell is irrelevant
  sorted[cell_start[cell_of[i]] + rank_of[i]] = p;
}

void SubMapIndex::init(hipStream_t st) {
  st_ = st;
  scratch_.reserve(16);
  hipLaunchKernelGGL(k_init_bbox, dim3(1), dim3(16), 0, st, scratch_.p);   // (once: every build's scatter leaves the bounds reset)
  d_desc_.reserve(1);
  tile_sums_.reserve(SCAN_SCRATCH_WORDS);
  LX_HIP(hipMemsetAsync(tile_sums_.p, 0, sizeof(uint32_t) * tile_sums_.cap, st));
}

void SubMapIndex::swap(SubMapIndex& o) {
  std::swap(n_, o.n_);
  auto sw = [](auto& a, auto& b) { std::swap(a.p, b.p); std::swap(a.cap, b.cap); };
  sw(sorted_, o.sorted_); sw(cell_of_, o.cell_of_); sw(rank_of_, o.rank_of_); sw(cell_start_, o.cell_start_); sw(cursor_, o.cursor_);
  sw(tile_sums_, o.tile_sums_); sw(scratch_, o.scratch_); sw(d_desc_, o.d_desc_);
}

void SubMapIndex::build(const float4* d_pts, uint32_t n, bool bounds_done) {
  n_ = n;
  if (n == 0) return;
  sorted_.reserve(n);
  cell_of_.reserve(n);
  rank_of_.reserve(n);
  cell_start_.reserve((size_t)LX_MAX_CELLS + 2);
  if (!cursor_.p) {   // the cell counters: cleared once, kept clear by every build (the scan clears them behind itself)
    cursor_.reserve((size_t)LX_MAX_CELLS + 2);
    LX_HIP(hipMemsetAsync(cursor_.p, 0, sizeof(uint32_t) * cursor_.cap, st_));
  }
  const uint32_t nb = (n + 255) / 256;
  // bounds (unless the kernel that produced the points folded them into d_bounds() as it wrote them) -> count (+ grid set-up) -> scan
  // (+ counters cleared) -> scatter (+ bounds reset): 3 - 4 launches (rounds 1-4: 7)
  if (!bounds_done) hipLaunchKernelGGL(k_bbox, dim3(nb < 128 ? nb : 128), dim3(256), 0, st_, d_pts, n, scratch_.p);
  hipLaunchKernelGGL(k_cell_count, dim3(nb), dim3(256), 0, st_, d_pts, n, scratch_.p, d_desc_.p, LX_MAX_CELLS, cell_of_.p, cursor_.p, rank_of_.p);
  exclusive_scan_u32(cursor_.p, cell_start_.p, tile_sums_.p, scratch_.p + 8, scratch_.p + 7, LX_MAX_CELLS, st_, nullptr, cursor_.p);
  hipLaunchKernelGGL(k_cell_scatter, dim3(nb), dim3(256), 0, st_, d_pts, n, cell_of_.p, rank_of_.p, cell_start_.p, sorted_.p, scratch_.p);
  LX_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------------------------
// SubMapIndexBatch: the same counting-sort build for K clouds with one set of launches
// scratch: [0] total cells + 1, [1] scan total, [2] total cells
// ----------------------------------------------------------------------------------------------------------------
__global__ void k_bb_init(uint32_t* enc, uint32_t K) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * K) enc[i * BB_STRIDE] = (i % 6) < 3 ? 0xffffffffu : 0u;
}
// grid = (blocks, K)
__global__ __launch_bounds__(256) void k_bb_bbox(const float4* __restrict__ pts, const uint32_t* __restrict__ off, uint32_t* __restrict__ enc) {
  const uint32_t c = blockIdx.y;
  const uint32_t a0 = off[c], a1 = off[c + 1];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = a0 + blockIdx.x * blockDim.x + threadIdx.x; i < a1; i += gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
    }
  }
  __shared__ float red[4][6];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { red[wid][a] = mn[a]; red[wid][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6 && a1 > a0 + blockIdx.x * blockDim.x) {
    const int a = threadIdx.x;
    float v = red[0][a];
    for (int w = 1; w < 4; w++) v = a < 3 ? fminf(v, red[w][a]) : fmaxf(v, red[w][a]);
    if (a < 3) atomicMin(&enc[bb_word(c, a)], enc_f32(v)); else atomicMax(&enc[bb_word(c, a)], enc_f32(v));
  }
}
// grid descriptor of cloud c from its accumulated bounds, within the per-cloud cell budget (cell_base is the caller's scan)
__device__ inline GridDescB bb_make_desc(const uint32_t* __restrict__ enc, const uint32_t* __restrict__ off, uint32_t c, uint32_t budget, float cell0) {
  GridDescB d;
  d.g.ox = d.g.oy = d.g.oz = 0.f; d.g.inv_h = 1.f; d.g.nx = d.g.ny = d.g.nz = 1;
  d.pt_base = off[c];
  d.cell_base = 0;
  d.g.ncell = 1;   // an empty cloud: a 1-cell grid
  if (off[c + 1] != off[c]) {
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = dec_f32(enc[bb_word(c, a)]); mx[a] = dec_f32(enc[bb_word(c, 3 + a)]); }
    float h = cell0;
    for (;;) {
      d.g.inv_h = 1.0f / h;
      d.g.ox = mn[0]; d.g.oy = mn[1]; d.g.oz = mn[2];
      d.g.nx = (int)floorf((mx[0] - mn[0]) * d.g.inv_h) + 1;
      d.g.ny = (int)floorf((mx[1] - mn[1]) * d.g.inv_h) + 1;
      d.g.nz = (int)floorf((mx[2] - mn[2]) * d.g.inv_h) + 1;
      const unsigned long long nc = (unsigned long long)d.g.nx * d.g.ny * d.g.nz;
      if (nc <= budget) { d.g.ncell = (uint32_t)nc; break; }
      h *= 1.25f;
    }
  }
  return d;
}
// one thread per cloud (one workgroup, K <= 4096 in rounds of 1024): grid descriptors within the per-cloud cell budget, table bases by a scan
// (leaves the bounds accumulators reset for the next build: k_bb_init runs only when K grows)
__global__ __launch_bounds__(1024) void k_bb_setup(uint32_t* __restrict__ enc, const uint32_t* __restrict__ off, uint32_t K, GridDescB* __restrict__ desc,
                                                   uint32_t* __restrict__ scratch, uint32_t max_cells_total, float cell0) {
  __shared__ uint32_t lds[17];
  const uint32_t budget = max_cells_total / (K ? K : 1);
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < K; c0 += 1024) {
    const uint32_t c = c0 + threadIdx.x;
    GridDescB d;
    d.g.ox = d.g.oy = d.g.oz = 0.f; d.g.inv_h = 1.f; d.g.nx = d.g.ny = d.g.nz = 1; d.g.ncell = 0;
    d.pt_base = 0; d.cell_base = 0;
    if (c < K) {
      d = bb_make_desc(enc, off, c, budget, cell0);
#pragma unroll
      for (int a = 0; a < 6; a++) enc[bb_word(c, a)] = a < 3 ? 0xffffffffu : 0u;
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(c < K ? d.g.ncell : 0u, lds, tot);
    if (c < K) {
      d.cell_base = carry + ex;
      desc[c] = d;
    }
    carry += tot;
  }
  if (threadIdx.x == 0) {
    scratch[0] = carry + 1;
    scratch[2] = carry;
  }
}
// Round 6: for a handful of clouds (K <= BB_FUSE_MAXK: the odometry's 2 x streams of a chain) the set-up above is folded into the count —
// every workgroup derives the K descriptors from the bounds itself (a few dependent loads, in parallel over the workgroups), workgroup 0
// publishes them and the table size; the bounds accumulators are reset by the scatter, the build's last kernel.  One launch (and one
// dependent-launch gap) less in the tail of every odometry pass.
constexpr uint32_t BB_FUSE_MAXK = 64;
template <bool FUSED>
__global__ __launch_bounds__(256) void k_bb_count(const float4* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ off, uint32_t K,
                                                  GridDescB* __restrict__ desc, uint32_t* __restrict__ cell_of,
                                                  uint32_t* __restrict__ counts, uint32_t* __restrict__ rank_of, const uint32_t* __restrict__ enc,
                                                  uint32_t* __restrict__ scratch, uint32_t max_cells_total, float cell0) {
  __shared__ GridDescB s_desc[FUSED ? BB_FUSE_MAXK : 1];
  if (FUSED) {
    __shared__ uint32_t lds[17];
    const uint32_t cc = threadIdx.x;
    GridDescB d;
    d.g.ncell = 0;
    if (cc < K) d = bb_make_desc(enc, off, cc, max_cells_total / (K ? K : 1), cell0);
    uint32_t tot;
    const uint32_t ex = block_excl_scan(cc < K ? d.g.ncell : 0u, lds, tot);
    if (cc < K) {
      d.cell_base = ex;
      s_desc[cc] = d;
      if (blockIdx.x == 0) desc[cc] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { scratch[0] = tot + 1; scratch[2] = tot; }
    __syncthreads();
  }
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  uint32_t c = 0;
  if (active) {
    uint32_t lo = 0, hi = K;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (off[mid] <= i) lo = mid; else hi = mid;
    }
    const GridDescB d = FUSED ? s_desc[lo] : desc[lo];
    const float4 p = pts[i];
    int cx, cy, cz;
    cell_coords(d.g, p.x, p.y, p.z, cx, cy, cz);
    c = d.cell_base + ((uint32_t)cz * d.g.ny + cy) * d.g.nx + cx;
    cell_of[i] = c;
  }
  // the counter's value before a run's bump is where the run's points go inside their cell: the scatter needs no atomics of its own
  int head, len;
  wave_runs(c, active, head, len);
  uint32_t base = 0;
  if (active && head == (int)__lane_id()) base = atomicAdd(&counts[c], (uint32_t)len);
  base = __shfl(base, head, 64);
  if (active) rank_of[i] = base + (uint32_t)((int)__lane_id() - head);
}

__global__ __launch_bounds__(256) void k_bb_scatter(const float4* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ off, uint32_t K,
                                                    const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ rank_of,
                                                    const uint32_t* __restrict__ cell_start, float4* __restrict__ sorted, int pack_ring,
                                                    uint32_t* __restrict__ enc_reset) {
  if (enc_reset && blockIdx.x == 0 && threadIdx.x < K) {   // (the fused set-up: every workgroup of the count has read the bounds by now)
#pragma unroll
    for (int a = 0; a < 6; a++) enc_reset[bb_word(threadIdx.x, a)] = a < 3 ? 0xffffffffu : 0u;
  }
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0, hi = K;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  float4 p = pts[i];
  const uint32_t li = i - off[lo];      // index inside its own cloud
  // pack_ring (the odometry's clouds: .w = ring id): the top byte carries the ring so that a search can filter by ring without a
  // second gather; 255 = unknown (ring id or index too large for the packing — its user then falls back, odometry.hip)
  uint32_t w = li;
  if (pack_ring) {
    const int ring = (int)p.w;
    w = (li <= 0xffffffu && ring >= 0 && ring < 255) ? (((uint32_t)ring << 24) | li) : (0xff000000u | (li & 0xffffffu));
  }
  p.w = __uint_as_float(w);
  sorted[cell_start[cell_of[i]] + rank_of[i]] = p;
}

void SubMapIndexBatch::init(hipStream_t st) {
  st_ = st;
  scratch_.reserve(16);
  tile_sums_.reserve(SCAN_SCRATCH_WORDS);
  LX_HIP(hipMemsetAsync(tile_sums_.p, 0, sizeof(uint32_t) * tile_sums_.cap, st));
}

// the bounding-box accumulators can be reset long before the points exist (e.g. ahead of the iterations whose result the
// points depend on): build(..., prepared = true) then skips that launch
void SubMapIndexBatch::prepare(uint32_t K) {
  LX_REQUIRE(K >= 1 && K <= 4096, "too many clouds in one index batch");
  reset_bounds_(K);
}
void SubMapIndexBatch::reset_bounds_(uint32_t K) {
  if (K <= enc_ready_) return;   // every build's k_bb_setup leaves the accumulators of its K clouds reset
  const uint32_t cap = std::max<uint32_t>(K, 64u);
  enc_.reserve(((size_t)6 * cap + 6) * BB_STRIDE);   // (growing discards the contents: all of it is initialised below)
  hipLaunchKernelGGL(k_bb_init, dim3((6 * cap + 255) / 256), dim3(256), 0, st_, enc_.p, cap);
  enc_ready_ = cap;
}

void SubMapIndexBatch::build(const float4* d_pts, const uint32_t* h_off, uint32_t K, const uint32_t* d_off_ready, bool prepared, bool bounds_done) {
  LX_REQUIRE(K >= 1 && K <= 4096, "too many clouds in one index batch");
  const uint32_t n = h_off[K];
  d_off_.reserve(K + 2);
  d_desc_.reserve(K + 1);
  const uint32_t* d_off = d_off_ready;   // the caller may already hold the offsets on the device
  if (!d_off) {
    h_off_pin_.reserve(K + 2);
    memcpy(h_off_pin_.p, h_off, sizeof(uint32_t) * (K + 1));
    LX_HIP(hipMemcpyAsync(d_off_.p, h_off_pin_.p, sizeof(uint32_t) * (K + 1), hipMemcpyHostToDevice, st_));
    d_off = d_off_.p;
  }
  sorted_.reserve((size_t)n + 1);
  cell_of_.reserve((size_t)n + 1);
  rank_of_.reserve((size_t)n + 1);
  cell_start_.reserve((size_t)LX_MAX_CELLS + 2);
  if (!cursor_.p) {   // the cell counters: cleared once, kept clear by every build
    cursor_.reserve((size_t)LX_MAX_CELLS + 2);
    LX_HIP(hipMemsetAsync(cursor_.p, 0, sizeof(uint32_t) * cursor_.cap, st_));
  }
  (void)prepared;
  reset_bounds_(K);
  uint32_t max_len = 0;
  for (uint32_t c = 0; c < K; c++) max_len = std::max(max_len, h_off[c + 1] - h_off[c]);
  const uint32_t nbx = std::min<uint32_t>(std::max<uint32_t>((max_len + 255) / 256, 1u), 32u);
  if (!bounds_done) hipLaunchKernelGGL(k_bb_bbox, dim3(nbx, K), dim3(256), 0, st_, d_pts, d_off, enc_.p);
  const bool fused = n > 0 && K <= BB_FUSE_MAXK;   // (see k_bb_count: set-up folded into the count, bounds reset by the scatter)
  if (!fused) hipLaunchKernelGGL(k_bb_setup, dim3(1), dim3(1024), 0, st_, enc_.p, d_off, K, d_desc_.p, scratch_.p, LX_MAX_CELLS, cell_size);
  if (n) {
    if (fused) hipLaunchKernelGGL(k_bb_count<true>, dim3((n + 255) / 256), dim3(256), 0, st_, d_pts, n, d_off, K, d_desc_.p, cell_of_.p, cursor_.p, rank_of_.p,
                                  enc_.p, scratch_.p, LX_MAX_CELLS, cell_size);
    else hipLaunchKernelGGL(k_bb_count<false>, dim3((n + 255) / 256), dim3(256), 0, st_, d_pts, n, d_off, K, d_desc_.p, cell_of_.p, cursor_.p, rank_of_.p,
                            enc_.p, scratch_.p, LX_MAX_CELLS, cell_size);
  }
  // (the cell counters are cleared behind the scan: they are empty again when the next build starts)
  exclusive_scan_u32(cursor_.p, cell_start_.p, tile_sums_.p, scratch_.p + 2, scratch_.p + 1, LX_MAX_CELLS, st_, nullptr, cursor_.p);
  if (n) hipLaunchKernelGGL(k_bb_scatter, dim3((n + 255) / 256), dim3(256), 0, st_, d_pts, n, d_off, K, cell_of_.p, rank_of_.p, cell_start_.p, sorted_.p, pack_ring ? 1 : 0,
                            fused ? enc_.p : nullptr);
  LX_HIP(hipGetLastError());
}

// ----------------------------------------------------------------------------------------------------------------
// stack round trip + voxel keys
// ----------------------------------------------------------------------------------------------------------------
// seg_minmax: per segment min ix,iy,iz / max ix,iy,iz
// src: when given, segment k's points are read from src[k] (device-resident inputs are not gathered first)
__global__ __launch_bounds__(256) void k_stack(const float4* __restrict__ in, const float4* const* __restrict__ src, uint32_t n,
                                               const uint32_t* __restrict__ seg_off, uint32_t nseg, const Pose* __restrict__ poses,
                                               float inv_corner, float inv_surf, float4* __restrict__ stack, int* __restrict__ ijk,
                                               int* __restrict__ seg_minmax) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  uint32_t seg = 0;
  int ix = 0, iy = 0, iz = 0;
  if (active) {
    seg = vox_find_seg(seg_off, nseg, i);
    const Pose T = poses[seg >> 1];
    const float4 p = src ? src[seg][i - seg_off[seg]] : in[i];
    float x = p.x, y = p.y, z = p.z;
    to_map(T, x, y, z);
    to_be_mapped(T, x, y, z);
    stack[i] = make_float4(x, y, z, p.w);
    const float inv = (seg & 1) ? inv_surf : inv_corner;
    ix = (int)floorf(x * inv); iy = (int)floorf(y * inv); iz = (int)floorf(z * inv);
    ijk[3 * i] = ix; ijk[3 * i + 1] = iy; ijk[3 * i + 2] = iz;
  }
  seg_minmax_update(seg_minmax, active, seg, ix, iy, iz);
}

// ----------------------------------------------------------------------------------------------------------------
// Gauss-Newton iteration
// ----------------------------------------------------------------------------------------------------------------
// (also resets the voxel bounds of the sweep's two segments: one launch less in front of k_stack)
__global__ void k_pose_init(const float* __restrict__ guess, uint32_t ns, Pose* __restrict__ poses, SweepStats* __restrict__ stats,
                            int* __restrict__ seg_minmax, uint32_t* __restrict__ ticket, uint32_t* __restrict__ full_done) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  ticket[s] = 0u;   // k_gn_iter's per-sweep arrival counter
  full_done[s] = 0u;
  for (int k = 0; k < 12; k++) seg_minmax[12 * s + k] = (k % 6) < 3 ? 2147483647 : (-2147483647 - 1);
  Pose T;
  pose_set_angles(T, guess[6 * s], guess[6 * s + 1], guess[6 * s + 2]);
  T.tx = guess[6 * s + 3]; T.ty = guess[6 * s + 4]; T.tz = guess[6 * s + 5];
  poses[s] = T;
  SweepStats z = {0, 0, 0, 0, 0, 0, 0, 0};
  stats[s] = z;
}

// overwrite the poses with host-supplied ones (the IMU blend of transformUpdate happens on the host)
__global__ void k_pose_set(const float* __restrict__ p6, uint32_t ns, Pose* __restrict__ poses) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  Pose T;
  pose_set_angles(T, p6[6 * s], p6[6 * s + 1], p6[6 * s + 2]);
  T.tx = p6[6 * s + 3]; T.ty = p6[6 * s + 4]; T.tz = p6[6 * s + 5];
  poses[s] = T;
}


// ----------------------------------------------------------------------------------------------------------------
// neighbour search helpers (used by knn5_group below)
// ----------------------------------------------------------------------------------------------------------------
constexpr float KNN_COVER2 = 1.05f * 1.05f * 0.9999f;   // every map point within this squared radius has been visited

// Branch-free insertion into a lane's ascending top-6.  A candidate's rank key is (bits(d2) << 32) | original index:
// d2 >= +0 and finite, so the unsigned order of the float bits is the order of the values and one 64-bit compare is the
// strict (d2, index) order of a sequential scan.
__device__ __forceinline__ void knn_insert6(unsigned long long (&bk)[6], uint32_t (&bp)[6], unsigned long long key, uint32_t pos) {
  bool lt[6];
#pragma unroll
  for (int j = 0; j < 6; j++) lt[j] = key < bk[j];
#pragma unroll
  for (int j = 5; j > 0; j--) {
    bk[j] = lt[j - 1] ? bk[j - 1] : (lt[j] ? key : bk[j]);
    bp[j] = lt[j - 1] ? bp[j - 1] : (lt[j] ? pos : bp[j]);
  }
  bk[0] = lt[0] ? key : bk[0];
  bp[0] = lt[0] ? pos : bp[0];
}
constexpr unsigned long long KNN_NONE = 0x7f7fffffffffffffull;   // (FLT_MAX, 0xffffffff): empty slot

// pop the group's six best (destructive on the lane lists); every lane of the group ends up with the same winners
template <int LPQ>
__device__ __forceinline__ void knn_pop6(unsigned long long (&bk)[6], uint32_t (&bp)[6], int gl, unsigned long long (&wk)[6], uint32_t (&win)[6]) {
#pragma unroll
  for (int k = 0; k < 6; k++) {
    unsigned long long key = bk[0];
    uint32_t pp = bp[0];
    int owner = gl;
#pragma unroll
    for (int m = LPQ / 2; m > 0; m >>= 1) {
      const unsigned long long ok = __shfl_xor(key, m, LPQ);
      const uint32_t op = __shfl_xor(pp, m, LPQ);
      const int oo = __shfl_xor(owner, m, LPQ);
      if (ok < key) { key = ok; pp = op; owner = oo; }   // distinct points have distinct keys; empty slots tie harmlessly
    }
    if (LPQ > 1 || k < 5) {
      if (owner == gl) {   // the winner's lane advances its list
#pragma unroll
        for (int j = 0; j < 5; j++) { bk[j] = bk[j + 1]; bp[j] = bp[j + 1]; }
        bk[5] = KNN_NONE;
      }
    }
    wk[k] = key;
    win[k] = pp;
  }
}

struct Row {
  float a[6];
  float b;
  bool sel;
};

// corner query: BasicLaserMapping.cpp:667-751.  JACOBI: the 3x3 eigen-decomposition by the cyclic Jacobi iteration the oracle uses
// for Eigen's solver (instruction for instruction) instead of the closed form (dev_math.hpp)
template <bool JACOBI>
__device__ __forceinline__ void corner_row(const Pose& T, const float4 po, const float4* __restrict__ pts, const uint32_t (&bp)[5], float& cx_,
                                  float& cy_, float& cz_, float& ci_, bool& sel) {
  sel = false;
  float x0 = po.x, y0 = po.y, z0 = po.z;
  to_map(T, x0, y0, z0);
  if (bp[4] == 0xffffffffu) return;
  float4 nb[5];
#pragma unroll
  for (int j = 0; j < 5; j++) nb[j] = pts[bp[j]];
  float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
  for (int j = 0; j < 5; j++) { vx += nb[j].x; vy += nb[j].y; vz += nb[j].z; }
  vx /= 5.0f; vy /= 5.0f; vz /= 5.0f;
  float a00 = 0.f, a10 = 0.f, a20 = 0.f, a11 = 0.f, a21 = 0.f, a22 = 0.f;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const float ax = nb[j].x - vx, ay = nb[j].y - vy, az = nb[j].z - vz;
    a00 += ax * ax; a10 += ax * ay; a20 += ax * az; a11 += ay * ay; a21 += ay * az; a22 += az * az;
  }
  a00 /= 5.0f; a10 /= 5.0f; a20 /= 5.0f; a11 /= 5.0f; a21 /= 5.0f; a22 /= 5.0f;
  float w0, w1, w2, ex, ey, ez;
  if (JACOBI) eig3_sym(a00, a10, a11, a20, a21, a22, w0, w1, w2, ex, ey, ez);
  else eig3_sym_direct(a00, a10, a11, a20, a21, a22, w0, w1, w2, ex, ey, ez);
  if (!(w2 > 3 * w1)) return;
  const float x1 = (float)(vx + 0.1 * ex), y1 = (float)(vy + 0.1 * ey), z1 = (float)(vz + 0.1 * ez);
  const float x2 = (float)(vx - 0.1 * ex), y2 = (float)(vy - 0.1 * ey), z2 = (float)(vz - 0.1 * ez);
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
  const float s = 1 - 0.9f * fabsf(ld2);
  cx_ = s * la; cy_ = s * lb; cz_ = s * lc; ci_ = s * ld2;
  sel = ((double)s > 0.1);
}

// surf query: BasicLaserMapping.cpp:756-816
__device__ __forceinline__ void surf_row(const Pose& T, const float4 po, const float4* __restrict__ pts, const uint32_t (&bp)[5], float& cx_,
                                float& cy_, float& cz_, float& ci_, bool& sel) {
  sel = false;
  float x0 = po.x, y0 = po.y, z0 = po.z;
  to_map(T, x0, y0, z0);
  if (bp[4] == 0xffffffffu) return;
  float A[5][3], b[5], X[3];
  float4 nb[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    nb[j] = pts[bp[j]];
    A[j][0] = nb[j].x; A[j][1] = nb[j].y; A[j][2] = nb[j].z;
    b[j] = -1.f;
  }
  qr_solve<5, 3>(A, b, X);
  float pa = X[0], pb = X[1], pc = X[2], pd = 1;
  const float ps = sqrtf(pa * pa + pb * pb + pc * pc);
  pa /= ps; pb /= ps; pc /= ps; pd /= ps;
  bool planeValid = true;
#pragma unroll
  for (int j = 0; j < 5; j++)
    if ((double)fabsf(pa * nb[j].x + pb * nb[j].y + pc * nb[j].z + pd) > 0.2) planeValid = false;
  if (!planeValid) return;
  const float pd2 = pa * x0 + pb * y0 + pc * z0 + pd;
  const float s = 1 - 0.9f * fabsf(pd2) / sqrtf(sqrtf(x0 * x0 + y0 * y0 + z0 * z0));
  cx_ = s * pa; cy_ = s * pb; cz_ = s * pc; ci_ = s * pd2;
  sel = ((double)s > 0.1);
}

// grid = (blocks per sweep, sweeps).  partials[(s*nblk + b)*LX_NSUM + k]
// the update step of one sweep, run by the LAST workgroup of k_residual to finish (all 256 threads): fixed-order
// (deterministic) reduction of the block partials — 9 groups of 28 threads each walk every 9th block, then the 9 group
// sums are added in order — then wave 0 solves and thread 0 updates the pose
constexpr int LX_SOLVE_GROUPS = 9;
#ifndef LX_SOLVE_MLP
#define LX_SOLVE_MLP 18
#endif
#ifdef LOAMX_PROF_GN
__device__ unsigned long long g_solve_ts[16];
#define SOLVE_TS(k) do { if (s == 0 && iter == 0 && threadIdx.x == 0) g_solve_ts[k] = wall_clock64(); } while (0)
#else
#define SOLVE_TS(k) do { } while (0)
#endif
// pose and statistics of one sweep into the pinned mirrors, the `done` flag last (behind the other stores) and a check
// word over what the host is about to read: Registrar::run_iterations polls these while later launches are still in flight
__device__ __forceinline__ void mirror_to_host(SweepStats* hs, Pose* hp, const SweepStats& st, const Pose& T) {
  SweepStats m = st;
  m.done = 0;
  m.pad1 = mirror_check_word(st, T);
  *hp = T;
  *hs = m;
  if (!st.done) return;
  xchg_stores_done();
  __hip_atomic_store(&hs->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void solve_sweep(uint32_t s, const uint32_t* __restrict__ ds_off, Pose* __restrict__ poses, SweepStats* __restrict__ stats,
                                   float* __restrict__ matP, const double* partials, uint32_t nblk, uint32_t nact, int iter,
                                   float delta_t_abort, float delta_r_abort, SweepStats* host_stats, Pose* host_poses) {
  __shared__ double gsum[LX_SOLVE_GROUPS][LX_NSUM];
  __shared__ double sums[LX_NSUM];
  __shared__ float ws[216];
  __shared__ float AtA[36], AtB[6], X[6], X2[6], trig[6];
  __shared__ Pose sP;
  __shared__ SweepStats sSt;
  __shared__ int s_cert;
  const int tid = (int)threadIdx.x;
  SOLVE_TS(0);
  if (tid == LX_RES_THREADS - 1) {   // (a lane of the last wave: the loads overlap with the partial sums below)
    sP = poses[s];
    sSt = stats[s];
  }
  if (tid < LX_SOLVE_GROUPS * LX_NSUM) {
    const uint32_t g = (uint32_t)tid / LX_NSUM, t = (uint32_t)tid % LX_NSUM;
    double x = 0.0;
    // LX_SOLVE_MLP agent-scope loads in flight (the partials come from the other XCDs' workgroups; see dev_math.hpp "exchange"), added
    // in tile order.  Unconditional loads with a clamped tile index: the conditional ones of round 3 were compiled into one branch +
    // wait per load — 34 dependent round trips, 9-12 of solve_sweep's ~20 us (in-kernel time stamps)
    for (uint32_t b0 = g; b0 < nact; b0 += LX_SOLVE_MLP * LX_SOLVE_GROUPS) {
      double v[LX_SOLVE_MLP];
#pragma unroll
      for (int u = 0; u < LX_SOLVE_MLP; u++) {
        const uint32_t b = min(b0 + u * LX_SOLVE_GROUPS, nact - 1u);
        v[u] = xchg_load_nowait(&partials[((size_t)s * nblk + b) * LX_NSUM + t]);
      }
      xchg_loads_done();
#pragma unroll
      for (int u = 0; u < LX_SOLVE_MLP; u++) {
        xchg_loaded(v[u]);
        x += (b0 + u * LX_SOLVE_GROUPS < nact) ? v[u] : 0.0;
      }
    }
    gsum[g][t] = x;
  }
  __syncthreads();
  SOLVE_TS(1);
  if (tid < LX_NSUM) {
    double x = 0.0;
#pragma unroll
    for (int g = 0; g < LX_SOLVE_GROUPS; g++) x += gsum[g][tid];
    sums[tid] = x;
    // scatter straight into the symmetric 6x6 / right-hand side (sum index t -> (i, j) of the upper triangle)
    if (tid < 21) {
      int i = 0, rem = tid;
      while (rem >= 6 - i) { rem -= 6 - i; i++; }
      const int j = i + rem;
      AtA[i * 6 + j] = AtA[j * 6 + i] = (float)x;
    } else if (tid < 27) {
      AtB[tid - 21] = (float)x;
    }
  }
  __syncthreads();
  const int sel_rows = (int)sums[27];   // block-uniform
  if (sel_rows < 50) {   // BasicLaserMapping.cpp:826-828: the iteration is burnt, pose untouched
    if (tid == 0) {
      SweepStats st = sSt;
      st.iterations = iter + 1;
      st.sel = sel_rows;
      st.corner_q = (int)(ds_off[2 * s + 1] - ds_off[2 * s]);
      st.surf_q = (int)(ds_off[2 * s + 2] - ds_off[2 * s + 1]);
      stats[s] = st;
      if (host_stats) mirror_to_host(host_stats + s, host_poses + s, st, sP);
    }
    return;
  }
  SOLVE_TS(2);
  if (tid < 64) qr_solve6_coop(AtA, AtB, X);   // wave 0, all lanes
  else if (tid == 64 && iter == 0) s_cert = certainly_not_degenerate(AtA, 100.f) ? 1 : 0;   // wave 1, meanwhile: the non-degeneracy certificate
  __syncthreads();
  SOLVE_TS(3);
  float* P = matP + 36 * s;
  if (tid == 0) {
    if (iter == 0) sSt.degenerate = s_cert ? 0 : (degeneracy_projector_full(AtA, 100.f, P, ws) ? 1 : 0);
    if (sSt.degenerate) {
      for (int r = 0; r < 6; r++) X2[r] = X[r];
      for (int r = 0; r < 6; r++) {
        float acc = 0.f;
        for (int c = 0; c < 6; c++) acc += P[r * 6 + c] * X2[c];
        X[r] = acc;
      }
    }
  }
  __syncthreads();
  SOLVE_TS(4);
  if (tid < 6) {   // sin / cos of the new angles, one per lane, double then rounded (pose_set_angles)
    const float ang = (tid < 2 ? sP.rx : (tid < 4 ? sP.ry : sP.rz)) + X[tid >> 1];
    trig[tid] = (float)((tid & 1) ? cos((double)ang) : sin((double)ang));
  }
  __syncthreads();
  SOLVE_TS(5);
  if (tid != 0) return;
  SweepStats st = sSt;
  st.iterations = iter + 1;
  st.sel = sel_rows;
  st.corner_q = (int)(ds_off[2 * s + 1] - ds_off[2 * s]);
  st.surf_q = (int)(ds_off[2 * s + 2] - ds_off[2 * s + 1]);
  Pose T = sP;
  T.rx = T.rx + X[0]; T.ry = T.ry + X[1]; T.rz = T.rz + X[2];
  T.srx = trig[0]; T.crx = trig[1]; T.sry = trig[2]; T.cry = trig[3]; T.srz = trig[4]; T.crz = trig[5];
  T.tx += X[3]; T.ty += X[4]; T.tz += X[5];
  poses[s] = T;
  // rad2deg(float) goes through double (math_utils.h:30-33)
  const float d0 = (float)(X[0] * 180.0 / M_PI), d1 = (float)(X[1] * 180.0 / M_PI), d2 = (float)(X[2] * 180.0 / M_PI);
  const float deltaR = (float)sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);
  const float t0 = X[3] * 100, t1 = X[4] * 100, t2 = X[5] * 100;
  const float deltaT = (float)sqrt((double)t0 * t0 + (double)t1 * t1 + (double)t2 * t2);
  if (deltaR < delta_r_abort && deltaT < delta_t_abort) st.done = 1;
  stats[s] = st;
  // mirror in host-visible (pinned, mapped) memory: a blocking caller polls the flags and reads the poses without a
  // stream sync and without two more copies on the stream
  if (host_stats) mirror_to_host(host_stats + s, host_poses + s, st, T);
  SOLVE_TS(6);
}

// ----------------------------------------------------------------------------------------------------------------
// k_gn_iter: ONE Gauss-Newton iteration of every sweep of the batch in one launch (BasicLaserMapping.cpp:660-923) —
// neighbour search, edge / plane fit, Jacobian row, normal equations and the update step fused; nothing per query goes
// to HBM between them (round 1 wrote five neighbour positions + a query state per query and re-gathered them).
//
// grid = (tiles rounded up to 8, sweeps), 256 threads.  A sweep's down-sampled queries are cut into tiles of 64 (corner
// tiles first, then surf tiles — the row order of the reference's A; a tile never mixes the two, so the sub-map, the grid
// descriptor and the fit are uniform per workgroup); XCD x % 8 gets a contiguous eighth of the voxel-ordered tiles, so its
// private L2 only has to hold that part of the map.  Per workgroup:
//   search   four lanes per query (knn5_group below): pointAssociateToMap, exact 5-NN inside the 1 m gate; the winners'
//            positions go to LDS
//   rows     one lane per query (wave 0): 3x3 Jacobi eigen edge fit or 5x3 pivoted-QR plane fit, weight, Jacobian row
//   sums     the tile's 28 sums: float products through an LDS-transposed table, fixed-order accumulation in double
//   update   the LAST workgroup of a sweep to deliver its sums adds the tile partials in a fixed order and runs the update
//            (solve_sweep: 6x6 pivoted QR on wave 0, iteration-0 degeneracy projector with the row-zeroing quirk :869-899,
//            pose update :907-912, stop test :914-922) — one kernel boundary per iteration; `done` turns later launches into
//            no-ops.
// The sums depend on the tiling (64 queries per tile, tiles in order) only — not on the batch composition.
// A persistent variant (all iterations in one launch, per-sweep ticket barrier) was built and measured first: it must keep
// every workgroup of a sweep resident, which caps it at ~2 waves per SIMD, and at that occupancy the search — latency- and
// issue-bound — ran three times slower than this kernel's 4-5 waves per SIMD (405 us against ~150 us for 3 iterations of
// 8 HDL-64E sweeps); see DESIGN.md §3.
//
// knn5_group — exact five nearest map points by (squared distance, original index):
//   trip 1   the boundaries of all 27 cells of the query's 3x3x3 neighbourhood: 9 rows (x is the fastest cell axis, so a
//            row's cells cx-1..cx+1 are one contiguous run) x 4 cell_start entries = 36 independent 4-byte loads dealt over
//            the group's lanes and exchanged through LDS.  Rows are addressed RELATIVE to the query (centre / nearer side /
//            farther side per axis).
//   trip 2   phase 1: the 2x2x2 block of cells nearest to the query (own cell + the neighbours across the nearer faces;
//            it contains the ball of radius h/2): 4 runs, candidates enumerated flat and strided over the lanes (adjacent
//            lanes read adjacent points), 4 independent 16-byte loads in flight per lane, branch-free insertion into
//            per-lane sorted top-6 lists on 64-bit rank keys, then a 4-lane arg-min butterfly pops the group's six best.
//   trip 3   phase 2: whatever else of the 27 cells lies within the sixth-best distance so far (or the covered radius):
//            the 4 far cells of the phase-1 rows + the 5 remaining rows clipped in x, pruned with a conservative gap
//            expression (margins err towards visiting).  Usually empty or one cell.
// The result is the strict (d2, index) order of a sequential scan over the whole map, independent of the visiting order
// and of the (nondeterministic) slot order inside a cell.
// ----------------------------------------------------------------------------------------------------------------
#ifndef KNN_LPQ
#define KNN_LPQ 4   // lanes per query in knn5_group
#endif
#ifndef KNN_MLP
#define KNN_MLP 4   // independent 16-byte loads in flight per lane
#endif
// flat, strided scan of NR concatenated runs by the LPQ lanes of a group: lane gl takes candidates gl, gl + LPQ, ...
// (adjacent lanes read adjacent points: a group's load covers one or two cache lines), MLP loads in flight per lane
template <int NR, int MLP, int LPQ>
__device__ __forceinline__ void knn_scan_group(const float4* __restrict__ pts, const uint32_t (&beg)[NR], const uint32_t (&len)[NR], float qx, float qy,
                                      float qz, int gl, unsigned long long (&bk)[6], uint32_t (&bp)[6]) {
  uint32_t E[NR], O[NR];   // cumulative candidate count after run k; position = candidate number + O[k] inside run k
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < NR; k++) {
    O[k] = beg[k] - acc;
    acc += len[k];
    E[k] = acc;
  }
  const uint32_t total = acc;
  for (uint32_t c0 = (uint32_t)gl; c0 < total; c0 += MLP * LPQ) {
    uint32_t pos[MLP];
    float4 p[MLP];
#pragma unroll
    for (int u = 0; u < MLP; u++) {
      const uint32_t cc = c0 + u * LPQ;
      uint32_t o = O[NR - 1];
#pragma unroll
      for (int k = NR - 2; k >= 0; k--) o = cc < E[k] ? O[k] : o;
      pos[u] = cc < total ? cc + o : 0u;   // (slot 0 exists: the index is never empty here)
    }
#pragma unroll
    for (int u = 0; u < MLP; u++) p[u] = pts[pos[u]];
#pragma unroll
    for (int u = 0; u < MLP; u++) {
      const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
      const float d2 = dx * dx + dy * dy + dz * dz;   // x -> y -> z accumulation (nanoflann.hpp:372-379)
      const bool ok = (c0 + u * LPQ) < total && d2 < KNN_COVER2;
      const unsigned long long key = ok ? ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(p[u].w) : KNN_NONE;
      knn_insert6(bk, bp, key, pos[u]);
    }
  }
}

// distance (cell units) from coordinate f inside cell c to the neighbouring cell on side d (-1 / +1), shrunk by 1e-4
// relative + 1e-5 so that rounding can only make the visit larger
__device__ __forceinline__ float knn_gap(float f, int c, int d) {
  const float v = d > 0 ? (float)(c + 1) - f : f - (float)c;
  const float w = v * 0.9999f - 1e-5f;
  return w > 0.f ? w : 0.f;
}

// The search of ONE query by the LPQ lanes of a group (all in one wave; every lane holds the same qx, qy, qz).
// tab: the group's column of an LDS table [36][QB] (entry e of this group at tab[e * QB]).
// out (all lanes): wk[0..5] ascending rank keys (KNN_NONE = empty), win[] = positions in the cell-sorted array
template <int LPQ, int QB>
__device__ __forceinline__ void knn5_group(const GridDesc& g, const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start, float qx, float qy,
                                  float qz, int gl, uint32_t* tab, unsigned long long (&wk)[6], uint32_t (&win)[6]) {
#pragma unroll
  for (int j = 0; j < 6; j++) { wk[j] = KNN_NONE; win[j] = 0u; }
  const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
  const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  // a query farther than two cells outside the grid has no map point within the gate (and the casts below stay in range)
  if (!(flx >= -2.f && flx <= (float)(g.nx + 1) && fly >= -2.f && fly <= (float)(g.ny + 1) && flz >= -2.f && flz <= (float)(g.nz + 1))) return;   // group-uniform
  const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
  const int sx = (fx - flx) < 0.5f ? -1 : 1, sy = (fy - fly) < 0.5f ? -1 : 1, sz = (fz - flz) < 0.5f ? -1 : 1;
  // ---- trip 1: the 36 table entries, dealt over the group's lanes.  Entry e = 4 r + k: cell_start of cell (cx - 1 + k) in
  // row r = 3 a + b; a (z) and b (y): 0 centre, 1 nearer side, 2 farther side.  Out-of-range x is clamped onto the row's
  // ends (those cells then come out empty), rows outside the grid are empty.
  {
    uint32_t v[(36 + LPQ - 1) / LPQ];
#pragma unroll
    for (int i = 0; i < (36 + LPQ - 1) / LPQ; i++) {
      const int e = gl + LPQ * i;
      const int r = e >> 2, k = e & 3, a = r / 3, b = r - 3 * a;
      const int z = cz + (a == 1 ? sz : (a == 2 ? -sz : 0)), y = cy + (b == 1 ? sy : (b == 2 ? -sy : 0));
      const bool in = e < 36 && z >= 0 && z < g.nz && y >= 0 && y < g.ny;
      v[i] = in ? cell_start[((uint32_t)z * g.ny + y) * g.nx + clampi(cx - 1 + k, 0, g.nx)] : 0u;
    }
#pragma unroll
    for (int i = 0; i < (36 + LPQ - 1) / LPQ; i++) {
      const int e = gl + LPQ * i;
      if (e < 36) tab[e * QB] = v[i];
    }
  }
  __builtin_amdgcn_wave_barrier();   // the group's lanes are in one wave: the LDS writes above are visible below
  unsigned long long bk[6];
  uint32_t bp[6];
#pragma unroll
  for (int j = 0; j < 6; j++) { bk[j] = KNN_NONE; bp[j] = 0u; }
  const int kn = sx < 0 ? 0 : 1;   // column of the first of the two cells {cx, cx + sx}; the far cell is column 2 resp. 0
  // ---- phase 1: rows (a, b) in {0,1}^2 (r = 0, 1, 3, 4), cells cx and cx + sx
  {
    uint32_t beg[4], len[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = i < 2 ? i : i + 1;
      const uint32_t lo = tab[(4 * r + kn) * QB], hi = tab[(4 * r + kn + 2) * QB];
      beg[i] = lo;
      len[i] = hi - lo;
    }
    knn_scan_group<4, KNN_MLP, LPQ>(pts, beg, len, qx, qy, qz, gl, bk, bp);
  }
  knn_pop6<LPQ>(bk, bp, gl, wk, win);
  // ---- phase 2: everything else within the radius that still matters (the sixth-best so far, or the covered radius)
  {
    const float rad2 = wk[5] != KNN_NONE ? __uint_as_float((uint32_t)(wk[5] >> 32)) : KNN_COVER2;
    const float r2c = rad2 * g.inv_h * g.inv_h;   // in cell units
    const float gx_near = knn_gap(fx, cx, sx), gx_far = knn_gap(fx, cx, -sx);
    const float gxl = sx < 0 ? gx_near : gx_far, gxr = sx < 0 ? gx_far : gx_near;
    const float gy[3] = {0.f, knn_gap(fy, cy, sy), knn_gap(fy, cy, -sy)};
    const float gz[3] = {0.f, knn_gap(fz, cz, sz), knn_gap(fz, cz, -sz)};
    uint32_t beg[9], len[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const int r = 3 * a + b;
        const float gyz = gy[b] * gy[b] + gz[a] * gz[a];
        uint32_t lo = 0u, hi = 0u;
        if (a < 2 && b < 2) {
          // a phase-1 row: only its far cell (cx - sx) is left
          if (gyz + gx_far * gx_far < r2c) { lo = tab[(4 * r + 2 - 2 * kn) * QB]; hi = tab[(4 * r + 3 - 2 * kn) * QB]; }
        } else if (gyz < r2c) {
          lo = tab[(4 * r + ((gyz + gxl * gxl < r2c) ? 0 : 1)) * QB];
          hi = tab[(4 * r + ((gyz + gxr * gxr < r2c) ? 3 : 2)) * QB];
        }
        beg[r] = lo;
        len[r] = hi - lo;
      }
    // re-seed the lists with the winners so far (lane 0 holds them, in order) and scan what is left
#pragma unroll
    for (int k = 0; k < 6; k++) { bk[k] = gl == 0 ? wk[k] : KNN_NONE; bp[k] = win[k]; }
    knn_scan_group<9, KNN_MLP, LPQ>(pts, beg, len, qx, qy, qz, gl, bk, bp);
  }
  knn_pop6<LPQ>(bk, bp, gl, wk, win);
}

struct GnArgs {
  const float4* ds_pts;
  const uint32_t* ds_off;
  Pose* poses;
  SweepStats* stats;
  const GridDesc* cdesc;
  const float4* cpts;
  const uint32_t* cstart;
  const GridDesc* sdesc;
  const float4* spts;
  const uint32_t* sstart;
  double* partials;      // [sweep][nblk][LX_NSUM]
  uint32_t* arrive;      // per sweep: workgroups that have delivered their sums (zero between launches)
  float* matP;           // 36 per sweep
  SweepStats* host_stats;
  Pose* host_poses;
  uint32_t nblk;         // tiles the partials array reserves per sweep
  unsigned long long* dbg;   // LOAMX_PROF_GN: per workgroup of sweep 0: 8 wall-clock stamps
  int iter;
  float delta_t_abort, delta_r_abort;
  const uint32_t* skip_word;   // a bucketed voxel stage that gave up (fail word == skip_value) left no valid query offsets: the launch does nothing
  uint32_t skip_value;
};

constexpr int GN_TILE = LX_RES_THREADS / KNN_LPQ;   // queries per workgroup
#ifdef LOAMX_PROF_GN
#define GN_TS(k) do { if (blockIdx.y == 0 && tid == 0 && A.iter == 0) A.dbg[(size_t)tile * 8 + (k)] = wall_clock64(); } while (0)
#else
#define GN_TS(k) do { } while (0)
#endif

// waves per SIMD the register allocation of k_gn_iter aims at: 6 = 80 VGPRs without scratch (the compiler's own choice is 83 = 5 waves);
// with 16 KB of LDS per workgroup six to seven 4-wave workgroups share a CU
#ifndef GN_WAVES
#define GN_WAVES 6
#endif
#define GN_KERNEL k_gn_iter
#define GN_JACOBI false
#include "gn_iter_kernel.inc"
#undef GN_KERNEL
#undef GN_JACOBI
#ifdef LOAMX_DIAG   // the oracle's cyclic Jacobi in the edge fit (bit-faithful, slower): a diagnostic build's LOAMX_EIG_JACOBI=1 selects it
#define GN_KERNEL k_gn_iter_jacobi
#define GN_JACOBI true
#include "gn_iter_kernel.inc"
#undef GN_KERNEL
#undef GN_JACOBI
#endif

// debug / parity hook: the 5-NN search of k_gn_iter for arbitrary map-frame query points (loamx_batch_knn_probe)
__global__ __launch_bounds__(256) void k_knn_probe(const float4* __restrict__ queries, uint32_t n, const GridDesc* __restrict__ desc,
                                                   const float4* __restrict__ pts, const uint32_t* __restrict__ cell_start,
                                                   uint32_t* __restrict__ idx5, float* __restrict__ d2_5) {
  constexpr int QB = 256 / KNN_LPQ;
  __shared__ uint32_t tab[36 * QB];
  const int grp = threadIdx.x / KNN_LPQ, gl = threadIdx.x % KNN_LPQ;
  const uint32_t q = blockIdx.x * QB + grp;
  if (q >= n) return;   // group-uniform (no workgroup-level synchronisation below)
  const float4 p = queries[q];
  unsigned long long wk[6];
  uint32_t win[6];
  knn5_group<KNN_LPQ, QB>(*desc, pts, cell_start, p.x, p.y, p.z, gl, tab + grp, wk, win);
  if (gl != 0) return;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    idx5[5 * (size_t)q + j] = wk[j] != KNN_NONE ? (uint32_t)(wk[j] & 0xffffffffull) : 0xffffffffu;
    d2_5[5 * (size_t)q + j] = wk[j] != KNN_NONE ? __uint_as_float((uint32_t)(wk[j] >> 32)) : FLT_MAX;
  }
}

// mode 0: every sweep.  mode 1 (enqueued together with the Gauss-Newton launches, before the host has seen their flags): only the
// sweeps that have converged, marked in full_done with this launch's tag; mode 2 (after further iterations): the sweeps not marked yet.
// skip_word / skip_value: a voxel stage that gave up leaves the clouds untouched for the repeated run.
__global__ __launch_bounds__(256) void k_transform_full(float4* __restrict__ full, uint32_t n, const uint32_t* __restrict__ full_off,
                                                        uint32_t ns, const Pose* __restrict__ poses, const SweepStats* __restrict__ stats,
                                                        uint32_t* __restrict__ full_done, int mode, uint32_t tag,
                                                        const uint32_t* __restrict__ skip_word, uint32_t skip_value) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (skip_word && *skip_word == skip_value) return;
  const uint32_t s = vox_find_seg(full_off, ns, i);
  if (mode == 1) {
    if (!stats[s].done) return;
    if (i == full_off[s]) full_done[s] = tag;
  } else if (mode == 2) {
    if (full_done[s] != 0u) return;   // (nothing writes full_done in this mode)
  }
  const Pose T = poses[s];
  float4 p = full[i];
  to_map(T, p.x, p.y, p.z);
  full[i] = p;
}

// copy nseg source ranges into one destination array: segment k -> dst[off[k] .. off[k+1])
__global__ __launch_bounds__(256) void k_gather_segments(float4* __restrict__ dst, const uint32_t* __restrict__ off, uint32_t nseg,
                                                         const float4* const* __restrict__ src, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = vox_find_seg(off, nseg, i);
  dst[i] = src[k][i - off[k]];
}

// One sweep (the sequential entry points): its full-resolution cloud is ONE range, and the parameter block of the call (guess, offsets,
// source pointers: 72 bytes) fits the kernel's arguments — thread 0..17 of the first workgroup write it for the kernels behind this one,
// the launch replaces the block's copy command AND k_gather_segments (one dependent launch less at the head of every sweep's mapping)
struct ParamBlock1 { uint32_t w[18]; };
__global__ __launch_bounds__(256) void k_gather_one(float4* __restrict__ dst, const float4* __restrict__ src, uint32_t n, ParamBlock1 blk,
                                                    uint32_t* __restrict__ blk_dst) {
  if (blockIdx.x == 0 && threadIdx.x < 18) blk_dst[threadIdx.x] = blk.w[threadIdx.x];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// ----------------------------------------------------------------------------------------------------------------
// Registrar (host)
// ----------------------------------------------------------------------------------------------------------------
Registrar::Registrar(int device, uint32_t max_sweeps) : device_(device), max_sweeps_(max_sweeps) {
  select_device(device);
  st_ = create_stream(env_priority("LOAMX_PRIO_REG", +1), 0, /*part=*/1);
  corner_index.init(st_);
  surf_index.init(st_);
  poses_.reserve(max_sweeps);
  stats_.reserve(max_sweeps);
  matP_.reserve((size_t)36 * max_sweeps);
  arrive_.reserve(max_sweeps);
  LX_HIP(hipMemsetAsync(arrive_.p, 0, sizeof(uint32_t) * max_sweeps, st_));
  guess_.reserve((size_t)6 * max_sweeps);
  h_guess_.reserve((size_t)64 * max_sweeps + 64);
  seg_off_.reserve(2 * max_sweeps + 2);
  full_off_.reserve(max_sweeps + 2);
  ds_off_.reserve(2 * max_sweeps + 2);
  vox_.init(st_);
  vb_.init(st_);
  vb_disabled_ = getenv("LOAMX_VOX_LEGACY") != nullptr;
  jacobi_eig_ = diag_env("LOAMX_EIG_JACOBI") != nullptr;
  full_done_.reserve(max_sweeps);
  h_stats_.reserve(max_sweeps);
  h_poses_.reserve(max_sweeps);
  ev_.resize(2 + 2 * 64);
  for (auto& e : ev_) LX_HIP(hipEventCreate(&e));
}

Registrar::~Registrar() {
  for (auto& e : ev_) (void)hipEventDestroy(e);
  if (ev_build_) (void)hipEventDestroy(ev_build_);
  if (ev_swap_) (void)hipEventDestroy(ev_swap_);
  if (ev_look_) (void)hipEventDestroy(ev_look_);
  if (st_build_) { (void)hipStreamSynchronize(st_build_); (void)hipStreamDestroy(st_build_); }
  if (st_) (void)hipStreamDestroy(st_);
}

void Registrar::set_submap_host(const loamx_cloud* corner, const loamx_cloud* surf) {
  check_cloud(corner, false);
  check_cloud(surf, false);
  LX_HIP(hipSetDevice(device_));
  PinBuf<float4> hc, hs;
  hc.reserve(corner->count);
  hs.reserve(surf->count);
  pack_cloud(corner, hc.p);
  pack_cloud(surf, hs.p);
  own_corner_.reserve(corner->count);
  own_surf_.reserve(surf->count);
  LX_HIP(hipMemcpyAsync(own_corner_.p, hc.p, sizeof(float4) * corner->count, hipMemcpyHostToDevice, st_));
  LX_HIP(hipMemcpyAsync(own_surf_.p, hs.p, sizeof(float4) * surf->count, hipMemcpyHostToDevice, st_));
  corner_index.build(own_corner_.p, corner->count);
  surf_index.build(own_surf_.p, surf->count);
  LX_HIP(hipStreamSynchronize(st_));
}

void Registrar::set_submap_device(const float4* d_corner, uint32_t nc, const float4* d_surf, uint32_t ns, bool sync) {
  LX_HIP(hipSetDevice(device_));
  corner_index.build(d_corner, nc);
  surf_index.build(d_surf, ns);
  if (sync) LX_HIP(hipStreamSynchronize(st_));
}

// the two indices built side by side: the corner index on a stream of the caller's (ordered by the caller: the registration's stream
// has to wait for what this enqueues there before run_async()), the surf index on the registration's stream
void Registrar::set_submap_device_split(const float4* d_corner, uint32_t nc, hipStream_t corner_stream, const float4* d_surf, uint32_t ns, bool bounds_done) {
  LX_HIP(hipSetDevice(device_));
  corner_index.bind(corner_stream);
  corner_index.build(d_corner, nc, bounds_done);
  corner_index.bind(st_);
  surf_index.build(d_surf, ns, bounds_done);
}

void Registrar::stage_submap_device(const float4* d_corner, uint32_t nc, const float4* d_surf, uint32_t ns, hipEvent_t wait_for) {
  LX_HIP(hipSetDevice(device_));
  if (!st_build_) {
    st_build_ = create_stream(0);   // (a lowest-priority stream is starved for milliseconds by the two latency-critical ones)
    LX_HIP(hipEventCreateWithFlags(&ev_build_, hipEventDisableTiming));
    LX_HIP(hipEventCreateWithFlags(&ev_swap_, hipEventDisableTiming));
    corner_next_.init(st_build_);
    surf_next_.init(st_build_);
  }
  // the buffers being rebuilt held the previous epoch's index until the last swap: registrations enqueued before that swap
  // may still read them
  if (swapped_once_) LX_HIP(hipStreamWaitEvent(st_build_, ev_swap_, 0));
  if (wait_for) LX_HIP(hipStreamWaitEvent(st_build_, wait_for, 0));   // the producer of the caller's buffers (a copy, an RCCL broadcast)
  corner_next_.bind(st_build_);
  surf_next_.bind(st_build_);
  corner_next_.build(d_corner, nc);
  surf_next_.build(d_surf, ns);
  LX_HIP(hipEventRecord(ev_build_, st_build_));
  next_staged_ = true;
}

void Registrar::stage_submap_host(const loamx_cloud* corner, const loamx_cloud* surf) {
  check_cloud(corner, false);
  check_cloud(surf, false);
  LX_HIP(hipSetDevice(device_));
  if (st_build_) LX_HIP(hipStreamSynchronize(st_build_));   // the pinned staging buffers of an earlier staging are free again
  h_next_corner_.reserve(corner->count + 1);
  h_next_surf_.reserve(surf->count + 1);
  pack_cloud(corner, h_next_corner_.p);
  pack_cloud(surf, h_next_surf_.p);
  next_corner_.reserve(corner->count + 1);
  next_surf_.reserve(surf->count + 1);
  if (!st_build_) stage_submap_device(nullptr, 0, nullptr, 0);   // creates the build stream and events
  LX_HIP(hipMemcpyAsync(next_corner_.p, h_next_corner_.p, sizeof(float4) * corner->count, hipMemcpyHostToDevice, st_build_));
  LX_HIP(hipMemcpyAsync(next_surf_.p, h_next_surf_.p, sizeof(float4) * surf->count, hipMemcpyHostToDevice, st_build_));
  stage_submap_device(next_corner_.p, corner->count, next_surf_.p, surf->count);
}

void Registrar::swap_submap() {
  LX_REQUIRE(next_staged_, "swap_submap() without a staged sub-map");
  LX_HIP(hipSetDevice(device_));
  LX_HIP(hipStreamWaitEvent(st_, ev_build_, 0));   // registrations enqueued from now on see the finished index
  LX_HIP(hipEventRecord(ev_swap_, st_));           // ... and everything enqueued so far still used the old one
  corner_index.swap(corner_next_);
  surf_index.swap(surf_next_);
  corner_index.bind(st_);
  surf_index.bind(st_);
  next_staged_ = false;
  swapped_once_ = true;
}

void Registrar::upload(uint32_t n_sweeps, const loamx_cloud* corner_last, const loamx_cloud* surf_last, const loamx_cloud* full_res,
                       const float* guess6, bool wait) {
  LX_REQUIRE(n_sweeps >= 1 && n_sweeps <= max_sweeps_, "n_sweeps out of range for this handle");
  LX_REQUIRE(corner_last && surf_last && guess6, "NULL input");
  LX_HIP(hipSetDevice(device_));
  n_sweeps_ = n_sweeps;
  h_seg_off_.assign(2 * n_sweeps + 1, 0);
  h_full_off_.assign(n_sweeps + 1, 0);
  max_q_per_sweep_ = 0;
  for (uint32_t s = 0; s < n_sweeps; s++) {
    check_cloud(&corner_last[s], false);
    check_cloud(&surf_last[s], false);
    h_seg_off_[2 * s + 1] = h_seg_off_[2 * s] + corner_last[s].count;
    h_seg_off_[2 * s + 2] = h_seg_off_[2 * s + 1] + surf_last[s].count;
    max_q_per_sweep_ = std::max(max_q_per_sweep_, corner_last[s].count + surf_last[s].count);
    if (full_res) {
      check_cloud(&full_res[s], false);
      h_full_off_[s + 1] = h_full_off_[s] + full_res[s].count;
    }
  }
  n_in_ = h_seg_off_[2 * n_sweeps];
  n_full_ = h_full_off_[n_sweeps];
  h_in_.reserve(n_in_ + 1);
  in_.reserve(n_in_ + 1);
  stack_.reserve(n_in_ + 1);
  ds_pts_.reserve(n_in_ + 1);
  vox_.reserve(n_in_ + 1, 2 * n_sweeps);
  LX_REQUIRE(n_in_ < SCAN_MAX_N, "too many feature points in one batch");
  for (uint32_t s = 0; s < n_sweeps; s++) {
    pack_cloud(&corner_last[s], h_in_.p + h_seg_off_[2 * s]);
    pack_cloud(&surf_last[s], h_in_.p + h_seg_off_[2 * s + 1]);
  }
  // (the feature clouds of the registration are finite by contract, as the odometry's: common.h packed_all_finite)
  if (!packed_all_finite(h_in_.p, n_in_)) throw Error(LOAMX_E_INVALID, "a feature cloud holds non-finite coordinates");
  fetch_from_pinned(in_.p, h_in_.p, n_in_, st_);   // (single-sweep sizes by kernel, a batch's MiB by the copy engine: pinned_copy.hpp)
  if (n_full_) {
    h_full_.reserve(n_full_);
    full_.reserve(n_full_);
    for (uint32_t s = 0; s < n_sweeps; s++) pack_cloud(&full_res[s], h_full_.p + h_full_off_[s]);
    fetch_from_pinned(full_.p, h_full_.p, n_full_, st_);
  }
  {   // guesses / offsets travel as ONE block through pinned memory owned by this object (as in upload_device; copies from the
      // pageable vectors were staged by the runtime, one wait each)
    const size_t nseg = 2 * (size_t)n_sweeps;
    const size_t o_off = sizeof(float) * 6 * n_sweeps, o_full = o_off + sizeof(uint32_t) * (nseg + 1);
    const size_t bytes = o_full + sizeof(uint32_t) * (n_sweeps + 1);
    h_blob_.reserve(bytes + 16);
    blob_.reserve(bytes + 16);
    memcpy(h_blob_.p, guess6, sizeof(float) * 6 * n_sweeps);
    memcpy(h_blob_.p + o_off, h_seg_off_.data(), sizeof(uint32_t) * (nseg + 1));
    memcpy(h_blob_.p + o_full, h_full_off_.data(), sizeof(uint32_t) * (n_sweeps + 1));
    LX_HIP(hipMemcpyAsync(blob_.p, h_blob_.p, bytes, hipMemcpyHostToDevice, st_));
    d_guess_ = (const float*)blob_.p;
    d_seg_off_ = (const uint32_t*)(blob_.p + o_off);
    d_full_off_ = (const uint32_t*)(blob_.p + o_full);
    d_src_ = nullptr;
  }
  nblk_ = max_q_per_sweep_ / GN_TILE + 2;   // >= ceil(corner / tile) + ceil(surf / tile) of every sweep
  partials_.reserve((size_t)n_sweeps * nblk_ * LX_NSUM);
  // the copies read this object's pinned staging, which the next upload() rewrites: a caller that does not synchronise with the
  // stream before then (loamx_batch_upload twice in a row) needs the wait; Mapper::process always ends synchronised
  if (wait) LX_HIP(hipStreamSynchronize(st_));
}

float4* Registrar::stage_full(uint32_t n_sweeps, const uint32_t* n_full) {
  LX_REQUIRE(n_sweeps >= 1 && n_sweeps <= max_sweeps_, "n_sweeps out of range for this handle");
  full_next_staged_ = false;   // (a pre-staged buffer that was not adopted is simply dropped)
  h_full_off_.assign(n_sweeps + 1, 0);
  for (uint32_t s = 0; s < n_sweeps; s++) h_full_off_[s + 1] = h_full_off_[s] + n_full[s];
  n_full_ = h_full_off_[n_sweeps];
  if (double_buffer_full) {   // the previous run's registered clouds stay readable (an asynchronous download may still be copying them)
    std::swap(full_.p, full_alt_.p);
    std::swap(full_.cap, full_alt_.cap);
  }
  full_.reserve((size_t)n_full_ + 1);
  full_staged_ = true;
  return full_.p;
}

// The staging area of the NEXT run while the current one is still in flight (the pipeline re-projects the next step's
// full-resolution clouds into it behind the current step's first Gauss-Newton launches: that work does not depend on the
// registration's poses).  Needs double_buffer_full.  adopt_full_next() makes it the current run's staging area.
float4* Registrar::stage_full_next(uint32_t n_sweeps, const uint32_t* n_full) {
  LX_REQUIRE(double_buffer_full, "internal: pre-staging needs the second full-resolution buffer");
  LX_REQUIRE(n_sweeps >= 1 && n_sweeps <= max_sweeps_, "n_sweeps out of range for this handle");
  next_full_off_.assign(n_sweeps + 1, 0);
  for (uint32_t s = 0; s < n_sweeps; s++) next_full_off_[s + 1] = next_full_off_[s] + n_full[s];
  if ((size_t)next_full_off_[n_sweeps] + 1 > full_alt_.cap) return nullptr;   // (growing would free a buffer a download may still read: the caller stages at the usual time)
  full_next_staged_ = true;
  return full_alt_.p;
}
bool Registrar::adopt_full_next(uint32_t n_sweeps, const uint32_t* n_full) {
  if (!full_next_staged_ || next_full_off_.size() != n_sweeps + 1) { full_next_staged_ = false; return false; }
  for (uint32_t s = 0; s < n_sweeps; s++)
    if (next_full_off_[s + 1] - next_full_off_[s] != n_full[s]) { full_next_staged_ = false; return false; }
  full_next_staged_ = false;
  h_full_off_ = next_full_off_;
  n_full_ = h_full_off_[n_sweeps];
  std::swap(full_.p, full_alt_.p);
  std::swap(full_.cap, full_alt_.cap);
  full_staged_ = true;
  return true;
}

void Registrar::upload_device(uint32_t n_sweeps, const float4* const* corner_last, const uint32_t* n_corner,
                              const float4* const* surf_last, const uint32_t* n_surf, const float4* const* full_res,
                              const uint32_t* n_full, const float* guess6) {
  LX_REQUIRE(n_sweeps >= 1 && n_sweeps <= max_sweeps_, "n_sweeps out of range for this handle");
  LX_HIP(hipSetDevice(device_));
  n_sweeps_ = n_sweeps;
  const bool staged = full_staged_ && !full_res;   // the caller wrote the full-resolution clouds through stage_full()
  full_staged_ = false;
  h_seg_off_.assign(2 * n_sweeps + 1, 0);
  if (!staged) h_full_off_.assign(n_sweeps + 1, 0);
  max_q_per_sweep_ = 0;
  for (uint32_t s = 0; s < n_sweeps; s++) {
    h_seg_off_[2 * s + 1] = h_seg_off_[2 * s] + n_corner[s];
    h_seg_off_[2 * s + 2] = h_seg_off_[2 * s + 1] + n_surf[s];
    max_q_per_sweep_ = std::max(max_q_per_sweep_, n_corner[s] + n_surf[s]);
    if (!staged) h_full_off_[s + 1] = h_full_off_[s] + (full_res ? n_full[s] : 0u);
  }
  n_in_ = h_seg_off_[2 * n_sweeps];
  if (!staged) n_full_ = h_full_off_[n_sweeps];
  in_.reserve(n_in_ + 1);
  stack_.reserve(n_in_ + 1);
  ds_pts_.reserve(n_in_ + 1);
  vox_.reserve(n_in_ + 1, 2 * n_sweeps);
  LX_REQUIRE(n_in_ < SCAN_MAX_N, "too many feature points in one batch");
  // guesses / offsets / source pointers travel as ONE block through pinned memory owned by this object
  const size_t nseg = 2 * (size_t)n_sweeps;
  const size_t o_off = sizeof(float) * 6 * n_sweeps, o_full = o_off + sizeof(uint32_t) * (nseg + 1);
  const size_t o_src = (o_full + sizeof(uint32_t) * (n_sweeps + 1) + 15) & ~(size_t)15;
  const size_t bytes = o_src + sizeof(float4*) * (nseg + n_sweeps);
  h_blob_.reserve(bytes + 16);
  blob_.reserve(bytes + 16);
  memcpy(h_blob_.p, guess6, sizeof(float) * 6 * n_sweeps);
  memcpy(h_blob_.p + o_off, h_seg_off_.data(), sizeof(uint32_t) * (nseg + 1));
  memcpy(h_blob_.p + o_full, h_full_off_.data(), sizeof(uint32_t) * (n_sweeps + 1));
  const float4** hsrc = (const float4**)(h_blob_.p + o_src);
  for (uint32_t s = 0; s < n_sweeps; s++) { hsrc[2 * s] = corner_last[s]; hsrc[2 * s + 1] = surf_last[s]; }
  for (uint32_t s = 0; s < n_sweeps; s++) hsrc[nseg + s] = full_res ? full_res[s] : nullptr;
  d_guess_ = (const float*)blob_.p;
  d_seg_off_ = (const uint32_t*)(blob_.p + o_off);
  d_full_off_ = (const uint32_t*)(blob_.p + o_full);
  d_src_ = (const float4* const*)(blob_.p + o_src);
  if (n_full_) full_.reserve(n_full_);
  if (n_sweeps == 1 && n_full_ && !staged && bytes == sizeof(ParamBlock1)) {   // the block travels in the gather's arguments (k_gather_one)
    ParamBlock1 pb1;
    memcpy(pb1.w, h_blob_.p, sizeof(pb1));
    hipLaunchKernelGGL(k_gather_one, dim3((n_full_ + 255) / 256), dim3(256), 0, st_, full_.p, full_res[0], n_full_, pb1, (uint32_t*)blob_.p);
  } else {
    LX_HIP(hipMemcpyAsync(blob_.p, h_blob_.p, bytes, hipMemcpyHostToDevice, st_));
    if (n_full_ && !staged)
      hipLaunchKernelGGL(k_gather_segments, dim3((n_full_ + 255) / 256), dim3(256), 0, st_, full_.p, d_full_off_, n_sweeps, d_src_ + nseg,
                         n_full_);
  }
  nblk_ = max_q_per_sweep_ / GN_TILE + 2;   // >= ceil(corner / tile) + ceil(surf / tile) of every sweep
  partials_.reserve((size_t)n_sweeps * nblk_ * LX_NSUM);
}

static double host_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// transformFullResToMap (BasicLaserMapping.cpp:235-240); mode as in k_transform_full
void Registrar::enqueue_full(int mode) {
  if (!n_full_) return;
  if (mode == 1 && ++full_tag_ == 0u) full_tag_ = 1u;
  const uint32_t* skip = vb_unchecked_ ? vb_.d_fail_word() : nullptr;   // an unverified bucketed voxel stage may have given up
  hipLaunchKernelGGL(k_transform_full, dim3((n_full_ + 255) / 256), dim3(256), 0, st_, full_.p, n_full_, d_full_off_, n_sweeps_, poses_.p,
                     stats_.p, full_done_.p, mode, full_tag_, skip, vb_.epoch());
}

// k_pose_init + stack round trip + voxel grid of the stack clouds
void Registrar::enqueue_front(bool legacy) {
  const uint32_t ns = n_sweeps_, nseg = 2 * ns, n = n_in_;
  hipLaunchKernelGGL(k_pose_init, dim3((ns + 63) / 64), dim3(64), 0, st_, d_guess_, ns, poses_.p, stats_.p, vox_.seg_minmax(), arrive_.p,
                     full_done_.p);
  if (n == 0) {
    LX_HIP(hipMemsetAsync(ds_off_.p, 0, sizeof(uint32_t) * (nseg + 1), st_));
    return;
  }
  if (legacy) {
    const uint32_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_stack, dim3(nb), dim3(256), 0, st_, in_.p, d_src_, n, d_seg_off_, nseg, poses_.p, 1.0f / params.corner_leaf,
                       1.0f / params.surf_leaf, stack_.p, vox_.ijk(), vox_.seg_minmax());
    vox_.sort_reduce(stack_.p, nullptr, n, d_seg_off_, nseg, ds_pts_.p, ds_off_.p);
  } else {
    vb_.run(in_.p, d_src_, n, d_seg_off_, h_seg_off_.data(), nseg, poses_.p, 1.0f / params.corner_leaf, 1.0f / params.surf_leaf, stack_.p,
            ds_pts_.p, ds_off_.p);
  }
}

// The Gauss-Newton iterations: one k_gn_iter launch each.  Converged sweeps turn the remaining launches into no-ops on the
// device (stats.done), which keeps run_async() free of host round trips.  A blocking caller (early_exit) enqueues one launch
// more than the previous call needed (a launch over converged sweeps costs a few microseconds, a host round trip in the middle
// of the chain ~90) together with the registration of the full-resolution clouds of the sweeps that converge within them, looks
// at the flags (mirrored into pinned memory by the update step), and only goes on launching while some sweep is not done.
// Returns true when the host, at its first look, finds that the bucketed voxel stage gave up: the caller runs again.
// Blocks until every sweep's mirror says "done" (polled: the flags are written by the update step of the launch that
// converged, ahead of the launches enqueued behind it) or until the stream reaches ev_look_.  Returns true when the flags
// ended the wait.  A mirror counts only when its check word matches the pose / statistics words read after the flag.
bool Registrar::wait_for_mirrors() {
  const bool no_poll = diag_env("LOAMX_NO_MIRROR_POLL") != nullptr;   // (diagnostic; read per call so that bench.py --ab can toggle it)
  if (no_poll) { LX_HIP(hipEventSynchronize(ev_look_)); return false; }
  const uint32_t ns = n_sweeps_;
  const volatile SweepStats* hs = h_stats_.p;
  for (unsigned spin = 0;; spin++) {
    bool all = true;
    for (uint32_t k = 0; k < ns && all; k++) all = hs[k].done != 0;
    if (all) {
      std::atomic_thread_fence(std::memory_order_acquire);
      for (uint32_t k = 0; k < ns && all; k++) {
        SweepStats st;
        Pose T;
        memcpy(&st, (const void*)&h_stats_.p[k], sizeof(st));
        memcpy(&T, (const void*)&h_poses_.p[k], sizeof(T));
        all = st.pad1 == mirror_check_word(st, T);
      }
      if (all) return true;
    }
    if ((spin & 15) == 15) {
      const hipError_t q = hipEventQuery(ev_look_);
      if (q == hipSuccess) return false;
      if (q != hipErrorNotReady) LX_HIP(q);
    }
    __builtin_ia32_pause();
  }
}

bool Registrar::run_iterations(bool trace, double& th2, double& th3) {
  TraceRange trace_range("loamx:registration:gauss-newton");
  const uint32_t ns = n_sweeps_;
  struct ProfDump { std::function<void()> f; ~ProfDump() { if (f) f(); } } prof_dump;
  GnArgs a;
  a.ds_pts = ds_pts_.p; a.ds_off = ds_off_.p; a.poses = poses_.p; a.stats = stats_.p;
  a.cdesc = corner_index.desc(); a.cpts = corner_index.sorted(); a.cstart = corner_index.cell_start();
  a.sdesc = surf_index.desc(); a.spts = surf_index.sorted(); a.sstart = surf_index.cell_start();
  a.partials = partials_.p; a.arrive = arrive_.p; a.matP = matP_.p;
  a.host_stats = h_stats_.p; a.host_poses = h_poses_.p;
  a.nblk = nblk_; a.delta_t_abort = params.delta_t_abort; a.delta_r_abort = params.delta_r_abort;
  const dim3 grid(8 * ((nblk_ + 7) / 8 + 1), ns);
  a.dbg = nullptr;
  a.skip_word = vb_unchecked_ ? vb_.d_fail_word() : nullptr;   // (offsets of a run that gave up part-way may be a mix)
  a.skip_value = vb_.epoch();
#ifdef LOAMX_PROF_GN
  static DevBuf<unsigned long long> dbg;
  dbg.reserve((size_t)nblk_ * 8 + 64);
  LX_HIP(hipMemsetAsync(dbg.p, 0, sizeof(unsigned long long) * ((size_t)nblk_ * 8 + 64), st_));
  a.dbg = dbg.p;
  prof_dump.f = [&]() {
    std::vector<unsigned long long> h((size_t)nblk_ * 8);
    (void)hipStreamSynchronize(st_);
    (void)hipMemcpy(h.data(), dbg.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (uint32_t t = 0; t < nblk_; t++) if (h[8 * t]) t0 = std::min(t0, h[8 * t]);
    fprintf(stderr, "[gn iter 0, sweep 0: tile: start / search / rows / reduce us (+solve)]");
    for (uint32_t t = 0; t < nblk_; t++) {
      const unsigned long long* v = &h[8 * t];
      if (!v[0]) continue;
      if (t % 16 == 0 || v[5]) fprintf(stderr, " %u:%.0f/%.0f/%.0f/%.0f", t, (v[0] - t0) * 0.01, (v[1] - v[0]) * 0.01, (v[2] - v[1]) * 0.01, (v[3] - v[2]) * 0.01);
      if (v[5]) fprintf(stderr, "(+solve %.0f, ends at %.0f)", (v[5] - v[4]) * 0.01, (v[5] - t0) * 0.01);
    }
    fprintf(stderr, "\n");
    unsigned long long ts[16];
    (void)hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_solve_ts), sizeof(ts));
    fprintf(stderr, "[solve_sweep, us since entry] partial sums %.1f  sums+AtA %.1f  qr %.1f  projector %.1f  trig %.1f  update+stores %.1f\n", (ts[1] - ts[0]) * 0.01,
            (ts[2] - ts[0]) * 0.01, (ts[3] - ts[0]) * 0.01, (ts[4] - ts[0]) * 0.01, (ts[5] - ts[0]) * 0.01, (ts[6] - ts[0]) * 0.01);
  };
#endif   // per XCD: ceil(corner tiles / 8) + ceil(surf tiles / 8) workgroups at most
  int it = 0;
  bool waited = false, spec_full = false, all_done = false;
  const int maxit = params.max_iterations;
  const bool want_full = n_full_ && !defer_full;
  // (LOAMX_GN_AHEAD, diagnostic: launches enqueued beyond the previous call's need — 1 by default; a launch over converged sweeps costs ~4 us
  // of the stream, a host round trip for a launch that turns out to be missing ~20)
  static const int ahead = diag_env("LOAMX_GN_AHEAD") ? std::max(0, atoi(diag_env("LOAMX_GN_AHEAD"))) : 1;
  int chunk = early_exit ? std::min(std::max(pred_iters_ + ahead, 2), maxit) : maxit;
  while (it < maxit) {
    const int end = std::min(maxit, it + chunk);
    for (; it < end; it++) {
      const bool tm = timing_ && launch_timing_ && n_res_launch_ < 64;
      if (tm) LX_HIP(hipEventRecord(ev_[2 + 2 * n_res_launch_], st_));
      a.iter = it;
#ifdef LOAMX_DIAG
      if (jacobi_eig_) hipLaunchKernelGGL(k_gn_iter_jacobi, grid, dim3(LX_RES_THREADS), 0, st_, a);
      else
#endif
      hipLaunchKernelGGL(k_gn_iter, grid, dim3(LX_RES_THREADS), 0, st_, a);
      if (tm) {
        LX_HIP(hipEventRecord(ev_[3 + 2 * n_res_launch_], st_));
        n_res_launch_++;
      }
    }
    if (!early_exit) break;
    if (!spec_full && want_full) { enqueue_full(1); spec_full = true; }
    if (it >= maxit) break;
    if (trace && th2 == 0) th2 = host_us();
    // the host waits for THIS point of the stream, not for the stream: the caller's on_first_wait may enqueue work of the next run
    // behind it (the pipeline's pre-staged re-projection), which must not hold this run's results back
    if (!ev_look_) LX_HIP(hipEventCreateWithFlags(&ev_look_, hipEventDisableTiming));
    LX_HIP(hipEventRecord(ev_look_, st_));
    if (on_first_wait && !waited) { waited = true; on_first_wait(); on_first_wait = nullptr; }   // host work that overlaps the wait
    if (trace && th3 == 0) th3 = host_us();
    const bool polled = wait_for_mirrors();
    if (vb_unchecked_) {
      vb_unchecked_ = false;
      if (vb_.failed()) { note_bucket_give_up(); return true; }
    }
    int need = 0;
    all_done = true;
    for (uint32_t k = 0; k < ns; k++) {
      all_done = all_done && h_stats_.p[k].done;
      need = std::max(need, h_stats_.p[k].iterations);
    }
    // poses / stats are final and on the host; the clouds registered by the speculative launch are behind them in the stream
    // (every reader of the clouds goes through the stream or an event recorded on it), and so may be a converged launch or two
    if (all_done) { pred_iters_ = need; mirrors_written_ = true; results_final_ = spec_full || !want_full; (void)polled; break; }
    pred_iters_ = it + 1;
    chunk = 1;
  }
  if (early_exit && want_full) {
    if (!spec_full) enqueue_full(0);
    else if (!all_done) enqueue_full(2);   // the sweeps that were not done when the first launch looked
    full_enqueued_ = true;
  }
  return false;
}

void Registrar::run_async() {
  LX_REQUIRE(n_sweeps_ > 0, "run() before upload()");
  TraceRange trace_range("loamx:registration");
  static const bool trace = getenv("LOAMX_REG_TRACE") != nullptr;
  const double th0 = trace ? host_us() : 0.0;
  double th1 = 0, th2 = 0, th3 = 0;
  LX_HIP(hipSetDevice(device_));
  const uint32_t ns = n_sweeps_, nseg = 2 * ns, n = n_in_;
  const bool can_bucket = !vb_disabled_ && VoxBucket::fits(n, nseg);
  for (int attempt = 0; attempt < 2; attempt++) {
    const bool legacy = attempt == 1 || !can_bucket;
    if (timing_) LX_HIP(hipEventRecord(ev_[0], st_));   // (per attempt: a run repeated through the general voxel kernel reports the attempt that counted — ADVICE.md round 3)
    n_res_launch_ = 0;
    if (early_exit) memset(h_stats_.p, 0, sizeof(SweepStats) * ns);   // mirrors of sweeps that never reach an update stay "not done"
    vb_unchecked_ = !legacy && n > 0;
    full_dl_sweep_ = -1;
    full_enqueued_ = false;
    results_final_ = false;
    enqueue_front(legacy);
    if (trace) th1 = host_us();
    mirrors_written_ = false;
    bool again = false;
    if (submap_sufficient() && n > 0) {   // BasicLaserMapping.cpp:628-629 guard
      again = run_iterations(trace, th2, th3);
      if (trace && th2 == 0) th2 = host_us();
    }
    if (!again && early_exit && vb_unchecked_) {   // a blocking caller goes on to use the voxel stage's output: look at its verdict now
      LX_HIP(hipStreamSynchronize(st_));
      vb_unchecked_ = false;
      again = vb_.failed();
      if (again) note_bucket_give_up();
    }
    if (!again) break;
  }
  if (n_full_ && !defer_full && !full_enqueued_) enqueue_full(0);
  if (timing_) { LX_HIP(hipEventRecord(ev_[1], st_)); timed_run_ = true; }
  LX_HIP(hipGetLastError());
  if (on_first_wait) on_first_wait();   // host work of the caller that overlaps the device work enqueued above
  if (trace && th3 == 0) th3 = host_us();
  if (trace)
    fprintf(stderr, "[reg] voxel stage enqueued %.0f us, first iterations enqueued %.0f, callback done %.0f, run_async returns %.0f\n", th1 - th0,
            th2 - th0, th3 - th0, host_us() - th0);
}

void Registrar::note_bucket_give_up() {
  vb_give_ups_++;
  static const bool trace = getenv("LOAMX_VB_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[voxbucket] gave up (reason mask 0x%x, %u points, %u sweeps): repeating through the general kernel\n", vb_.why(), n_in_, n_sweeps_);
}

// after a stream synchronisation: a bucketed voxel stage that gave up left an empty query set (and the full-resolution clouds
// untouched); run the sweeps again through the general kernel
void Registrar::redo_if_bucket_path_failed() {
  if (!vb_unchecked_) return;
  vb_unchecked_ = false;
  if (!vb_.failed()) return;
  note_bucket_give_up();
  const bool keep = vb_disabled_;
  vb_disabled_ = true;
  try { run_async(); } catch (...) { vb_disabled_ = keep; throw; }
  vb_disabled_ = keep;
  LX_HIP(hipStreamSynchronize(st_));
}

// replace the device poses (6 floats per sweep) and register the full-resolution clouds with them — the tail of a
// run_async() that ran with defer_full
void Registrar::finish_with_poses(const float* poses6) {
  const uint32_t ns = n_sweeps_;
  h_guess_.reserve((size_t)6 * ns + 8);
  memcpy(h_guess_.p, poses6, sizeof(float) * 6 * ns);
  guess_.reserve((size_t)6 * ns + 8);
  LX_HIP(hipMemcpyAsync(guess_.p, h_guess_.p, sizeof(float) * 6 * ns, hipMemcpyHostToDevice, st_));
  hipLaunchKernelGGL(k_pose_set, dim3((ns + 63) / 64), dim3(64), 0, st_, guess_.p, ns, poses_.p);
  enqueue_full(0);
  full_dl_sweep_ = -1;
  mirrors_written_ = false;   // the device poses were replaced
  results_final_ = false;     // ... and new device work was enqueued: the next fetch waits for the stream
  LX_HIP(hipGetLastError());
}

// parity hook: the product's neighbour search for arbitrary map-frame points (see k_knn_probe)
void Registrar::knn_probe(int which, const float* xyz, uint32_t n, uint32_t* idx5, float* d2_5) {
  LX_REQUIRE(which == 0 || which == 1, "which must be 0 (corner sub-map) or 1 (surf sub-map)");
  LX_REQUIRE(xyz && idx5 && d2_5, "NULL argument");
  SubMapIndex& ix = which == 0 ? corner_index : surf_index;
  LX_REQUIRE(ix.size() >= 1, "the sub-map is empty");
  LX_HIP(hipSetDevice(device_));
  if (!n) return;
  DevBuf<float4> dq;
  DevBuf<uint32_t> di;
  DevBuf<float> dd;
  dq.reserve(n); di.reserve((size_t)5 * n); dd.reserve((size_t)5 * n);
  std::vector<float4> hq(n);
  for (uint32_t i = 0; i < n; i++) hq[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f);
  LX_HIP(hipMemcpyAsync(dq.p, hq.data(), sizeof(float4) * n, hipMemcpyHostToDevice, st_));
  hipLaunchKernelGGL(k_knn_probe, dim3((n + 256 / KNN_LPQ - 1) / (256 / KNN_LPQ)), dim3(256), 0, st_, dq.p, n, ix.desc(), ix.sorted(), ix.cell_start(), di.p, dd.p);
  LX_HIP(hipMemcpyAsync(idx5, di.p, sizeof(uint32_t) * 5 * n, hipMemcpyDeviceToHost, st_));
  LX_HIP(hipMemcpyAsync(d2_5, dd.p, sizeof(float) * 5 * n, hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
}

// parity hook: the update steps' 6x6 solve, wave-cooperative (as the kernels run it) next to the scalar routine
__global__ __launch_bounds__(64) void k_qr6_probe(const float* __restrict__ ata, const float* __restrict__ atb, uint32_t n, float* __restrict__ x_coop,
                                                  float* __restrict__ x_scalar) {
  __shared__ float A[36], B[6], X[6];
  const uint32_t i = blockIdx.x;
  const int tid = (int)threadIdx.x;
  if (tid < 36) A[tid] = ata[36 * (size_t)i + tid];
  if (tid < 6) { B[tid] = atb[6 * (size_t)i + tid]; X[tid] = 0.f; }
  __syncthreads();
  qr_solve6_coop(A, B, X);
  __syncthreads();
  if (tid < 6) x_coop[6 * (size_t)i + tid] = X[tid];
  if (tid == 0) {
    float M[6][6], b[6], x[6];
    for (int r = 0; r < 6; r++) {
      b[r] = B[r];
      for (int c = 0; c < 6; c++) M[r][c] = A[r * 6 + c];
    }
    qr_solve<6, 6>(M, b, x);
    for (int r = 0; r < 6; r++) x_scalar[6 * (size_t)i + r] = x[r];
  }
}
void Registrar::qr6_probe(const float* ata, const float* atb, uint32_t n, float* x_coop, float* x_scalar) {
  LX_REQUIRE(ata && atb && x_coop && x_scalar, "NULL argument");
  LX_HIP(hipSetDevice(device_));
  if (!n) return;
  DevBuf<float> da, db, dx, dy;
  da.reserve((size_t)36 * n); db.reserve((size_t)6 * n); dx.reserve((size_t)6 * n); dy.reserve((size_t)6 * n);
  LX_HIP(hipMemcpyAsync(da.p, ata, sizeof(float) * 36 * n, hipMemcpyHostToDevice, st_));
  LX_HIP(hipMemcpyAsync(db.p, atb, sizeof(float) * 6 * n, hipMemcpyHostToDevice, st_));
  hipLaunchKernelGGL(k_qr6_probe, dim3(n), dim3(64), 0, st_, da.p, db.p, n, dx.p, dy.p);
  LX_HIP(hipMemcpyAsync(x_coop, dx.p, sizeof(float) * 6 * n, hipMemcpyDeviceToHost, st_));
  LX_HIP(hipMemcpyAsync(x_scalar, dy.p, sizeof(float) * 6 * n, hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
}

// Stress probe of the exchange primitive k_odom_lm rests on (dev_math.hpp: xrec_store / xrec_load).  The assumption: an aligned 16-byte
// agent-scope store is observed whole or split at 8 bytes, never finer, so a record read with BOTH tags equal to k holds both halves
// of value k.  Workgroup pairs (2 p, 2 p + 1) — consecutive workgroups run on different XCDs — hammer shared records: thread t of the
// producer writes versions 1 .. rounds of record (p, t) back to back, thread t of the consumer reads the record as fast as it can until
// it has seen the last version.  Counted: accepted reads (tags equal), torn reads (tags differ: harmless, the reader would poll again)
// and — the property itself — accepted reads whose value is not the one its tag names.  out[0..3] = accepted, torn, inconsistent, timed-out threads
__device__ __forceinline__ double xrec_probe_value(unsigned k) {
  return __hiloint2double((int)(k * 2654435761u ^ 0x5bd1e995u), (int)(~k * 40503u + 0x9e3779b9u));
}
__global__ __launch_bounds__(64) void k_xrec_stress(xrec_t* __restrict__ rec, unsigned rounds, unsigned long long* __restrict__ out) {
  const unsigned p = blockIdx.x >> 1, t = threadIdx.x;
  xrec_t* r = rec + (size_t)p * 64 + t;
  if ((blockIdx.x & 1u) == 0u) {
    for (unsigned k = 1; k <= rounds; k++) xrec_store(r, xrec_probe_value(k), k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  unsigned long long acc = 0, torn = 0, bad = 0, late = 0;
  const unsigned long long t0 = wall_clock64();
  unsigned last = 0, polls = 0;
  while (last != rounds) {
    const xrec_t v = xrec_load(r);
    if (v.x == v.w) {
      if (v.x != 0u) {
        acc++;
        const double want = xrec_probe_value(v.x);
        if (__double2hiint(xrec_value(v)) != __double2hiint(want) || __double2loint(xrec_value(v)) != __double2loint(want)) bad++;
        last = v.x;
      }
    } else {
      torn++;
    }
    if ((++polls & 4095u) == 0u && wall_clock64() - t0 > 300000000ull) { late = 1; break; }   // ~3 s: the producer never became resident
  }
  atomicAdd(&out[0], acc); atomicAdd(&out[1], torn); atomicAdd(&out[2], bad); atomicAdd(&out[3], late);
}
void Registrar::xrec_stress(uint32_t pairs, uint32_t rounds, unsigned long long out4[4]) {
  LX_REQUIRE(out4 && pairs >= 1 && pairs <= 512 && rounds >= 1, "invalid argument");
  LX_HIP(hipSetDevice(device_));
  DevBuf<xrec_t> rec;
  DevBuf<unsigned long long> out;
  rec.reserve((size_t)pairs * 64);
  out.reserve(4);
  LX_HIP(hipMemsetAsync(rec.p, 0, sizeof(xrec_t) * pairs * 64, st_));
  LX_HIP(hipMemsetAsync(out.p, 0, sizeof(unsigned long long) * 4, st_));
  hipLaunchKernelGGL(k_xrec_stress, dim3(2 * pairs), dim3(64), 0, st_, rec.p, rounds, out.p);
  LX_HIP(hipMemcpyAsync(out4, out.p, sizeof(unsigned long long) * 4, hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
}

void Registrar::sync() { wait_stream(st_); }

// final poses / statistics on the host: an early-exit run that saw every sweep converge already holds them (the update
// step mirrors them into pinned memory), anything else is copied
void Registrar::fetch_results() {
  if (results_final_ && !vb_unchecked_) { vox_.check(); vb_.check(); scan_check_errors(); return; }   // run_iterations saw everything complete at its event: no need to wait for what has been enqueued since
  LX_HIP(hipStreamSynchronize(st_));
  vox_.check();
  vb_.check();
  scan_check_errors();
  redo_if_bucket_path_failed();
  if (!mirrors_written_) {
    LX_HIP(hipMemcpyAsync(h_poses_.p, poses_.p, sizeof(Pose) * n_sweeps_, hipMemcpyDeviceToHost, st_));
    LX_HIP(hipMemcpyAsync(h_stats_.p, stats_.p, sizeof(SweepStats) * n_sweeps_, hipMemcpyDeviceToHost, st_));
    LX_HIP(hipStreamSynchronize(st_));
    mirrors_written_ = true;
  }
}

void Registrar::download_stats(SweepStats* out) {
  fetch_results();
  memcpy(out, h_stats_.p, sizeof(SweepStats) * n_sweeps_);
}

void Registrar::download(float* poses6, int* stats4) {
  LX_REQUIRE(n_sweeps_ > 0, "download() before run()");
  fetch_results();
  for (uint32_t s = 0; s < n_sweeps_; s++) {
    if (poses6) {
      const Pose& T = h_poses_.p[s];
      float* o = poses6 + 6 * s;
      o[0] = T.rx; o[1] = T.ry; o[2] = T.rz; o[3] = T.tx; o[4] = T.ty; o[5] = T.tz;
    }
    if (stats4) {
      const SweepStats& st = h_stats_.p[s];
      int* o = stats4 + 4 * s;
      o[0] = st.iterations; o[1] = st.sel; o[2] = st.corner_q; o[3] = st.surf_q;
    }
  }
}

// The registered full-resolution cloud of one sweep, asked for ahead of time: the copy into pinned memory is enqueued behind the
// registration's last launch and overlaps whatever the caller enqueues next (Mapper: the map update); download_full_res() of the same
// sweep then only waits for the stream and unpacks.
void Registrar::download_full_res_async(uint32_t sweep, const loamx_cloud* into) {
  LX_REQUIRE(sweep < n_sweeps_, "sweep index out of range");
  const uint32_t a = h_full_off_[sweep], b = h_full_off_[sweep + 1];
  // a landing area of packed records in runtime-pinned memory with room for the cloud takes the copy itself (download_full_res() then
  // only waits); anything else goes through this object's pinned block and is unpacked there
  full_dl_direct_ = nullptr;
  if (into && b > a && into->count >= b - a && packed_layout(into) && host_pinned(into->data, sizeof(float4) * (b - a))) {
    // (a kernel storing to the pinned block instead of this copy was measured: 8 us slower per sweep — profiles/r05_ab.md section 6)
    LX_HIP(hipMemcpyAsync(into->data, full_.p + a, sizeof(float4) * (b - a), hipMemcpyDeviceToHost, st_));
    full_dl_direct_ = into->data;
  } else {
    h_full_dl_.reserve((size_t)(b - a) + 1);
    if (b > a) LX_HIP(hipMemcpyAsync(h_full_dl_.p, full_.p + a, sizeof(float4) * (b - a), hipMemcpyDeviceToHost, st_));
  }
  full_dl_sweep_ = (int)sweep;
}

int Registrar::download_full_res(uint32_t sweep, loamx_cloud* out) {
  LX_REQUIRE(sweep < n_sweeps_, "sweep index out of range");
  check_cloud(out, false);
  fetch_results();
  const uint32_t a = h_full_off_[sweep], b = h_full_off_[sweep + 1];
  const bool landed = full_dl_sweep_ == (int)sweep && full_dl_direct_ && full_dl_direct_ == out->data;   // the copy went straight into `out`
  if (full_dl_sweep_ != (int)sweep || (full_dl_direct_ && !landed)) {   // not asked for ahead of time (or the clouds were registered again since, or into other memory): copy now
    h_full_dl_.reserve((size_t)(b - a) + 1);   // (pinned: a copy into pageable memory is staged by the runtime)
    if (b > a) LX_HIP(hipMemcpyAsync(h_full_dl_.p, full_.p + a, sizeof(float4) * (b - a), hipMemcpyDeviceToHost, st_));
  }
  full_dl_sweep_ = -1;
  full_dl_direct_ = nullptr;
  wait_stream(st_);
  if (landed) { out->count = b - a; return LOAMX_OK; }
  return unpack_cloud(h_full_dl_.p, b - a, out);
}

void Registrar::download_ds(uint32_t sweep, std::vector<float4>& corner_ds, std::vector<float4>& surf_ds) {
  LX_REQUIRE(sweep < n_sweeps_, "sweep index out of range");
  fetch_results();
  uint32_t off[3];
  LX_HIP(hipMemcpyAsync(off, ds_off_.p + 2 * sweep, sizeof(off), hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
  corner_ds.resize(off[1] - off[0]);
  surf_ds.resize(off[2] - off[1]);
  if (!corner_ds.empty())
    LX_HIP(hipMemcpyAsync(corner_ds.data(), ds_pts_.p + off[0], sizeof(float4) * corner_ds.size(), hipMemcpyDeviceToHost, st_));
  if (!surf_ds.empty())
    LX_HIP(hipMemcpyAsync(surf_ds.data(), ds_pts_.p + off[1], sizeof(float4) * surf_ds.size(), hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
}

void Registrar::get_timing(float ms[4], uint64_t counts[4]) {
  for (int k = 0; k < 4; k++) { ms[k] = 0.f; counts[k] = 0; }
  if (!timing_ || !timed_run_) return;
  LX_HIP(hipEventSynchronize(ev_[1]));
  LX_HIP(hipEventElapsedTime(&ms[0], ev_[0], ev_[1]));
  for (int k = 0; k < n_res_launch_; k++) {   // the Gauss-Newton launches
    float t = 0.f;
    LX_HIP(hipEventElapsedTime(&t, ev_[2 + 2 * k], ev_[3 + 2 * k]));
    ms[1] += t;
  }
  counts[0] = n_res_launch_;
  std::vector<SweepStats> st(n_sweeps_);
  download_stats(st.data());
  for (auto& s : st) {
    counts[1] += (uint64_t)s.iterations * (uint64_t)(s.corner_q + s.surf_q);
    counts[2] += (uint64_t)(s.corner_q + s.surf_q);
  }
}

}  // namespace loamx
