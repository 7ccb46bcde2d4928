// Segmented voxel-grid down-sampling kernels (see voxel.hpp).
#include "voxel.hpp"
#include <vector>

namespace loamx {

__global__ void k_vox_init_minmax(int* mm, uint32_t nseg) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * nseg) mm[i] = (i % 6) < 3 ? 2147483647 : (-2147483647 - 1);
}

// the bounds' initial values and k_vox_ds's cleared block (counters, histograms, tile status) in one launch
__global__ __launch_bounds__(256) void k_vox_prepare(int* __restrict__ mm, uint32_t nseg, uint4* __restrict__ zero, uint32_t nquads) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * nseg) mm[i] = (i % 6) < 3 ? 2147483647 : (-2147483647 - 1);
  for (uint32_t q = i; q < nquads; q += gridDim.x * blockDim.x) zero[q] = make_uint4(0u, 0u, 0u, 0u);
}

// seg_ids given (scattered segment ids: the map's cube slots): the bounds of up to VIJK_LDS_SEGS segments are reduced in LDS per
// workgroup and flushed with one global atomic per touched bound — per-lane global atomics on a few dozen segments' six words
// serialise (measured: up to 2 ms per launch on a 166 k-point sub-map).
constexpr int VIJK_LDS_SEGS = 256;
__global__ __launch_bounds__(256) void k_vox_ijk(const float4* __restrict__ pts, const uint8_t* __restrict__ valid, uint32_t n,
                                                 const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_ids,
                                                 uint32_t nseg, float inv_even, float inv_odd, int* __restrict__ ijk,
                                                 int* __restrict__ seg_minmax) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n && !(valid && !valid[i]);
  uint32_t seg = 0;
  int ix = 0, iy = 0, iz = 0;
  if (active) {
    seg = seg_ids ? seg_ids[i] : vox_find_seg(seg_off, nseg, i);
    const float inv = (seg & 1) ? inv_odd : inv_even;
    const float4 p = pts[i];
    ix = (int)floorf(p.x * inv); iy = (int)floorf(p.y * inv); iz = (int)floorf(p.z * inv);
    ijk[3 * i] = ix; ijk[3 * i + 1] = iy; ijk[3 * i + 2] = iz;
  }
  if (seg_ids && nseg <= (uint32_t)VIJK_LDS_SEGS) {   // (kernel-uniform)
    __shared__ int s_mm[VIJK_LDS_SEGS * 6];
    for (uint32_t e = threadIdx.x; e < 6 * nseg; e += 256) s_mm[e] = (e % 6) < 3 ? 2147483647 : (-2147483647 - 1);
    __syncthreads();
    if (active) {
      int* mm = s_mm + 6 * seg;
      atomicMin(&mm[0], ix); atomicMin(&mm[1], iy); atomicMin(&mm[2], iz);
      atomicMax(&mm[3], ix); atomicMax(&mm[4], iy); atomicMax(&mm[5], iz);
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < 6 * nseg; e += 256) {
      const int v = s_mm[e];
      if ((e % 6) < 3) { if (v != 2147483647) atomicMin(&seg_minmax[e], v); }
      else if (v != (-2147483647 - 1)) atomicMax(&seg_minmax[e], v);
    }
    return;
  }
  seg_minmax_update(seg_minmax, active, seg, ix, iy, iz);
}

// ----------------------------------------------------------------------------------------------------------------
// k_vox_ds: the whole segmented voxel grid in ONE persistent launch — keys, a stable LSD radix sort (8-bit digits, only
// as many passes as the keys have bits), run heads, their scan, the per-voxel means and the per-segment offsets.
// (Round 1 ran this as ~28 launches, 18 of them a library merge sort.)
//
// Key of a point = segment << B | PCL's own linear voxel index inside the segment's box,
//   ix + iy * dx + iz * dx * dy  with (ix, iy, iz) relative to the box's min corner and (dx, dy, dz) its extent in voxels
// (pcl::VoxelGrid::applyFilter) — it fits 31 bits exactly when PCL filters at all: a segment whose box has more than INT_MAX
// voxels is passed through unfiltered, as PCL does ("leaf size is too small"): every point keeps a key of its own.  B is the
// number of bits the largest linear index of the batch needs (found on the device from the segment boxes), so a typical batch
// of sweeps sorts on ~30 bits = 4 passes.  Ignored slots (valid[i] == 0) get the pseudo-segment nseg and sort behind
// everything.  The sort is stable, so the points of a voxel stay in input order and the float means are accumulated in the
// order of the reference's loop over a stably sorted index (PCL's std::sort leaves that order unspecified).
//
// grid = G workgroups of 256 threads working on tiles of 2048 elements.  Tiles are claimed from a counter and a phase is complete
// when its tiles are (TileSync below) — the launch does not rely on all of its workgroups being resident; a wait that is not
// satisfied within ~1 s raises the error word and ends the kernel instead of hanging the GPU.  Phases:
//   keys   key of every element + the digit histograms of ALL passes (the multiset of keys never changes)
//   pass p every workgroup scans the 256 digit totals, ranks its tile's elements stably (per wave: 8 ballots give the lanes with
//          the same digit; waves are chained through LDS), publishes the tile's digit counts, sums the counts of all earlier
//          tiles (look-back, batches of independent loads) and scatters keys + values to the other buffer
//   H      run heads, output positions (look-back over the tiles' head counts), voxel means walking each run in sorted = input
//          order out of LDS, per-segment output offsets
// ----------------------------------------------------------------------------------------------------------------
#ifdef LOAMX_PROF_VDS
__device__ unsigned long long g_vds_ts[64];
__device__ unsigned long long g_vds_arr[16][1024][2];   // per barrier, per workgroup: arrival, exit
__device__ int g_vds_n;
#define VDS_TS() do { if (blockIdx.x == 0 && threadIdx.x == 0) { int k_ = g_vds_n; if (k_ < 64) { g_vds_ts[k_] = wall_clock64(); g_vds_n = k_ + 1; } } } while (0)
#else
#define VDS_TS() do { } while (0)
#endif
constexpr int VDS_TILE = 2048;        // elements per tile (8 per thread)
constexpr int VDS_LOOK = 32;          // predecessor counts fetched at once
constexpr uint32_t VDS_SPIN_LIMIT = 1u << 20;
constexpr int VDS_ROW = 256 + 64;     // words between two tiles' published counts: not a power of two, so that the rows a look-back
                                      // fetches together fall on different memory channels


struct VdsArgs {
  const float4* pts;
  const uint8_t* valid;
  const uint32_t* seg_off;
  const uint32_t* seg_ids;
  const int* ijk;
  const int* seg_minmax;
  unsigned long long* keys[2];
  uint32_t* vals[2];
  uint32_t* gh;         // [8][256]  global digit histograms of the passes (zero at launch)
  uint32_t* status;     // [passes][ntiles][256]  a tile's digit counts + 1 once published (zero at launch)
  uint32_t* tile_cnt;   // [ntiles]  run heads of a tile + 1 once published (zero at launch)
  uint32_t* link;       // [ntiles]  TileSync's per-workgroup tile lists
  uint32_t* barrier;    // TileSync counters (zero at launch): [0..15] tiles done per phase, [16] tiles claimed
  uint32_t* err;        // host-visible error word
  float4* out;
  uint32_t* out_off;    // [nseg + 1]
  uint32_t n, nseg, ntiles;
};

// Work distribution that does not depend on how many workgroups are resident.  In the first phase the tiles are CLAIMED from a
// counter; a workgroup keeps the tiles it claimed for all later phases (it walks them through `link`), and a phase is complete
// when its tiles are — not when every workgroup of the grid has arrived.  A workgroup the dispatcher has not started yet thus
// owns nothing anybody waits for: several of these kernels run side by side on a process's streams and may together exceed the
// device, where a classic grid barrier would let two half-resident launches wait for each other forever.  Look-backs stay safe
// too: the claim counter is monotonic, so every tile below a claimed one belongs to a workgroup that is running, and tiles
// publish before they look back.  The claim of the NEXT tile is issued before the current one is processed, so its round trip
// is never exposed.
struct TileSync {
  uint32_t* next;   // tiles claimed (first phase)
  uint32_t* done;   // [phase]  tiles completed
  uint32_t* link;   // [tile]   the next tile of the same workgroup (VSEG_NONE: last)
  uint32_t* err;
  uint32_t ntiles;
  uint32_t first;   // the workgroup's first tile
  uint32_t seen;    // thread 0: done[phase] as read right after the workgroup's last completion
  bool ok;
};
constexpr uint32_t VSEG_NONE = 0xffffffffu;

// thread 0's view of the tile after `tile`: claimed (first phase) or read from the workgroup's list
__device__ inline uint32_t tile_peek(const TileSync& g, uint32_t ph, uint32_t tile) {
  if (threadIdx.x != 0) return VSEG_NONE;
  if (ph == 0) {
    const uint32_t c = atomicAdd(g.next, 1u);
    return c < g.ntiles ? c : VSEG_NONE;
  }
  return g.link[tile];
}
// the workgroup finished `tile` of phase ph (nxt: thread 0's tile_peek, taken before the tile was processed)
__device__ inline uint32_t tile_done(TileSync& g, uint32_t ph, uint32_t tile, uint32_t nxt, bool count) {
  __shared__ uint32_t sh_tile;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ph == 0) g.link[tile] = nxt;
    if (count) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // ONE thread per workgroup fences (an agent-scope fence writes back whole caches here)
      __hip_atomic_fetch_add(&g.done[ph], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nxt == VSEG_NONE) g.seen = __hip_atomic_load(&g.done[ph], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sh_tile = nxt;
  }
  __syncthreads();
  return sh_tile;
}
// wait until every tile of phase ph is complete
__device__ inline void phase_wait(TileSync& g, uint32_t ph) {
  __shared__ int sh_fail;
  __syncthreads();
  if (threadIdx.x == 0) {
    sh_fail = 0;
#ifdef LOAMX_PROF_VDS
    if (ph < 15 && blockIdx.x < 1024) g_vds_arr[ph][blockIdx.x][0] = wall_clock64();
#endif
    uint32_t spins = 0;
    while (g.seen < g.ntiles) {
      g.seen = __hip_atomic_load(&g.done[ph], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g.seen >= g.ntiles) break;
      __builtin_amdgcn_s_sleep(8);
      if (++spins > VDS_SPIN_LIMIT) { *g.err = 1u; sh_fail = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifdef LOAMX_PROF_VDS
    if (ph < 15 && blockIdx.x < 1024) g_vds_arr[ph][blockIdx.x][1] = wall_clock64();
#endif
  }
  __syncthreads();
  if (sh_fail) g.ok = false;
  g.seen = 0;
}

__device__ inline uint32_t vds_bits(unsigned long long v) { return v ? 64u - (uint32_t)__builtin_clzll(v) : 0u; }

__global__ __launch_bounds__(256) void k_vox_ds(const VdsArgs A) {
  __shared__ uint32_t s_gh[8 * 256];   // the tile's digit counts for every pass (key phase)
  __shared__ uint32_t s_base[256];
  __shared__ uint32_t s_toff[256];
  __shared__ uint32_t s_wcnt[4][256];
  __shared__ uint32_t s_scan[17];
  __shared__ unsigned long long s_max;
  __shared__ float4 s_pts[VDS_TILE + VDS_TILE / 8];               // phase H: the tile's points in sorted order (padded, see lp)
  __shared__ unsigned long long s_key[VDS_TILE + VDS_TILE / 8];   //          and their keys
  __shared__ unsigned long long s_prevk;
  __shared__ uint32_t s_first;
  const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t n = A.n, nseg = A.nseg, ntiles = A.ntiles;
#ifdef LOAMX_PROF_VDS
  if (blockIdx.x == 0 && threadIdx.x == 0) g_vds_n = 0;
#endif
  VDS_TS();
  uint32_t first_claim = 0;
  if (tid == 0) first_claim = atomicAdd(&A.barrier[16], 1u);   // the workgroup's first tile: the round trip overlaps the set-up below

  // ---- B: bits of the largest linear voxel index of the batch (pass-through segments: of the largest point index)
  if (tid == 0) s_max = 0ull;
  __syncthreads();
  {
    unsigned long long mx = 0ull;
    for (uint32_t sg = (uint32_t)tid; sg < nseg; sg += 256) {
      const int* mm = A.seg_minmax + 6 * (size_t)sg;
      if (mm[3] < mm[0]) continue;   // empty segment
      const long long dx = (long long)mm[3] - mm[0] + 1, dy = (long long)mm[4] - mm[1] + 1, dz = (long long)mm[5] - mm[2] + 1;
      // (three factors below 2^32 each: test the product without overflowing)
      const bool pass = dx > 2147483647LL || dy > 2147483647LL || dx * dy > 2147483647LL || dz > 2147483647LL / (dx * dy) + 1 || dx * dy * dz > 2147483647LL;
      const unsigned long long top = pass ? (unsigned long long)n : (unsigned long long)(dx * dy * dz - 1);
      mx = mx > top ? mx : top;
    }
    atomicMax(&s_max, mx);
  }
  __syncthreads();
  const uint32_t B = vds_bits(s_max);
  const uint32_t total_bits = B + vds_bits((unsigned long long)nseg);   // the pseudo-segment nseg must fit too
  const uint32_t P = (total_bits + 7) / 8 ? (total_bits + 7) / 8 : 1u;
  if (tid == 0) s_first = first_claim < ntiles ? first_claim : VSEG_NONE;
  __syncthreads();
  TileSync ts{A.barrier + 16, A.barrier, A.link, A.err, ntiles, s_first, 0u, true};
  uint32_t tile = ts.first;

  // ---- keys (into buffer 0) + the global digit histograms of ALL passes (the multiset of keys does not change from pass to
  // pass, so every pass's digit bases are known up front)
  while (tile != VSEG_NONE) {
    const uint32_t nxt = tile_peek(ts, 0, tile);
    for (uint32_t e = (uint32_t)tid; e < P * 256; e += 256) s_gh[e] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VDS_TILE / 256; j++) {
      const uint32_t i = tile * VDS_TILE + (uint32_t)(j * 256 + tid);
      if (i < n) {
        unsigned long long key;
        if (A.valid && !A.valid[i]) {
          key = (unsigned long long)nseg << B;
        } else {
          const uint32_t sg = A.seg_ids ? A.seg_ids[i] : vox_find_seg(A.seg_off, nseg, i);
          const int* mm = A.seg_minmax + 6 * (size_t)sg;
          const long long dx = (long long)mm[3] - mm[0] + 1, dy = (long long)mm[4] - mm[1] + 1, dz = (long long)mm[5] - mm[2] + 1;
          const bool pass = dx > 2147483647LL || dy > 2147483647LL || dx * dy > 2147483647LL || dz > 2147483647LL / (dx * dy) + 1 || dx * dy * dz > 2147483647LL;
          unsigned long long lin;
          if (pass) lin = (unsigned long long)(A.seg_ids ? i : i - A.seg_off[sg]);
          else lin = (unsigned long long)((long long)(A.ijk[3 * (size_t)i] - mm[0]) + (long long)(A.ijk[3 * (size_t)i + 1] - mm[1]) * dx +
                                          (long long)(A.ijk[3 * (size_t)i + 2] - mm[2]) * dx * dy);
          key = ((unsigned long long)sg << B) | lin;
        }
        A.keys[0][i] = key;
        A.vals[0][i] = i;
        for (uint32_t p = 0; p < P; p++) atomicAdd(&s_gh[p * 256 + (uint32_t)((key >> (8 * p)) & 255ull)], 1u);
      }
    }
    __syncthreads();
    for (uint32_t e = (uint32_t)tid; e < P * 256; e += 256)
      if (s_gh[e]) atomicAdd(&A.gh[e], s_gh[e]);
    tile = tile_done(ts, 0, tile, nxt, true);
  }
  VDS_TS();
  phase_wait(ts, 0);
  VDS_TS();
  if (!ts.ok) return;

  // ---- the passes: ONE phase each.  Per tile: stable ranks (per wave: 8 ballots give the lanes with the same digit; waves
  // chained through LDS), the tile's digit counts published (count + 1, 0 = not there yet), the counts of all earlier tiles
  // summed — independent loads, re-polled while a predecessor has not published (tiles are taken in increasing order by
  // running workgroups, see TileSync, so every predecessor is being worked on) —, then the scatter to the other buffer.
  for (uint32_t p = 0; p < P; p++) {
    const uint32_t shift = 8 * p;
    const unsigned long long* __restrict__ ksrc = A.keys[p & 1];
    const uint32_t* __restrict__ vsrc = A.vals[p & 1];
    unsigned long long* __restrict__ kdst = A.keys[(p & 1) ^ 1];
    uint32_t* __restrict__ vdst = A.vals[(p & 1) ^ 1];
    uint32_t* st = A.status + (size_t)p * ntiles * VDS_ROW;
    {
      uint32_t tot;
      const uint32_t ex = block_excl_scan(A.gh[p * 256 + tid], s_scan, tot);
      s_base[tid] = ex;
    }
    tile = ts.first;
    while (tile != VSEG_NONE) {
      const uint32_t nxt = tile_peek(ts, 1 + p, tile);
      __syncthreads();
#pragma unroll
      for (int w = 0; w < 4; w++) s_wcnt[w][tid] = 0u;
      __syncthreads();
      // wave w owns elements [512 w, 512 w + 512) of the tile, in 8 slots of 64 consecutive elements
      unsigned long long key[8];
      uint32_t val[8], rank[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = tile * VDS_TILE + (uint32_t)(wid * 512 + j * 64 + lane);
        const bool in = i < n;
        key[j] = in ? ksrc[i] : ~0ull;
        val[j] = in ? vsrc[i] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = tile * VDS_TILE + (uint32_t)(wid * 512 + j * 64 + lane);
        const bool in = i < n;
        const uint32_t d = (uint32_t)((key[j] >> shift) & 255ull);
        // lanes holding the same digit (lanes beyond the end match nobody: they are never scattered)
        unsigned long long m = __ballot(in);
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const unsigned long long bal = __ballot((d >> b) & 1u);
          m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        const int leader = __builtin_ctzll(m | (1ull << 63));
        uint32_t old = 0u;
        if (in && lane == leader) {
          old = s_wcnt[wid][d];
          s_wcnt[wid][d] = old + (uint32_t)__popcll(m);
        }
        old = __shfl(old, leader, 64);
        rank[j] = old + (uint32_t)__popcll(below);
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      {   // thread d: exclusive prefix over the four waves, the tile's count, the earlier tiles' counts
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t c = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += c; }
        __hip_atomic_store(&st[(size_t)tile * VDS_ROW + tid], run + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        for (uint32_t t0 = 0; t0 < tile; t0 += VDS_LOOK) {
          uint32_t v[VDS_LOOK];
          uint32_t spins = 0;
          for (;;) {   // the whole batch is re-fetched while any predecessor is unpublished: one round trip per retry, not per tile
            bool pending = false;
#pragma unroll
            for (int u = 0; u < VDS_LOOK; u++) v[u] = (t0 + u < tile) ? __hip_atomic_load(&st[(size_t)(t0 + u) * VDS_ROW + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
#pragma unroll
            for (int u = 0; u < VDS_LOOK; u++) pending |= v[u] == 0u;
            if (!pending) break;
            if (++spins > VDS_SPIN_LIMIT) { *A.err = 1u; break; }
            __builtin_amdgcn_s_sleep(4);
          }
#pragma unroll
          for (int u = 0; u < VDS_LOOK; u++) excl += v[u] ? v[u] - 1u : 0u;
        }
        s_toff[tid] = excl;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = tile * VDS_TILE + (uint32_t)(wid * 512 + j * 64 + lane);
        if (i < n) {
          const uint32_t d = (uint32_t)((key[j] >> shift) & 255ull);
          const uint32_t pos = s_base[d] + s_toff[d] + s_wcnt[wid][d] + rank[j];
          kdst[pos] = key[j];
          vdst[pos] = val[j];
        }
      }
      tile = tile_done(ts, 1 + p, tile, nxt, true);
    }
    // (the next pass reads what this pass scattered)
    VDS_TS();
    phase_wait(ts, 1 + p);
    VDS_TS();
    if (!ts.ok) return;
  }
  const unsigned long long* __restrict__ keys = A.keys[P & 1];
  const uint32_t* __restrict__ vals = A.vals[P & 1];

  // ---- H: run heads, output positions, voxel means and segment offsets in one phase (see k_vox_ds_seg: the tile's sorted keys
  // and points are staged in LDS, its head count is published and the heads of all earlier tiles come from a look-back).  A mean
  // is the sequential sum of the run in sorted (= input) order; a run that crosses the tile's end continues through global memory.
  // out_off[s] = voxels emitted before the first sorted element whose segment is >= s: written by the thread that owns that
  // element (for every segment skipped at the boundary), the tail by the owner of the last element.
  auto lp = [](uint32_t l) { return l + (l >> 3); };
  tile = ts.first;
  while (tile != VSEG_NONE) {
    const uint32_t nxt = tile_peek(ts, 1 + P, tile);
    const uint32_t beg = tile * VDS_TILE, end = min(beg + (uint32_t)VDS_TILE, n), tl = end - beg;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VDS_TILE / 256; j++) {
      const uint32_t l = (uint32_t)(j * 256 + tid);
      if (l < tl) {
        s_key[lp(l)] = keys[beg + l];
        s_pts[lp(l)] = A.pts[vals[beg + l]];
      }
    }
    if (tid == 0) s_prevk = beg > 0 ? keys[beg - 1] : ~0ull;
    __syncthreads();
    const uint32_t l0 = (uint32_t)tid * 8;
    unsigned long long k8[8];
    bool head[8];
    uint32_t cnt = 0;
    const unsigned long long prev0 = l0 == 0 ? s_prevk : (l0 <= tl ? s_key[lp(l0 - 1)] : ~0ull);
    unsigned long long prev = prev0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t l = l0 + j;
      k8[j] = l < tl ? s_key[lp(l)] : ~0ull;
      head[j] = l < tl && (k8[j] >> B) < nseg && (beg + l == 0 || k8[j] != prev);
      cnt += head[j] ? 1u : 0u;
      prev = k8[j];
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(cnt, s_scan, tot);
    if (tid == 0) __hip_atomic_store(&A.tile_cnt[tile], tot + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t part = 0;
    for (uint32_t t = (uint32_t)tid; t < tile; t += 256) {
      uint32_t v = __hip_atomic_load(&A.tile_cnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), spins = 0;
      while (v == 0u) {
        __builtin_amdgcn_s_sleep(4);
        v = __hip_atomic_load(&A.tile_cnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > VDS_SPIN_LIMIT) { *A.err = 1u; v = 1u; }
      }
      part += v - 1u;
    }
    uint32_t base;
    (void)block_excl_scan(part, s_scan, base);   // base = heads emitted by all earlier tiles
    uint32_t pos = base + ex;
    prev = prev0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t l = l0 + j;
      if (l < tl) {
        const uint32_t sg = (uint32_t)min(k8[j] >> B, (unsigned long long)nseg);
        const uint32_t sp = beg + l == 0 ? 0u : (uint32_t)min(prev >> B, (unsigned long long)nseg) + 1u;   // first segment not yet given an offset
        for (uint32_t q = sp; q <= sg; q++) A.out_off[q] = pos;   // (empty unless the segment changes here)
        if (beg + l == n - 1)
          for (uint32_t q = sg + 1; q <= nseg; q++) A.out_off[q] = pos + (head[j] ? 1u : 0u);
        if (head[j]) {
          float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
          uint32_t cntp = 0, e = l;
          do {
            const float4 q = s_pts[lp(e)];
            sx += q.x; sy += q.y; sz += q.z; si += q.w;
            cntp++;
            e++;
          } while (e < tl && s_key[lp(e)] == k8[j]);
          if (e == tl) {   // the run may continue in the next tile
            uint32_t g = end;
            while (g < n && keys[g] == k8[j]) {
              const float4 q = A.pts[vals[g]];
              sx += q.x; sy += q.y; sz += q.z; si += q.w;
              cntp++;
              g++;
            }
          }
          const float c = (float)cntp;
          A.out[pos] = make_float4(sx / c, sy / c, sz / c, si / c);
          pos++;
        }
        prev = k8[j];
      }
    }
    tile = tile_done(ts, 1 + P, tile, nxt, false);   // nobody waits for the last phase
  }
  VDS_TS();
}

// ----------------------------------------------------------------------------------------------------------------
// k_vox_ds_seg: the same job for the common case of CONTIGUOUS segments (seg_off given, every slot valid, <= 256 segments:
// the registration's stack clouds — sweep x {corner, surf}).  The input is already grouped by segment, so nothing has to be
// sorted across segments: tiles never straddle a segment, the key is the linear voxel index alone (<= 31 bits, typically
// 23-26 -> 3 passes of 9-bit digits instead of 4 of 8), digit histograms and bases are per segment, and a tile's look-back only
// covers the earlier tiles of its own segment (<= 17 for a 33 k-point surf cloud instead of all 146 tiles of the batch).
// ----------------------------------------------------------------------------------------------------------------
constexpr int VSEG_DB = 9, VSEG_NB = 1 << VSEG_DB, VSEG_ROW = VSEG_NB + 64;   // digit bits / bins
constexpr int VSEG_MAXSEG = 256;
constexpr int VSEG_IDX = 24;                         // low bits of a sort element: the point's position inside its segment
constexpr int VSEG_MAXPASS = 4;                      // 31 bits / 9

struct VsegArgs {
  const float4* pts;
  const uint32_t* seg_off;
  const int* ijk;
  const int* seg_minmax;
  unsigned long long* keys[2];   // (voxel index << VSEG_IDX) | position inside the segment: one 8-byte element carries key and payload
  uint32_t* gh;         // [nseg][VSEG_MAXPASS][VSEG_NB]  per-segment digit histograms (zero at launch)
  uint32_t* status;     // [VSEG_MAXPASS][max_tiles][VSEG_NB]  (zero at launch)
  uint32_t* tile_cnt;   // [max_tiles]  run heads of a tile + 1 once published (zero at launch)
  uint32_t* link;       // [max_tiles]  TileSync's per-workgroup tile lists
  uint32_t* barrier;
  uint32_t* err;
  float4* out;
  uint32_t* out_off;    // [nseg + 1]
  uint32_t n, nseg, max_tiles;
};

__global__ __launch_bounds__(256) void k_vox_ds_seg(const VsegArgs A) {
  __shared__ uint32_t s_gh[VSEG_MAXPASS * VSEG_NB];
  __shared__ uint32_t s_base[VSEG_NB];
  __shared__ uint32_t s_toff[VSEG_NB];
  __shared__ uint32_t s_wcnt[4][VSEG_NB];
  __shared__ uint32_t s_tbase[VSEG_MAXSEG + 1];   // first tile of every segment; [nseg] = number of tiles
  __shared__ uint32_t s_soff[VSEG_MAXSEG + 1];
  __shared__ uint32_t s_scan[17];
  __shared__ unsigned long long s_max;
  __shared__ float4 s_pts[VDS_TILE + VDS_TILE / 8];   // phase H: the tile's points in sorted order (padded, see lp)
  __shared__ uint32_t s_vox[VDS_TILE + VDS_TILE / 8];   //          and their voxel indices
  __shared__ uint32_t s_prev;
  const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t n = A.n, nseg = A.nseg;
#ifdef LOAMX_PROF_VDS
  if (blockIdx.x == 0 && threadIdx.x == 0) g_vds_n = 0;
#endif
  VDS_TS();
  uint32_t first_claim = 0;
  if (tid == 0) first_claim = atomicAdd(&A.barrier[16], 1u);   // the workgroup's first tile: the round trip overlaps the set-up below
  // ---- tiles per segment, B
  if (tid == 0) s_max = 0ull;
  {
    const uint32_t a = tid < (int)nseg ? A.seg_off[tid] : n, b = tid < (int)nseg ? A.seg_off[tid + 1] : n;
    if (tid < (int)nseg) s_soff[tid] = a;
    if (tid == 0) s_soff[nseg] = n;
    uint32_t tot;
    const uint32_t ex = block_excl_scan((b - a + VDS_TILE - 1) / VDS_TILE, s_scan, tot);
    if (tid < (int)nseg) s_tbase[tid] = ex;
    if (tid == 0) s_tbase[nseg] = tot;
  }
  __syncthreads();
  {
    unsigned long long mx = 0ull;
    if (tid < (int)nseg) {
      const int* mm = A.seg_minmax + 6 * (size_t)tid;
      if (mm[3] >= mm[0]) {
        const long long dx = (long long)mm[3] - mm[0] + 1, dy = (long long)mm[4] - mm[1] + 1, dz = (long long)mm[5] - mm[2] + 1;
        const bool pass = dx > 2147483647LL || dy > 2147483647LL || dx * dy > 2147483647LL || dz > 2147483647LL / (dx * dy) + 1 || dx * dy * dz > 2147483647LL;
        mx = pass ? (unsigned long long)(s_soff[tid + 1] - s_soff[tid]) : (unsigned long long)(dx * dy * dz - 1);
      }
    }
    atomicMax(&s_max, mx);
  }
  __syncthreads();
  const uint32_t B = vds_bits(s_max);
  const uint32_t P = (B + VSEG_DB - 1) / VSEG_DB ? (B + VSEG_DB - 1) / VSEG_DB : 1u;
  const uint32_t ntiles = s_tbase[nseg];
  if (tid == 0) s_prev = first_claim < ntiles ? first_claim : VSEG_NONE;
  __syncthreads();
  TileSync ts{A.barrier + 16, A.barrier, A.link, A.err, ntiles, s_prev, 0u, true};
  uint32_t tile = ts.first;
  auto seg_of_tile = [&](uint32_t t) {   // last segment whose first tile is <= t and that has tiles
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_tbase[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
  };

  // ---- keys + per-segment digit histograms of all passes
  while (tile != VSEG_NONE) {
    const uint32_t nxt = tile_peek(ts, 0, tile);
    const uint32_t sg = seg_of_tile(tile);
    const uint32_t beg = s_soff[sg] + (tile - s_tbase[sg]) * VDS_TILE, end = min(beg + (uint32_t)VDS_TILE, s_soff[sg + 1]);
    for (uint32_t e = (uint32_t)tid; e < P * VSEG_NB; e += 256) s_gh[e] = 0u;
    __syncthreads();
    const int* mm = A.seg_minmax + 6 * (size_t)sg;
    const long long dx = (long long)mm[3] - mm[0] + 1, dy = (long long)mm[4] - mm[1] + 1, dz = (long long)mm[5] - mm[2] + 1;
    const bool pass = dx > 2147483647LL || dy > 2147483647LL || dx * dy > 2147483647LL || dz > 2147483647LL / (dx * dy) + 1 || dx * dy * dz > 2147483647LL;
#pragma unroll
    for (int j = 0; j < VDS_TILE / 256; j++) {
      const uint32_t i = beg + (uint32_t)(j * 256 + tid);
      if (i < end) {
        unsigned long long key;
        if (pass) key = (unsigned long long)(i - s_soff[sg]);
        else key = (unsigned long long)((long long)(A.ijk[3 * (size_t)i] - mm[0]) + (long long)(A.ijk[3 * (size_t)i + 1] - mm[1]) * dx +
                                        (long long)(A.ijk[3 * (size_t)i + 2] - mm[2]) * dx * dy);
        A.keys[0][i] = (key << VSEG_IDX) | (unsigned long long)(i - s_soff[sg]);   // sort element: voxel index | position inside the segment
        for (uint32_t p = 0; p < P; p++) atomicAdd(&s_gh[p * VSEG_NB + (uint32_t)((key >> (VSEG_DB * p)) & (VSEG_NB - 1))], 1u);
      }
    }
    __syncthreads();
    for (uint32_t e = (uint32_t)tid; e < P * VSEG_NB; e += 256)
      if (s_gh[e]) atomicAdd(&A.gh[((size_t)sg * VSEG_MAXPASS) * VSEG_NB + e], s_gh[e]);
    tile = tile_done(ts, 0, tile, nxt, true);
  }
  VDS_TS();
  phase_wait(ts, 0);
  VDS_TS();
  if (!ts.ok) return;

  // ---- the passes (one phase each, see k_vox_ds); thread t owns bins t and t + 256
  for (uint32_t p = 0; p < P; p++) {
    const uint32_t shift = VSEG_IDX + VSEG_DB * p;
    const unsigned long long* __restrict__ ksrc = A.keys[p & 1];
    unsigned long long* __restrict__ kdst = A.keys[(p & 1) ^ 1];
    uint32_t* st = A.status + (size_t)p * A.max_tiles * VSEG_ROW;
    uint32_t cur_seg = 0xffffffffu;
    tile = ts.first;
    while (tile != VSEG_NONE) {
      const uint32_t nxt = tile_peek(ts, 1 + p, tile);
      const uint32_t sg = seg_of_tile(tile);
      const uint32_t sbeg = s_soff[sg];
      const uint32_t beg = sbeg + (tile - s_tbase[sg]) * VDS_TILE, end = min(beg + (uint32_t)VDS_TILE, s_soff[sg + 1]);
      __syncthreads();
      if (sg != cur_seg) {   // the segment's digit bases: exclusive scan of its 512-bin histogram (two bins per thread)
        cur_seg = sg;
        const uint32_t* h = A.gh + ((size_t)sg * VSEG_MAXPASS + p) * VSEG_NB;
        const uint32_t h0 = h[2 * tid], h1 = h[2 * tid + 1];
        uint32_t tot;
        const uint32_t ex = block_excl_scan(h0 + h1, s_scan, tot);
        s_base[2 * tid] = ex;
        s_base[2 * tid + 1] = ex + h0;
      }
#pragma unroll
      for (int w = 0; w < 4; w++) { s_wcnt[w][tid] = 0u; s_wcnt[w][tid + 256] = 0u; }
      __syncthreads();
      unsigned long long key[8];
      uint32_t rank[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = beg + (uint32_t)(wid * 512 + j * 64 + lane);
        const bool in = i < end;
        key[j] = in ? ksrc[i] : ~0ull;
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = beg + (uint32_t)(wid * 512 + j * 64 + lane);
        const bool in = i < end;
        const uint32_t d = (uint32_t)((key[j] >> shift) & (VSEG_NB - 1));
        unsigned long long m = __ballot(in);
#pragma unroll
        for (int b = 0; b < VSEG_DB; b++) {
          const unsigned long long bal = __ballot((d >> b) & 1u);
          m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        const int leader = __builtin_ctzll(m | (1ull << 63));
        uint32_t old = 0u;
        if (in && lane == leader) {
          old = s_wcnt[wid][d];
          s_wcnt[wid][d] = old + (uint32_t)__popcll(m);
        }
        old = __shfl(old, leader, 64);
        rank[j] = old + (uint32_t)__popcll(below);
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
#pragma unroll
      for (int half = 0; half < 2; half++) {   // bins tid and tid + 256: wave prefix, publish, look back inside the segment
        const uint32_t d = (uint32_t)tid + 256u * half;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t c = s_wcnt[w][d]; s_wcnt[w][d] = run; run += c; }
        __hip_atomic_store(&st[(size_t)tile * VSEG_ROW + d], run + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        for (uint32_t t0 = s_tbase[sg]; t0 < tile; t0 += 16) {
          uint32_t v[16];
          uint32_t spins = 0;
          for (;;) {   // see k_vox_ds: re-fetch the batch while any predecessor is unpublished
            bool pending = false;
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = (t0 + u < tile) ? __hip_atomic_load(&st[(size_t)(t0 + u) * VSEG_ROW + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
#pragma unroll
            for (int u = 0; u < 16; u++) pending |= v[u] == 0u;
            if (!pending) break;
            if (++spins > VDS_SPIN_LIMIT) { *A.err = 1u; break; }
            __builtin_amdgcn_s_sleep(4);
          }
#pragma unroll
          for (int u = 0; u < 16; u++) excl += v[u] ? v[u] - 1u : 0u;
        }
        s_toff[d] = excl;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = beg + (uint32_t)(wid * 512 + j * 64 + lane);
        if (i < end) {
          const uint32_t d = (uint32_t)((key[j] >> shift) & (VSEG_NB - 1));
          const uint32_t pos = sbeg + s_base[d] + s_toff[d] + s_wcnt[wid][d] + rank[j];
          kdst[pos] = key[j];
        }
      }
      tile = tile_done(ts, 1 + p, tile, nxt, true);
    }
    VDS_TS();
    phase_wait(ts, 1 + p);
    VDS_TS();
    if (!ts.ok) return;
  }
  const unsigned long long* __restrict__ keys = A.keys[P & 1];

  // ---- H: run heads, output positions and voxel means in one phase.  A tile stages its sorted elements' voxel indices and points
  // in LDS, counts its run heads and publishes the count; the heads emitted by all earlier tiles come from a look-back over the
  // published counts (as in the passes), so no grid barrier separates counting from emitting.  A mean is the sequential sum of the
  // run in sorted order (the reference accumulates in that order): the walk reads LDS and only continues through global memory
  // when a run crosses the tile's end.
  constexpr unsigned long long IDX_MASK = (1ull << VSEG_IDX) - 1ull;
  auto lp = [](uint32_t l) { return l + (l >> 3); };   // one pad slot per eight: a thread's eight consecutive elements start in distinct banks
  tile = ts.first;
  while (tile != VSEG_NONE) {
    const uint32_t nxt = tile_peek(ts, 1 + P, tile);
    const uint32_t sg = seg_of_tile(tile);
    const uint32_t sbeg = s_soff[sg], send = s_soff[sg + 1];
    const uint32_t beg = sbeg + (tile - s_tbase[sg]) * VDS_TILE, end = min(beg + (uint32_t)VDS_TILE, send);
    const uint32_t tl = end - beg;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VDS_TILE / 256; j++) {
      const uint32_t l = (uint32_t)(j * 256 + tid);
      if (l < tl) {
        const unsigned long long k = keys[beg + l];
        s_vox[lp(l)] = (uint32_t)(k >> VSEG_IDX);
        s_pts[lp(l)] = A.pts[sbeg + (uint32_t)(k & IDX_MASK)];
      }
    }
    if (tid == 0) s_prev = beg > sbeg ? (uint32_t)(keys[beg - 1] >> VSEG_IDX) : 0xffffffffu;
    __syncthreads();
    const uint32_t l0 = (uint32_t)tid * 8;
    bool head[8];
    uint32_t cnt = 0;
    uint32_t prev = l0 == 0 ? s_prev : (l0 <= tl ? s_vox[lp(l0 - 1)] : 0xffffffffu);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t l = l0 + j;
      const uint32_t v = l < tl ? s_vox[lp(l)] : 0xffffffffu;
      head[j] = l < tl && (beg + l == sbeg || v != prev);
      cnt += head[j] ? 1u : 0u;
      prev = v;
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan(cnt, s_scan, tot);
    if (tid == 0) __hip_atomic_store(&A.tile_cnt[tile], tot + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t part = 0;
    for (uint32_t t = (uint32_t)tid; t < tile; t += 256) {
      uint32_t v = __hip_atomic_load(&A.tile_cnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), spins = 0;
      while (v == 0u) {
        __builtin_amdgcn_s_sleep(4);
        v = __hip_atomic_load(&A.tile_cnt[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > VDS_SPIN_LIMIT) { *A.err = 1u; v = 1u; }
      }
      part += v - 1u;
    }
    uint32_t base;
    (void)block_excl_scan(part, s_scan, base);   // base = heads emitted by all earlier tiles
    for (uint32_t s = (uint32_t)tid; s <= nseg; s += 256) {   // output offsets: a segment starts where its first tile starts (empty segments: where the next one does)
      if (s_tbase[s] == tile) A.out_off[s] = base;
      else if (tile + 1 == ntiles && s_tbase[s] == ntiles) A.out_off[s] = base + tot;
    }
    uint32_t pos = base + ex;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (head[j]) {
        const uint32_t l = l0 + j;
        const uint32_t v = s_vox[lp(l)];
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        uint32_t cntp = 0, e = l;
        do {
          const float4 q = s_pts[lp(e)];
          sx += q.x; sy += q.y; sz += q.z; si += q.w;
          cntp++;
          e++;
        } while (e < tl && s_vox[lp(e)] == v);
        if (e == tl) {   // the run may continue in the segment's next tile
          uint32_t g = end;
          while (g < send) {
            const unsigned long long k = keys[g];
            if ((uint32_t)(k >> VSEG_IDX) != v) break;
            const float4 q = A.pts[sbeg + (uint32_t)(k & IDX_MASK)];
            sx += q.x; sy += q.y; sz += q.z; si += q.w;
            cntp++;
            g++;
          }
        }
        const float c = (float)cntp;
        A.out[pos] = make_float4(sx / c, sy / c, sz / c, si / c);
        pos++;
      }
    }
    tile = tile_done(ts, 1 + P, tile, nxt, false);   // nobody waits for the last phase
  }
  VDS_TS();
#ifdef LOAMX_PROF_VDS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_vds_arr[15][blockIdx.x][0] = wall_clock64();
#endif
}

void VoxelPipeline::init(hipStream_t st) {
  st_ = st;
}

void VoxelPipeline::reserve(uint32_t n, uint32_t nseg) {
  LX_REQUIRE(n < SCAN_MAX_N, "too many points for one voxel pass");
  LX_REQUIRE(nseg < (1u << 20), "too many voxel segments");
  ijk_.reserve((size_t)3 * n + 3);
  seg_minmax_.reserve((size_t)6 * nseg + 6);
  const size_t ntiles = ((size_t)n + VDS_TILE - 1) / VDS_TILE + 1;
  for (int k = 0; k < 2; k++) {
    keys_[k].reserve((size_t)n + 1);
    vals_[k].reserve((size_t)n + 1);
  }
  uint32_t seg_bits = 0;
  while ((nseg >> seg_bits) != 0) seg_bits++;
  const uint32_t passes = (31u + seg_bits + 7u) / 8u;   // upper bound of the kernel's pass count (31 bits of voxel index + the segment)
  status_words_ = (size_t)passes * ntiles * VDS_ROW;
  zero_words_ = 64 + 8 * 256 + status_words_ + ntiles;   // TileSync counters | gh | status | tile head counts
  zero_.reserve(zero_words_ + ntiles + 4);                                     // + the tile lists (not cleared)
  if (!h_err_.p) { h_err_.reserve(1); *h_err_.p = 0u; }
}

void VoxelPipeline::reset_minmax(uint32_t nseg) {
  hipLaunchKernelGGL(k_vox_init_minmax, dim3((6 * nseg + 255) / 256), dim3(256), 0, st_, seg_minmax_.p, nseg);
}

void VoxelPipeline::compute_ijk(const float4* pts, const uint8_t* valid, uint32_t n, const uint32_t* d_seg_off, uint32_t nseg,
                                float inv_even, float inv_odd, const uint32_t* d_seg_ids) {
  zero_ready_ = false;
  if (n == 0) { reset_minmax(nseg); return; }
  if (d_seg_ids) {   // the general kernel follows (sort_reduce): clear its block here instead of with a memset of its own
    reserve(n, nseg);
    const uint32_t nquads = (uint32_t)((zero_words_ + 3) / 4);   // (the tile lists behind the cleared block may be overwritten: they are not read before they are written)
    const uint32_t blocks = std::max<uint32_t>((6 * nseg + 255) / 256, std::min<uint32_t>((nquads + 255) / 256, 1024u));
    hipLaunchKernelGGL(k_vox_prepare, dim3(blocks), dim3(256), 0, st_, seg_minmax_.p, nseg, reinterpret_cast<uint4*>(zero_.p), nquads);
    zero_ready_ = true; zero_ready_n_ = n; zero_ready_nseg_ = nseg;
  } else {
    reset_minmax(nseg);
  }
  hipLaunchKernelGGL(k_vox_ijk, dim3((n + 255) / 256), dim3(256), 0, st_, pts, valid, n, d_seg_off, d_seg_ids, nseg, inv_even, inv_odd, ijk_.p,
                     seg_minmax_.p);
}

void VoxelPipeline::sort_reduce(const float4* pts, const uint8_t* valid, uint32_t n, const uint32_t* d_seg_off, uint32_t nseg,
                                float4* out, uint32_t* d_out_off, const uint32_t* d_seg_ids) {
  if (n == 0) {
    LX_HIP(hipMemsetAsync(d_out_off, 0, sizeof(uint32_t) * (nseg + 1), st_));
    return;
  }
  reserve(n, nseg);
  if (!slots_) {   // workgroups of k_vox_ds the device can hold at once; an upper bound of the useful grid size
    int per_cu = 0, dev = 0;
    LX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_vox_ds, 256, 0));
    LX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    LX_HIP(hipGetDeviceProperties(&prop, dev));
    slots_ = (uint32_t)std::max(per_cu, 1) * (uint32_t)std::max(prop.multiProcessorCount, 1);
  }
  if (!valid && !d_seg_ids && nseg <= VSEG_MAXSEG && n < (1u << VSEG_IDX) && !getenv("LOAMX_VDS_GLOBAL")) {   // contiguous segments: sort inside each segment only
    const uint32_t max_tiles = (n + VDS_TILE - 1) / VDS_TILE + nseg;
    const size_t words = 64 + (size_t)nseg * VSEG_MAXPASS * VSEG_NB + (size_t)VSEG_MAXPASS * max_tiles * VSEG_ROW + max_tiles;
    zero_.reserve(words + max_tiles);   // + the tile lists (no need to clear them)
    VsegArgs a;
    a.pts = pts; a.seg_off = d_seg_off; a.ijk = ijk_.p; a.seg_minmax = seg_minmax_.p;
    a.keys[0] = keys_[0].p; a.keys[1] = keys_[1].p;
    a.barrier = zero_.p; a.gh = zero_.p + 64; a.status = zero_.p + 64 + (size_t)nseg * VSEG_MAXPASS * VSEG_NB;
    a.tile_cnt = a.status + (size_t)VSEG_MAXPASS * max_tiles * VSEG_ROW;
    a.link = a.tile_cnt + max_tiles;
    a.err = h_err_.p; a.out = out; a.out_off = d_out_off;
    a.n = n; a.nseg = nseg; a.max_tiles = max_tiles;
    LX_HIP(hipMemsetAsync(zero_.p, 0, sizeof(uint32_t) * words, st_));
    if (!slots_seg_) {
      int per_cu = 0, dev = 0;
      LX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_vox_ds_seg, 256, 0));
      LX_HIP(hipGetDevice(&dev));
      hipDeviceProp_t prop;
      LX_HIP(hipGetDeviceProperties(&prop, dev));
      slots_seg_ = (uint32_t)std::max(per_cu, 1) * (uint32_t)std::max(prop.multiProcessorCount, 1);
    }
    uint32_t G = std::max<uint32_t>(1u, std::min<uint32_t>(max_tiles, slots_seg_));   // tiles are claimed, not dealt: residency is not a correctness condition
    if (const char* e = getenv("LOAMX_VDS_WGS")) { const int v = atoi(e); if (v >= 1) G = std::min<uint32_t>(G, (uint32_t)v); }
    hipLaunchKernelGGL(k_vox_ds_seg, dim3(G), dim3(256), 0, st_, a);
    LX_HIP(hipGetLastError());
#ifdef LOAMX_PROF_VDS
    {
      unsigned long long ts[64];
      int cnt = 0;
      LX_HIP(hipStreamSynchronize(st_));
      LX_HIP(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_vds_ts), sizeof(ts)));
      LX_HIP(hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_vds_n), sizeof(cnt)));
      {
        static std::vector<unsigned long long> arr(16 * 1024 * 2);
        LX_HIP(hipMemcpyFromSymbol(arr.data(), HIP_SYMBOL(g_vds_arr), sizeof(unsigned long long) * arr.size()));
        for (int b = 0; b < 4; b++) {
          unsigned long long a0 = ~0ull, a1 = 0, e0 = ~0ull, e1 = 0;
          for (uint32_t w = 0; w < G && w < 1024; w++) {
            const unsigned long long a_ = arr[((size_t)b * 1024 + w) * 2], e_ = arr[((size_t)b * 1024 + w) * 2 + 1];
            a0 = std::min(a0, a_); a1 = std::max(a1, a_); e0 = std::min(e0, e_); e1 = std::max(e1, e_);
          }
          fprintf(stderr, "[barrier %d] arrivals spread %.1f us, last arrival -> first exit %.1f us, exits spread %.1f us\n", b, (a1 - a0) * 0.01, ((double)e0 - (double)a1) * 0.01, (e1 - e0) * 0.01);
        }
      }
      fprintf(stderr, "[vox_ds_seg n=%u segs=%u G=%u] us (work | barrier wait):", n, nseg, G);
      for (int k = 1; k < cnt; k++) fprintf(stderr, "%s%.1f", (k & 1) ? "  " : "|", (ts[k] - ts[k - 1]) * 0.01);
      unsigned long long last = 0;
      {
        static std::vector<unsigned long long> arr(16 * 1024 * 2);
        LX_HIP(hipMemcpyFromSymbol(arr.data(), HIP_SYMBOL(g_vds_arr), sizeof(unsigned long long) * arr.size()));
        for (uint32_t w = 0; w < G && w < 1024; w++) last = std::max(last, arr[((size_t)15 * 1024 + w) * 2]);
      }
      fprintf(stderr, "  first workgroup %.1f, last %.1f\n", (ts[cnt - 1] - ts[0]) * 0.01, (last - ts[0]) * 0.01);
    }
#endif
    return;
  }
  VdsArgs a;
  a.pts = pts; a.valid = valid; a.seg_off = d_seg_off; a.seg_ids = d_seg_ids; a.ijk = ijk_.p; a.seg_minmax = seg_minmax_.p;
  a.keys[0] = keys_[0].p; a.keys[1] = keys_[1].p; a.vals[0] = vals_[0].p; a.vals[1] = vals_[1].p;
  a.err = h_err_.p; a.out = out; a.out_off = d_out_off;
  a.n = n; a.nseg = nseg; a.ntiles = (n + VDS_TILE - 1) / VDS_TILE;
  a.barrier = zero_.p; a.gh = zero_.p + 64; a.status = zero_.p + 64 + 8 * 256;
  a.tile_cnt = a.status + status_words_;
  a.link = zero_.p + zero_words_;
  if (!(zero_ready_ && zero_ready_n_ == n && zero_ready_nseg_ == nseg))
    LX_HIP(hipMemsetAsync(zero_.p, 0, sizeof(uint32_t) * zero_words_, st_));   // one block: barrier counter, digit histograms, tile status
  zero_ready_ = false;
  uint32_t G = std::max<uint32_t>(1u, std::min<uint32_t>(a.ntiles, slots_));   // tiles are claimed: residency is not a correctness condition
  if (const char* e = getenv("LOAMX_VDS_WGS")) { const int v = atoi(e); if (v >= 1) G = std::min<uint32_t>(G, (uint32_t)v); }
  hipLaunchKernelGGL(k_vox_ds, dim3(G), dim3(256), 0, st_, a);
  LX_HIP(hipGetLastError());
#ifdef LOAMX_PROF_VDS
  {
    unsigned long long ts[64];
    int cnt = 0;
    LX_HIP(hipStreamSynchronize(st_));
    LX_HIP(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_vds_ts), sizeof(ts)));
    LX_HIP(hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_vds_n), sizeof(cnt)));
    fprintf(stderr, "[vox_ds n=%u tiles=%u G=%u] us (work | barrier wait):", n, a.ntiles, G);
    for (int k = 1; k < cnt; k++) fprintf(stderr, "%s%.1f", (k & 1) ? "  " : "|", (ts[k] - ts[k - 1]) * 0.01);
    fprintf(stderr, "  total %.1f\n", (ts[cnt - 1] - ts[0]) * 0.01);
  }
#endif
}

// a grid barrier of k_vox_ds that timed out raised the (host-visible) error word: call after the stream has been synchronised
void VoxelPipeline::check() {
  if (h_err_.p && *(volatile uint32_t*)h_err_.p) {
    *h_err_.p = 0u;
    throw Error(LOAMX_E_HIP, "voxel grid: a wait inside k_vox_ds timed out");
  }
}

}  // namespace loamx
