// Segmented voxel-grid down-sampling kernels (see voxel.cuh).
#include "voxel.cuh"

namespace loamx {

__global__ void k_vox_init_minmax(int* mm, uint32_t nseg) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * nseg) mm[i] = (i % 6) < 3 ? 2147483647 : (-2147483647 - 1);
}

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
// grid = G workgroups of 256 threads, tiles of 2048 elements dealt round robin; phases are separated by a grid barrier
// (arrival counter in HBM, agent scope; the launcher keeps G within half of what the device can hold, and a barrier that is
// not passed within ~1 s raises the error word and ends the kernel instead of hanging the GPU).  Per pass:
//   A  per-tile digit histograms (LDS atomics)                            -> hist[tile][256]
//   B  one wave per digit scans its column over the tiles                 -> toff[tile][256], totals[256]
//   C  every workgroup scans the 256 totals, ranks its tile's elements stably (per wave: 8 ballots give the lanes with the
//      same digit; waves are chained through LDS) and scatters keys + values to the other buffer
// then  H1 run heads per tile -> H2 scan of the tile counts -> H3 output position of every head, the voxel's mean (walking the
// run in sorted = input order) -> H4 per-segment output offsets by binary search.
// ----------------------------------------------------------------------------------------------------------------
#ifdef LOAMX_PROF_VDS
__device__ unsigned long long g_vds_ts[64];
__device__ int g_vds_n;
#define VDS_TS() do { if (blockIdx.x == 0 && threadIdx.x == 0) { int k_ = g_vds_n; if (k_ < 64) { g_vds_ts[k_] = wall_clock64(); g_vds_n = k_ + 1; } } } while (0)
#else
#define VDS_TS() do { } while (0)
#endif
constexpr int VDS_TILE = 2048;
constexpr int VDS_LOOK = 32;          // predecessor counts fetched at once        // elements per tile (8 per thread)
constexpr uint32_t VDS_SPIN_LIMIT = 1u << 20;

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
  uint32_t* tile_cnt;   // [ntiles + 1]  heads per tile, then (in place) exclusive prefix; [ntiles] = total
  uint32_t* head_scan;  // [n + 1]
  float4* gathered;     // [n] the points in sorted order
  uint32_t* barrier;    // arrival counter, zero at launch
  uint32_t* err;        // host-visible error word
  float4* out;
  uint32_t* out_off;    // [nseg + 1]
  uint32_t n, nseg, ntiles;
};

struct GridSync {
  uint32_t* counter;
  uint32_t* err;
  uint32_t epoch;
  bool ok;
};
// all threads of all workgroups call it the same number of times.  ONE thread per workgroup issues the release / acquire
// fences: an agent-scope fence writes back / invalidates whole caches on this multi-XCD device, so hundreds of waves fencing
// at every barrier (the first version) cost ~25 us per barrier; the workgroup barriers around it extend the ordering to the
// other threads (their writes happen-before thread 0's release, their later reads happen-after its acquire; a workgroup lives
// on one CU, whose L1 thread 0 invalidates).
__device__ inline void grid_barrier(GridSync& g) {
  __shared__ int sh_fail;
  __syncthreads();
  g.epoch++;
  if (threadIdx.x == 0) {
    sh_fail = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // release this workgroup's writes
    atomicAdd(g.counter, 1u);
    const uint32_t target = g.epoch * gridDim.x;
    uint32_t spins = 0;
    while (__hip_atomic_load(g.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(32);   // ~1 us between polls: a hundred workgroups hammering one counter slow the L2 channel for everybody
      if (++spins > VDS_SPIN_LIMIT) { *g.err = 1u; sh_fail = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // acquire the other workgroups' writes
  }
  __syncthreads();
  if (sh_fail) g.ok = false;
}

__device__ inline uint32_t vds_bits(unsigned long long v) { return v ? 64u - (uint32_t)__builtin_clzll(v) : 0u; }

__global__ __launch_bounds__(256) void k_vox_ds(const VdsArgs A) {
  __shared__ uint32_t s_gh[8 * 256];   // the tile's digit counts for every pass (key phase)
  __shared__ uint32_t s_base[256];
  __shared__ uint32_t s_toff[256];
  __shared__ uint32_t s_wcnt[4][256];
  __shared__ uint32_t s_scan[17];
  __shared__ unsigned long long s_max;
  const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t n = A.n, nseg = A.nseg, ntiles = A.ntiles, G = gridDim.x;
  GridSync gs{A.barrier, A.err, 0u, true};
#ifdef LOAMX_PROF_VDS
  if (blockIdx.x == 0 && threadIdx.x == 0) g_vds_n = 0;
#endif
  VDS_TS();

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

  // ---- keys (into buffer 0) + the global digit histograms of ALL passes (the multiset of keys does not change from pass to
  // pass, so every pass's digit bases are known up front)
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += G) {
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
    __syncthreads();
  }
  VDS_TS();
  grid_barrier(gs);
  VDS_TS();
  if (!gs.ok) return;

  // ---- the passes: ONE phase each.  Per tile: stable ranks (per wave: 8 ballots give the lanes with the same digit; waves
  // chained through LDS), the tile's digit counts published (count + 1, 0 = not there yet), the counts of all earlier tiles
  // summed — independent loads, re-polled while a predecessor has not published (tiles are taken in increasing order by
  // resident workgroups, so every predecessor is being worked on) —, then the scatter to the other buffer.
  for (uint32_t p = 0; p < P; p++) {
    const uint32_t shift = 8 * p;
    const unsigned long long* __restrict__ ksrc = A.keys[p & 1];
    const uint32_t* __restrict__ vsrc = A.vals[p & 1];
    unsigned long long* __restrict__ kdst = A.keys[(p & 1) ^ 1];
    uint32_t* __restrict__ vdst = A.vals[(p & 1) ^ 1];
    uint32_t* st = A.status + (size_t)p * ntiles * 256;
    {
      uint32_t tot;
      const uint32_t ex = block_excl_scan(A.gh[p * 256 + tid], s_scan, tot);
      s_base[tid] = ex;
    }
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += G) {
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
        __hip_atomic_store(&st[(size_t)tile * 256 + tid], run + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        for (uint32_t t0 = 0; t0 < tile; t0 += VDS_LOOK) {
          uint32_t v[VDS_LOOK];
#pragma unroll
          for (int u = 0; u < VDS_LOOK; u++) v[u] = (t0 + u < tile) ? __hip_atomic_load(&st[(size_t)(t0 + u) * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
#pragma unroll
          for (int u = 0; u < VDS_LOOK; u++) {
            uint32_t spins = 0;
            while (v[u] == 0u) {   // a predecessor that has not published yet
              __builtin_amdgcn_s_sleep(16);
              v[u] = __hip_atomic_load(&st[(size_t)(t0 + u) * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (++spins > VDS_SPIN_LIMIT) { *A.err = 1u; v[u] = 1u; }
            }
            excl += v[u] - 1u;
          }
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
    }
    // (the next pass reads what this pass scattered)
    VDS_TS();
    grid_barrier(gs);
    VDS_TS();
    if (!gs.ok) return;
  }
  const unsigned long long* __restrict__ keys = A.keys[P & 1];
  const uint32_t* __restrict__ vals = A.vals[P & 1];

  // ---- H1: run heads per tile (thread t owns 8 consecutive elements)
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += G) {
    const uint32_t i0 = tile * VDS_TILE + (uint32_t)tid * 8;
    uint32_t cnt = 0;
    unsigned long long prev = (i0 > 0 && i0 <= n) ? keys[i0 - 1] : ~0ull;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = i0 + j;
      if (i < n) {
        const unsigned long long k = keys[i];
        if ((k >> B) < nseg && (i == 0 || k != prev)) cnt++;
        prev = k;
      }
    }
    uint32_t tot;
    (void)block_excl_scan(cnt, s_scan, tot);
    if (tid == 0) A.tile_cnt[tile] = tot;
    // the points in sorted order (independent gathers, coalesced stores): H3 then walks its runs through contiguous memory
#pragma unroll
    for (int j = 0; j < VDS_TILE / 256; j++) {
      const uint32_t i = tile * VDS_TILE + (uint32_t)(j * 256 + tid);
      if (i < n) A.gathered[i] = A.pts[vals[i]];
    }
  }
  VDS_TS();
  grid_barrier(gs);
  VDS_TS();
  if (!gs.ok) return;
  // ---- H2: exclusive scan of the tile counts (one wave of the first workgroup)
  if (blockIdx.x == 0 && wid == 0) {
    uint32_t running = 0;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 64) {
      const uint32_t t = t0 + (uint32_t)lane;
      const uint32_t v = t < ntiles ? A.tile_cnt[t] : 0u;
      const uint32_t inc = wave_incl_scan(v, lane);
      if (t < ntiles) A.tile_cnt[t] = running + inc - v;
      running += __shfl(inc, 63, 64);
    }
    if (lane == 0) { A.tile_cnt[ntiles] = running; A.head_scan[n] = running; }
  }
  VDS_TS();
  grid_barrier(gs);
  VDS_TS();
  if (!gs.ok) return;
  // ---- H3: output position of every element's voxel, the means
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += G) {
    const uint32_t i0 = tile * VDS_TILE + (uint32_t)tid * 8;
    unsigned long long k8[8];
    bool head[8];
    uint32_t cnt = 0;
    unsigned long long prev = (i0 > 0 && i0 <= n) ? keys[i0 - 1] : ~0ull;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = i0 + j;
      k8[j] = i < n ? keys[i] : ~0ull;
      head[j] = i < n && (k8[j] >> B) < nseg && (i == 0 || k8[j] != prev);
      cnt += head[j] ? 1u : 0u;
      prev = k8[j];
    }
    uint32_t tot;
    uint32_t pos = A.tile_cnt[tile] + block_excl_scan(cnt, s_scan, tot);
    float4 g8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) g8[j] = (i0 + j < n) ? A.gathered[i0 + j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = i0 + j;
      if (i < n) {
        A.head_scan[i] = pos;
        if (head[j]) {
          // float means of x, y, z, intensity over the run, in sorted (= input) order: first inside this thread's own eight
          // elements (registers), then — a run that continues — from memory
          float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
          uint32_t cntp = 0;
          bool open = true;
#pragma unroll
          for (int jj = 0; jj < 8; jj++) {
            if (jj >= j && open) {
              if (i0 + jj < n && k8[jj] == k8[j]) { sx += g8[jj].x; sy += g8[jj].y; sz += g8[jj].z; si += g8[jj].w; cntp++; }
              else open = false;
            }
          }
          if (open) {
            uint32_t e = i0 + 8;
            while (e < n && keys[e] == k8[j]) {
              const float4 q = A.gathered[e];
              sx += q.x; sy += q.y; sz += q.z; si += q.w;
              cntp++;
              e++;
            }
          }
          const float c = (float)cntp;
          A.out[pos] = make_float4(sx / c, sy / c, sz / c, si / c);
          pos++;
        }
      }
    }
  }
  VDS_TS();
  grid_barrier(gs);
  VDS_TS();
  if (!gs.ok) return;
  // ---- H4: out_off[s] = voxels emitted before segment s (first sorted slot whose segment >= s); out_off[nseg] = total
  for (uint32_t sg = blockIdx.x * 256 + (uint32_t)tid; sg <= nseg; sg += G * 256) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if ((keys[mid] >> B) >= sg) hi = mid; else lo = mid + 1;
    }
    A.out_off[sg] = A.head_scan[lo];
  }
  VDS_TS();
}

void VoxelPipeline::init(hipStream_t st) {
  st_ = st;
  tile_sums_.reserve(8192);
  scratch_.reserve(16);
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
  zero_words_ = 64 + 8 * 256 + (size_t)passes * ntiles * 256;   // barrier | gh | status
  zero_.reserve(zero_words_);
  tile_cnt_.reserve(ntiles + 2);
  head_scan_.reserve((size_t)n + 2);
  gathered_.reserve((size_t)n + 1);
  if (!h_err_.p) { h_err_.reserve(1); *h_err_.p = 0u; }
}

void VoxelPipeline::reset_minmax(uint32_t nseg) {
  hipLaunchKernelGGL(k_vox_init_minmax, dim3((6 * nseg + 255) / 256), dim3(256), 0, st_, seg_minmax_.p, nseg);
}

void VoxelPipeline::compute_ijk(const float4* pts, const uint8_t* valid, uint32_t n, const uint32_t* d_seg_off, uint32_t nseg,
                                float inv_even, float inv_odd, const uint32_t* d_seg_ids) {
  reset_minmax(nseg);
  if (n == 0) return;
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
  if (!slots_) {   // workgroups of k_vox_ds the device can hold at once; the launch uses at most half of them (its barriers spin)
    int per_cu = 0, dev = 0;
    LX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_vox_ds, 256, 0));
    LX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    LX_HIP(hipGetDeviceProperties(&prop, dev));
    slots_ = (uint32_t)std::max(per_cu, 1) * (uint32_t)std::max(prop.multiProcessorCount, 1);
  }
  VdsArgs a;
  a.pts = pts; a.valid = valid; a.seg_off = d_seg_off; a.seg_ids = d_seg_ids; a.ijk = ijk_.p; a.seg_minmax = seg_minmax_.p;
  a.keys[0] = keys_[0].p; a.keys[1] = keys_[1].p; a.vals[0] = vals_[0].p; a.vals[1] = vals_[1].p;
  a.tile_cnt = tile_cnt_.p; a.head_scan = head_scan_.p; a.gathered = gathered_.p;
  a.err = h_err_.p; a.out = out; a.out_off = d_out_off;
  a.n = n; a.nseg = nseg; a.ntiles = (n + VDS_TILE - 1) / VDS_TILE;
  a.barrier = zero_.p; a.gh = zero_.p + 64; a.status = zero_.p + 64 + 8 * 256;
  LX_HIP(hipMemsetAsync(zero_.p, 0, sizeof(uint32_t) * zero_words_, st_));   // one block: barrier counter, digit histograms, tile status
  uint32_t G = std::max<uint32_t>(1u, std::min<uint32_t>(a.ntiles, std::max<uint32_t>(slots_ / 2, 1u)));
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
    throw Error(LOAMX_E_HIP, "voxel grid: a grid barrier of k_vox_ds timed out (device over-subscribed?)");
  }
}

}  // namespace loamx
