// Bucketed voxel grid of the registration's stack clouds (see voxbucket.cuh).  gfx950, wave64.
#include "voxbucket.cuh"
#include "scan.cuh"
#include "voxel.cuh"

namespace loamx {

struct VbArgs {
  const float4* in;             // concatenated input points, or
  const float4* const* src;     // one pointer per segment
  const uint32_t* seg_off;      // [nseg + 1]
  const Pose* poses;            // per sweep (segment / 2)
  VbSeg* segs;
  VbBucket* buckets;
  uint16_t* bin2bucket;
  int* mm;                      // [nseg][6] voxel bounds of the untransformed points (min x y z, max x y z)
  uint32_t* hist;               // [nseg][VB_BINS], zero between runs (k_vb_scan clears what it has read)
  uint32_t* cnt;
  uint32_t* heads;
  uint32_t* ctl;                // [0] fail epoch, [1] claim counter
  uint32_t* h_fail;             // pinned: [0] fail epoch, [1] timeout, [2 + r] epoch of the last run that met reason r
  unsigned long long* elems;
  float4* stack;
  float4* out;
  uint32_t* out_off;
  uint32_t n, nseg, nb, epoch, claim_base;
  float inv_even, inv_odd;
};

__device__ inline uint32_t vb_bits(unsigned long long v) { return v ? 64u - (uint32_t)__builtin_clzll(v) : 0u; }
// reasons (h_fail[2 + reason] = epoch, for the diagnostics of VoxBucket::why()): 0 a coordinate without a voxel, 1 more buckets
// than a segment may have, 2 padded box beyond INT_MAX voxels, 3 a histogram bin larger than a bucket takes, 4 a point outside
// the predicted box, 5 a bucket over capacity
__device__ inline void vb_fail(const VbArgs& A, int reason) {
  A.ctl[0] = A.epoch;      // (every writer stores the same value)
  A.h_fail[0] = A.epoch;
  A.h_fail[2 + reason] = A.epoch;
}
// voxel coordinate of one axis exactly as pcl::VoxelGrid forms it: floor(v * inverse leaf), float arithmetic
__device__ inline bool vb_voxel(float v, float inv, int& i) {
  const float f = floorf(v * inv);
  if (!(fabsf(f) < 1.0e9f)) return false;   // also catches NaN / inf
  i = (int)f;
  return true;
}
constexpr int VB_BAD = 2147483647;   // mm[seg][3] (max x) of a segment with a coordinate that has no voxel

// the padded box of a segment and what follows from it; false: the segment cannot take the bucketed path (reason)
__device__ inline bool vb_box(const int* __restrict__ mm, uint32_t ns, VbSeg& P, int& reason) {
  P.pos_bits = vb_bits(ns ? (unsigned long long)(ns - 1) : 0ull);
  if (P.pos_bits == 0) P.pos_bits = 1;
  P.mn[0] = P.mn[1] = P.mn[2] = 0;
  P.dim[0] = P.dim[1] = P.dim[2] = 0u;   // nothing falls into an empty box
  P.shift = 0u;
  P.bucket0 = P.nbuckets = 0u;
  reason = -1;
  if (ns == 0) return true;
  if (mm[3] == VB_BAD || mm[3] < mm[0]) { reason = 0; return false; }
  unsigned long long d[3];
#pragma unroll
  for (int a = 0; a < 3; a++) d[a] = (unsigned long long)((long long)mm[3 + a] - (long long)mm[a] + 3);   // one voxel of margin on every side
  // a padded box beyond INT_MAX voxels goes to the general kernel, which applies PCL's own (unpadded) pass-through test
  if (d[0] * d[1] > 2147483647ull || d[0] * d[1] * d[2] > 2147483647ull) { reason = 2; return false; }
#pragma unroll
  for (int a = 0; a < 3; a++) { P.mn[a] = mm[a] - 1; P.dim[a] = (uint32_t)d[a]; }
  const uint32_t kb = vb_bits(d[0] * d[1] * d[2] - 1ull);
  P.shift = kb > (uint32_t)VB_BIN_BITS ? kb - (uint32_t)VB_BIN_BITS : 0u;
  return true;
}

// ----------------------------------------------------------------------------------------------------------------
// plan, step 1 — k_vb_bounds: voxel box of the untransformed points of every segment.  grid = (blocks, segments), 256 threads
// ----------------------------------------------------------------------------------------------------------------
__global__ void k_vb_bounds_init(int* mm, uint32_t nseg) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * nseg) mm[i] = (i % 6) < 3 ? 2147483647 : (-2147483647 - 1);
}
__global__ __launch_bounds__(256) void k_vb_bounds(const VbArgs A) {
  const uint32_t seg = blockIdx.y;
  const uint32_t a0 = A.seg_off[seg], ns = A.seg_off[seg + 1] - a0;
  if (blockIdx.x * 2048u >= ns) return;
  const float4* __restrict__ pts = A.src ? A.src[seg] : A.in + a0;
  const float inv = (seg & 1) ? A.inv_odd : A.inv_even;
  int mn[3] = {2147483647, 2147483647, 2147483647}, mx[3] = {-2147483647 - 1, -2147483647 - 1, -2147483647 - 1};
  bool bad = false;
  for (uint32_t i0 = blockIdx.x * 2048u + threadIdx.x; i0 < ns; i0 += gridDim.x * 2048u) {
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t i = i0 + 256u * (uint32_t)u;
      p[u] = pts[i < ns ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int v[3];
      if (!(vb_voxel(p[u].x, inv, v[0]) && vb_voxel(p[u].y, inv, v[1]) && vb_voxel(p[u].z, inv, v[2]))) { bad = true; continue; }
#pragma unroll
      for (int a = 0; a < 3; a++) { mn[a] = v[a] < mn[a] ? v[a] : mn[a]; mx[a] = v[a] > mx[a] ? v[a] : mx[a]; }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const int lo = __shfl_xor(mn[a], d, 64), hi = __shfl_xor(mx[a], d, 64);
      mn[a] = lo < mn[a] ? lo : mn[a];
      mx[a] = hi > mx[a] ? hi : mx[a];
    }
  }
  __shared__ int s_mm[4][6];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (__any(bad)) mx[0] = VB_BAD;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { s_mm[wid][a] = mn[a]; s_mm[wid][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = (int)threadIdx.x;
    int v = s_mm[0][a];
    for (int w = 1; w < 4; w++) v = a < 3 ? (s_mm[w][a] < v ? s_mm[w][a] : v) : (s_mm[w][a] > v ? s_mm[w][a] : v);
    if (a < 3) atomicMin(&A.mm[6 * seg + a], v); else atomicMax(&A.mm[6 * seg + a], v);
  }
}

// ----------------------------------------------------------------------------------------------------------------
// plan, step 2 — k_vb_hist: histogram of the predicted linear voxel indices.  grid = (blocks, segments), 256 threads
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vb_hist(const VbArgs A) {
  const uint32_t seg = blockIdx.y;
  const uint32_t a0 = A.seg_off[seg], ns = A.seg_off[seg + 1] - a0;
  if (blockIdx.x * 2048u >= ns) return;
  VbSeg P;
  int reason;
  if (!vb_box(A.mm + 6 * seg, ns, P, reason)) return;   // (k_vb_scan raises the fail word)
  const float4* __restrict__ pts = A.src ? A.src[seg] : A.in + a0;
  const float inv = (seg & 1) ? A.inv_odd : A.inv_even;
  uint32_t* __restrict__ hist = A.hist + (size_t)seg * VB_BINS;
  for (uint32_t i0 = blockIdx.x * 2048u + threadIdx.x; i0 < ns; i0 += gridDim.x * 2048u) {
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t i = i0 + 256u * (uint32_t)u;
      p[u] = pts[i < ns ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (i0 + 256u * (uint32_t)u >= ns) continue;
      int v[3];
      (void)vb_voxel(p[u].x, inv, v[0]); (void)vb_voxel(p[u].y, inv, v[1]); (void)vb_voxel(p[u].z, inv, v[2]);
      const uint32_t key = (uint32_t)(v[0] - P.mn[0]) + ((uint32_t)(v[1] - P.mn[1]) + (uint32_t)(v[2] - P.mn[2]) * P.dim[1]) * P.dim[0];
      atomicAdd(&hist[key >> P.shift], 1u);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// plan, step 3 — k_vb_scan: bins -> buckets.  grid = segments, 1024 threads (thread t owns VB_BINS / 1024 consecutive bins)
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_vb_scan(const VbArgs A) {
  __shared__ uint32_t s_first[VB_MAXBUCK], s_last[VB_MAXBUCK];
  __shared__ uint32_t s_scan[17];
  __shared__ uint32_t s_bad;
  const int tid = (int)threadIdx.x;
  const uint32_t seg = blockIdx.x;
  const uint32_t ns = A.seg_off[seg + 1] - A.seg_off[seg];
  // first bucket of this segment: every segment owns max(1, ceil(points / T)) buckets
  uint32_t bucket0;
  {
    uint32_t part = 0;
    for (uint32_t s = (uint32_t)tid; s < seg; s += 1024) {
      const uint32_t m = A.seg_off[s + 1] - A.seg_off[s];
      part += m ? (m + VB_T - 1) / VB_T : 1u;
    }
    uint32_t tot;
    (void)block_excl_scan(part, s_scan, tot);
    bucket0 = tot;
  }
  const uint32_t nbuckets = ns ? (ns + VB_T - 1) / VB_T : 1u;
  VbSeg P;
  int reason;
  const bool box_ok = vb_box(A.mm + 6 * seg, ns, P, reason);   // block-uniform (k_vb_hist counted this segment's points iff box_ok)
  bool seg_bad = !box_ok;
  if (!seg_bad && nbuckets > (uint32_t)VB_MAXBUCK) { seg_bad = true; reason = 1; }
  if (seg_bad) P.dim[0] = P.dim[1] = P.dim[2] = 0u;   // a point of a given-up segment raises the fail word again, harmlessly
  P.bucket0 = bucket0;
  P.nbuckets = nbuckets;
  unsigned long long nkeys = (unsigned long long)P.dim[0] * P.dim[1] * P.dim[2];
  if (nkeys == 0ull) nkeys = 1ull;
  const uint32_t nbins = (uint32_t)((nkeys - 1ull) >> P.shift) + 1u;   // <= VB_BINS
  if (tid == 0) s_bad = 0u;
  for (uint32_t e = (uint32_t)tid; e < (uint32_t)VB_MAXBUCK; e += 1024) { s_first[e] = 0xffffffffu; s_last[e] = 0u; }
  __syncthreads();
  // bucket of a bin = exclusive prefix / T.  A bucket's index range covers every bin mapped to it, empty ones included: a stray
  // may land in a bin the prediction left empty.
  {
    constexpr int BPT = VB_BINS / 1024;
    uint32_t* __restrict__ hist = A.hist + (size_t)seg * VB_BINS + BPT * tid;
    uint32_t h[BPT], sum = 0;
    bool big = false;
#pragma unroll
    for (int k = 0; k < BPT; k++) {
      h[k] = (box_ok && ns) ? hist[k] : 0u;
      sum += h[k];
      big = big || h[k] > (uint32_t)VB_MAXBIN;
    }
#pragma unroll
    for (int k = 0; k < BPT; k++)
      if (h[k]) hist[k] = 0u;   // ready for the next run
    if (big) s_bad = 1u;
    uint32_t tot;
    uint32_t ex = block_excl_scan(sum, s_scan, tot);
    uint16_t* tab = A.bin2bucket + (size_t)seg * VB_BINS + BPT * tid;
    uint32_t run_b = 0xffffffffu, run_first = 0u, run_last = 0u;
#pragma unroll
    for (int k = 0; k < BPT; k++) {
      const uint32_t bin = (uint32_t)(BPT * tid + k);
      if (bin < nbins && !seg_bad) {
        uint32_t b = ex / (uint32_t)VB_T;
        if (b >= nbuckets) b = nbuckets - 1;   // empty bins behind the last point when the segment fills its buckets exactly
        tab[k] = (uint16_t)b;
        if (b != run_b) {
          if (run_b != 0xffffffffu) { atomicMin(&s_first[run_b], run_first); atomicMax(&s_last[run_b], run_last); }
          run_b = b;
          run_first = bin;
        }
        run_last = bin;
      }
      ex += h[k];
    }
    if (run_b != 0xffffffffu) { atomicMin(&s_first[run_b], run_first); atomicMax(&s_last[run_b], run_last); }
  }
  __syncthreads();
  if (!seg_bad && s_bad) { seg_bad = true; reason = 3; }
  if (seg_bad && tid == 0) vb_fail(A, reason);
  for (uint32_t k = (uint32_t)tid; k < nbuckets; k += 1024) {
    VbBucket B;
    B.seg = seg;
    B.pad = 0u;
    if (seg_bad || k >= (uint32_t)VB_MAXBUCK || s_first[k] == 0xffffffffu) { B.key_lo = 0u; B.key_bits = 1u; }
    else {
      B.key_lo = s_first[k] << P.shift;
      const unsigned long long span = ((unsigned long long)(s_last[k] - s_first[k]) + 1ull) << P.shift;
      B.key_bits = vb_bits(span - 1ull);
      if (B.key_bits == 0u) B.key_bits = 1u;
    }
    A.buckets[bucket0 + k] = B;
    A.cnt[bucket0 + k] = 0u;
    A.heads[bucket0 + k] = 0u;
  }
  if (tid == 0) A.segs[seg] = P;
}

// ----------------------------------------------------------------------------------------------------------------
// k_vb_stack: grid = ceil(n / 256), 256 threads
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vb_stack(const VbArgs A) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  bool act = i < A.n;
  uint32_t g = 0u;
  unsigned long long elem = 0ull;
  if (act) {
    const uint32_t seg = vox_find_seg(A.seg_off, A.nseg, i);
    const uint32_t a0 = A.seg_off[seg];
    const Pose T = A.poses[seg >> 1];
    const float4 p = A.src ? A.src[seg][i - a0] : A.in[i];
    float x = p.x, y = p.y, z = p.z;
    to_map(T, x, y, z);          // BasicLaserMapping.cpp:282-292 via :512-516
    to_be_mapped(T, x, y, z);
    A.stack[i] = make_float4(x, y, z, p.w);
    const VbSeg P = A.segs[seg];
    const float inv = (seg & 1) ? A.inv_odd : A.inv_even;
    int v[3];
    bool ok = vb_voxel(x, inv, v[0]) && vb_voxel(y, inv, v[1]) && vb_voxel(z, inv, v[2]);
    uint32_t r[3] = {0u, 0u, 0u};
    if (ok) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const long long d = (long long)v[a] - (long long)P.mn[a];
        ok = ok && d >= 0 && d < (long long)P.dim[a];
        r[a] = (uint32_t)d;
      }
    }
    if (!ok) {   // outside the predicted box (or the plan gave the segment up: its box is empty)
      vb_fail(A, 4);
      act = false;
    } else {
      const uint32_t key = r[0] + (r[1] + r[2] * P.dim[1]) * P.dim[0];
      g = P.bucket0 + (uint32_t)A.bin2bucket[(size_t)seg * VB_BINS + (key >> P.shift)];
      elem = ((unsigned long long)key << 24) | (unsigned long long)(i - a0);
    }
  }
  // Slots: the lanes of a wave that go to the same bucket are counted together, the waves of the workgroup meet in an LDS table
  // (256 entries, bucket % 256 with a tag: the buckets a workgroup touches are a few neighbours of one or two segments), and ONE
  // thread per touched bucket bumps the global counter — hundreds of waves hammering ~150 counters serialise in L2 otherwise.
  // A bucket that finds its table entry taken by another one goes to the global counter directly.
  __shared__ uint32_t s_tag[256], s_cnt[256], s_gbase[256];
  s_tag[threadIdx.x] = 0xffffffffu;
  s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  unsigned long long rem = __ballot(act);
  uint32_t slot = 0u;      // rank inside the workgroup's share of the bucket (table path) or the final slot (direct path)
  bool direct = false;
  while (rem) {
    const int l0 = __builtin_ctzll(rem);
    const uint32_t g0 = (uint32_t)__shfl((int)g, l0, 64);
    const unsigned long long m = __ballot(act && g == g0);
    uint32_t base = 0u, dir = 0u;
    if (lane == l0) {
      const uint32_t h = g0 & 255u;
      const uint32_t old = atomicCAS(&s_tag[h], 0xffffffffu, g0);
      if (old == 0xffffffffu || old == g0) base = atomicAdd(&s_cnt[h], (uint32_t)__popcll(m));
      else { base = atomicAdd(&A.cnt[g0], (uint32_t)__popcll(m)); dir = 1u; }
    }
    base = (uint32_t)__shfl((int)base, l0, 64);
    dir = (uint32_t)__shfl((int)dir, l0, 64);
    if (act && g == g0) { slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); direct = dir != 0u; }
    rem &= ~m;
  }
  __syncthreads();
  if (s_tag[threadIdx.x] != 0xffffffffu) s_gbase[threadIdx.x] = atomicAdd(&A.cnt[s_tag[threadIdx.x]], s_cnt[threadIdx.x]);
  __syncthreads();
  if (act) {
    if (!direct) slot += s_gbase[g & 255u];
    if (slot < (uint32_t)VB_CAP) A.elems[(size_t)g * VB_CAP + slot] = elem;
    else vb_fail(A, 5);
  }
}

// ----------------------------------------------------------------------------------------------------------------
// k_vb_reduce: grid = buckets, VB_CAP / 8 threads, 8 elements per thread
// ----------------------------------------------------------------------------------------------------------------
constexpr uint32_t VB_SPIN_LIMIT = 1u << 20;
constexpr int VB_THREADS = VB_CAP / 8, VB_WAVES = VB_THREADS / 64;

__global__ __launch_bounds__(VB_THREADS) void k_vb_reduce(const VbArgs A) {
  __shared__ unsigned long long s_w[VB_CAP + 1];   // the sort buffer (one: every thread holds its eight words in registers while a pass scatters)
  __shared__ uint32_t s_wcnt[VB_WAVES][256];
  __shared__ uint32_t s_base[256];
  __shared__ uint32_t s_scan[17];
  __shared__ uint32_t s_bidx;
  const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // buckets are taken in the order the workgroups start, so everything a look-back waits for is already running
  if (tid == 0) s_bidx = atomicAdd(&A.ctl[1], 1u) - A.claim_base;
  __syncthreads();
  const uint32_t b = s_bidx;
  if (b >= A.nb) return;   // (cannot happen: the grid has exactly nb workgroups)
  if (__hip_atomic_load(&A.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.epoch) {   // given up: an empty, well-formed result
    if (b == 0)
      for (uint32_t s = (uint32_t)tid; s <= A.nseg; s += VB_THREADS) A.out_off[s] = 0u;
    return;
  }
  const VbBucket bk = A.buckets[b];
  const uint32_t c = min(A.cnt[b], (uint32_t)VB_CAP);
  const VbSeg P = A.segs[bk.seg];
  const uint32_t sbeg = A.seg_off[bk.seg];
  const unsigned long long* __restrict__ e = A.elems + (size_t)b * VB_CAP;
  const uint32_t pbits = P.pos_bits;
  const unsigned long long pmask = (1ull << pbits) - 1ull;
  const uint32_t total_bits = bk.key_bits + pbits;
  const uint32_t npass = (total_bits + 7u) / 8u;
  // sort word: (linear voxel index - bucket base) << pos_bits | input position inside the segment — all distinct
  unsigned long long w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
    if (i < c) {
      const unsigned long long x = e[i];
      w[j] = (((x >> 24) - (unsigned long long)bk.key_lo) << pbits) | (x & 0xffffffull);
    } else {
      w[j] = ~0ull;
    }
  }
  const bool wave_has = (uint32_t)(wid * 512) < c;   // wave-uniform: a wave whose slots are all beyond the end only keeps the barriers
  for (uint32_t p = 0; p < npass; p++) {
    const uint32_t shift = 8u * p;
    if (tid < 256) {
#pragma unroll
      for (int k = 0; k < VB_WAVES; k++) s_wcnt[k][tid] = 0u;
    }
    __syncthreads();
    uint32_t rank[8];
#pragma unroll
    for (int j = 0; j < 8; j++) rank[j] = 0u;
    if (wave_has) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
        const bool in = i < c;
        const uint32_t d = (uint32_t)((w[j] >> shift) & 255ull);
        unsigned long long m = __ballot(in);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
          const unsigned long long bal = __ballot((d >> bit) & 1u);
          m &= ((d >> bit) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        const int leader = __builtin_ctzll(m | (1ull << 63));
        uint32_t old = 0u;
        if (in && lane == leader) {
          old = s_wcnt[wid][d];
          s_wcnt[wid][d] = old + (uint32_t)__popcll(m);
        }
        old = (uint32_t)__shfl((int)old, leader, 64);
        rank[j] = old + (uint32_t)__popcll(below);
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    {   // thread d < 256: exclusive prefix over the waves, then (all threads) over the digits
      uint32_t run = 0u;
      if (tid < 256) {
#pragma unroll
        for (int k = 0; k < VB_WAVES; k++) { const uint32_t v = s_wcnt[k][tid]; s_wcnt[k][tid] = run; run += v; }
      }
      uint32_t tot;
      const uint32_t ex = block_excl_scan(run, s_scan, tot);
      if (tid < 256) s_base[tid] = ex;
    }
    __syncthreads();   // (also: every thread has its words in registers, the buffer may be overwritten)
    if (wave_has) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
        if (i < c) {
          const uint32_t d = (uint32_t)((w[j] >> shift) & 255ull);
          s_w[s_base[d] + s_wcnt[wid][d] + rank[j]] = w[j];
        }
      }
    }
    __syncthreads();
    if (p + 1 < npass) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
        w[j] = i < c ? s_w[i] : ~0ull;
      }
    }
  }
  // (npass >= 1: s_w holds the sorted words.)  Every thread takes eight CONSECUTIVE sorted elements and fetches their points
  // itself (eight independent gathers in flight); a voxel's mean is the sequential sum of its run in sorted = input order: the
  // part of a run inside its head's thread comes out of registers, a run that goes on into the following threads' elements is
  // continued through LDS (the words) and global memory (the points, just fetched by the neighbour: cache hits).
  if (tid == 0) s_w[c] = ~0ull;   // sentinel behind the last element
  __syncthreads();
  const uint32_t l0 = (uint32_t)tid * 8u;
  uint32_t vox[8];
  float4 pt[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t l = l0 + (uint32_t)j;
    const unsigned long long x = l < c ? s_w[l] : ~0ull;
    vox[j] = l < c ? (uint32_t)(x >> pbits) : 0xffffffffu;
    pt[j] = l < c ? A.stack[sbeg + (uint32_t)(x & pmask)] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  bool head[8];
  uint32_t nh = 0u;
  uint32_t prev = (l0 == 0u || l0 > c) ? 0xffffffffu : (uint32_t)(s_w[l0 - 1u] >> pbits);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t l = l0 + (uint32_t)j;
    head[j] = l < c && (l == 0u || vox[j] != prev);   // (a voxel never straddles two buckets: buckets are index ranges)
    nh += head[j] ? 1u : 0u;
    prev = vox[j];
  }
  uint32_t tot;
  const uint32_t ex = block_excl_scan(nh, s_scan, tot);
  if (tid == 0) __hip_atomic_store(&A.heads[b], tot + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t part = 0u;
  for (uint32_t t = (uint32_t)tid; t < b; t += VB_THREADS) {
    uint32_t v = __hip_atomic_load(&A.heads[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), spins = 0u;
    while (v == 0u) {
      __builtin_amdgcn_s_sleep(2);
      v = __hip_atomic_load(&A.heads[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (++spins > VB_SPIN_LIMIT) { A.h_fail[1] = 1u; v = 1u; }
    }
    part += v - 1u;
  }
  uint32_t base;
  (void)block_excl_scan(part, s_scan, base);   // voxels emitted by all earlier buckets
  if (tid == 0) {
    if (P.bucket0 == b) A.out_off[bk.seg] = base;
    if (b + 1u == A.nb) A.out_off[A.nseg] = base + tot;
  }
  uint32_t pos = base + ex;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if (head[j]) {
      // (the reference's accumulators start at zero: 0 + x, kept for the sign of a zero coordinate)
      float sx = 0.f + pt[j].x, sy = 0.f + pt[j].y, sz = 0.f + pt[j].z, si = 0.f + pt[j].w;
      uint32_t cntp = 1u;
      bool open = true;   // the run is still going
#pragma unroll
      for (int q = j + 1; q < 8; q++) {
        open = open && vox[q] == vox[j];
        if (open) { sx += pt[q].x; sy += pt[q].y; sz += pt[q].z; si += pt[q].w; cntp++; }
      }
      if (open) {   // the run reaches the end of this thread's elements: it may go on
        uint32_t l = l0 + 8u;
        for (;;) {
          const unsigned long long x = s_w[l < c ? l : c];
          if (l >= c || (uint32_t)(x >> pbits) != vox[j]) break;
          const float4 t = A.stack[sbeg + (uint32_t)(x & pmask)];
          sx += t.x; sy += t.y; sz += t.z; si += t.w;
          cntp++;
          l++;
        }
      }
      const float cf = (float)cntp;
      A.out[pos] = make_float4(sx / cf, sy / cf, sz / cf, si / cf);
      pos++;
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
void VoxBucket::run(const float4* in, const float4* const* d_src, uint32_t n, const uint32_t* d_seg_off, const uint32_t* h_seg_off, uint32_t nseg,
                    const Pose* d_poses, float inv_even, float inv_odd, float4* stack, float4* out, uint32_t* d_out_off) {
  LX_REQUIRE(fits(n, nseg), "internal: VoxBucket::run outside its limits");
  uint32_t nb = 0, max_len = 0;
  for (uint32_t s = 0; s < nseg; s++) {
    const uint32_t m = h_seg_off[s + 1] - h_seg_off[s];
    nb += m ? (m + VB_T - 1) / VB_T : 1u;
    max_len = std::max(max_len, m);
  }
  nb_ = nb;
  segs_.reserve(nseg);
  buckets_.reserve(nb);
  bin2bucket_.reserve((size_t)nseg * VB_BINS);
  mm_.reserve((size_t)6 * nseg);
  cnt_.reserve(nb);
  heads_.reserve(nb);
  elems_.reserve((size_t)nb * VB_CAP);
  if (hist_.cap < (size_t)nseg * VB_BINS) {   // (a fresh allocation: cleared once; k_vb_scan leaves it cleared)
    hist_.reserve((size_t)nseg * VB_BINS);
    LX_HIP(hipMemsetAsync(hist_.p, 0, sizeof(uint32_t) * hist_.cap, st_));
  }
  if (!ctl_ready_) {
    ctl_.reserve(4);
    h_fail_.reserve(8);
    for (int k = 0; k < 8; k++) h_fail_.p[k] = 0u;
    LX_HIP(hipMemsetAsync(ctl_.p, 0, sizeof(uint32_t) * 4, st_));
    ctl_ready_ = true;
  }
  if (++epoch_ == 0u) epoch_ = 1u;
  VbArgs a;
  a.in = in; a.src = d_src; a.seg_off = d_seg_off; a.poses = d_poses;
  a.segs = segs_.p; a.buckets = buckets_.p; a.bin2bucket = bin2bucket_.p; a.mm = mm_.p; a.hist = hist_.p; a.cnt = cnt_.p; a.heads = heads_.p;
  a.ctl = ctl_.p; a.h_fail = h_fail_.p; a.elems = elems_.p; a.stack = stack; a.out = out; a.out_off = d_out_off;
  a.n = n; a.nseg = nseg; a.nb = nb; a.epoch = epoch_; a.claim_base = claim_base_;
  a.inv_even = inv_even; a.inv_odd = inv_odd;
  const uint32_t bx = std::max(1u, std::min(16u, (max_len + 2047u) / 2048u));   // blocks per segment of the two passes over the points
  hipLaunchKernelGGL(k_vb_bounds_init, dim3((6 * nseg + 255) / 256), dim3(256), 0, st_, mm_.p, nseg);
  hipLaunchKernelGGL(k_vb_bounds, dim3(bx, nseg), dim3(256), 0, st_, a);
  hipLaunchKernelGGL(k_vb_hist, dim3(bx, nseg), dim3(256), 0, st_, a);
  hipLaunchKernelGGL(k_vb_scan, dim3(nseg), dim3(1024), 0, st_, a);
  hipLaunchKernelGGL(k_vb_stack, dim3((n + 255) / 256), dim3(256), 0, st_, a);
  hipLaunchKernelGGL(k_vb_reduce, dim3(nb), dim3(VB_THREADS), 0, st_, a);
  LX_HIP(hipGetLastError());
  claim_base_ += nb;   // (wraps together with the device counter)
}

// after failed(): which conditions the last run met (bit r = reason r of vb_fail)
uint32_t VoxBucket::why() const {
  uint32_t m = 0;
  for (int r = 0; r < 6; r++)
    if (((volatile uint32_t*)h_fail_.p)[2 + r] == epoch_) m |= 1u << r;
  return m;
}

void VoxBucket::check() {
  if (h_fail_.p && ((volatile uint32_t*)h_fail_.p)[1]) {
    h_fail_.p[1] = 0u;
    throw Error(LOAMX_E_HIP, "voxel grid: a look-back wait inside k_vb_reduce timed out");
  }
}

}  // namespace loamx
