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
  uint32_t* cnt;
  uint32_t* heads;
  uint32_t* ctl;                // [0] fail epoch, [1] claim counter
  uint32_t* h_fail;             // pinned: [0] fail epoch, [1] timeout
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

// ----------------------------------------------------------------------------------------------------------------
// k_vb_plan: grid = segments, 1024 threads
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_vb_plan(const VbArgs A) {
  __shared__ uint32_t s_hist[VB_BINS];
  __shared__ uint32_t s_first[VB_MAXBUCK], s_last[VB_MAXBUCK];
  __shared__ int s_mm[6];
  __shared__ uint32_t s_scan[17];
  __shared__ uint32_t s_bad;
  const int tid = (int)threadIdx.x;
  const uint32_t seg = blockIdx.x;
  const uint32_t a0 = A.seg_off[seg], a1 = A.seg_off[seg + 1], ns = a1 - a0;
  const float4* __restrict__ pts = A.src ? A.src[seg] : A.in + a0;
  const float inv = (seg & 1) ? A.inv_odd : A.inv_even;
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
  if (tid < 6) s_mm[tid] = tid < 3 ? 2147483647 : (-2147483647 - 1);
  if (tid == 0) s_bad = nbuckets > (uint32_t)VB_MAXBUCK ? 2u : 0u;   // bit 0: no voxel for a coordinate, bit 1: too many buckets, bit 2: a bin too large
  for (uint32_t e = (uint32_t)tid; e < (uint32_t)VB_BINS; e += 1024) s_hist[e] = 0u;
  for (uint32_t e = (uint32_t)tid; e < (uint32_t)VB_MAXBUCK; e += 1024) { s_first[e] = 0xffffffffu; s_last[e] = 0u; }
  __syncthreads();
  // ---- box of the untransformed points, voxel units
  {
    int mn[3] = {2147483647, 2147483647, 2147483647}, mx[3] = {-2147483647 - 1, -2147483647 - 1, -2147483647 - 1};
    bool bad = false;
    for (uint32_t i0 = (uint32_t)tid; i0 < ns; i0 += 8 * 1024) {   // eight loads in flight per thread: one workgroup walks a whole segment
      float4 p[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + (uint32_t)u * 1024u;
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
    if ((tid & 63) == 0) {
#pragma unroll
      for (int a = 0; a < 3; a++) { atomicMin(&s_mm[a], mn[a]); atomicMax(&s_mm[3 + a], mx[a]); }
    }
    if (bad) atomicOr(&s_bad, 1u);
  }
  __syncthreads();
  VbSeg P;
  P.bucket0 = bucket0;
  P.nbuckets = nbuckets;
  P.pos_bits = vb_bits(ns ? (unsigned long long)(ns - 1) : 0ull);
  if (P.pos_bits == 0) P.pos_bits = 1;
  unsigned long long nkeys = 1ull;
  bool seg_bad = s_bad != 0u, box_bad = false;   // block-uniform (every thread derives the same plan from the shared bounds)
  if (ns != 0 && !seg_bad) {
    unsigned long long d[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      P.mn[a] = s_mm[a] - 1;   // one voxel of margin on every side for the strays of the round trip
      d[a] = (unsigned long long)((long long)s_mm[3 + a] - (long long)s_mm[a] + 3);
      P.dim[a] = (uint32_t)d[a];
    }
    // a padded box beyond INT_MAX voxels goes to the general kernel, which applies PCL's own (unpadded) pass-through test
    if (d[0] * d[1] > 2147483647ull || d[0] * d[1] * d[2] > 2147483647ull) seg_bad = box_bad = true;
    else nkeys = d[0] * d[1] * d[2];
  }
  if (ns == 0 || seg_bad) {   // nothing can fall into an empty box: a point of a given-up segment raises the fail word again, harmlessly
    P.mn[0] = P.mn[1] = P.mn[2] = 0;
    P.dim[0] = P.dim[1] = P.dim[2] = 0u;
    nkeys = 1ull;
  }
  const uint32_t kb = vb_bits(nkeys - 1ull);
  P.shift = kb > (uint32_t)VB_BIN_BITS ? kb - (uint32_t)VB_BIN_BITS : 0u;
  const uint32_t nbins = (uint32_t)((nkeys - 1ull) >> P.shift) + 1u;   // <= VB_BINS
  __syncthreads();
  // ---- histogram of the predicted linear voxel indices
  if (!seg_bad) {
    for (uint32_t i0 = (uint32_t)tid; i0 < ns; i0 += 8 * 1024) {
      float4 p[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + (uint32_t)u * 1024u;
        p[u] = pts[i < ns ? i : i0];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (i0 + (uint32_t)u * 1024u >= ns) continue;
        int v[3];
        (void)vb_voxel(p[u].x, inv, v[0]); (void)vb_voxel(p[u].y, inv, v[1]); (void)vb_voxel(p[u].z, inv, v[2]);
        const uint32_t key = (uint32_t)(v[0] - P.mn[0]) + ((uint32_t)(v[1] - P.mn[1]) + (uint32_t)(v[2] - P.mn[2]) * P.dim[1]) * P.dim[0];
        atomicAdd(&s_hist[key >> P.shift], 1u);
      }
    }
  }
  __syncthreads();
  // ---- bin -> bucket: bucket = exclusive prefix / T (thread t owns VB_BINS / 1024 consecutive bins).  A bucket's index range covers
  // every bin mapped to it, empty ones included: a stray may land in a bin the prediction left empty.
  {
    constexpr int BPT = VB_BINS / 1024;
    uint32_t h[BPT], sum = 0;
    bool big = false;
#pragma unroll
    for (int k = 0; k < BPT; k++) {
      h[k] = s_hist[BPT * tid + k];
      sum += h[k];
      big = big || h[k] > (uint32_t)VB_MAXBIN;
    }
    if (big) atomicOr(&s_bad, 4u);
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
  const bool bad = seg_bad || s_bad != 0u;   // (s_bad: also a bin too large for a bucket)
  if (bad && tid == 0) vb_fail(A, box_bad ? 2 : ((s_bad & 1u) ? 0 : ((s_bad & 2u) ? 1 : 3)));
  for (uint32_t k = (uint32_t)tid; k < nbuckets; k += 1024) {
    VbBucket B;
    B.seg = seg;
    B.pad = 0u;
    if (bad || s_first[k] == 0xffffffffu) { B.key_lo = 0u; B.key_bits = 1u; }
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
  // one atomic per run of lanes that go to the same bucket
  unsigned long long rem = __ballot(act);
  uint32_t slot = 0u;
  while (rem) {
    const int l0 = __builtin_ctzll(rem);
    const uint32_t g0 = (uint32_t)__shfl((int)g, l0, 64);
    const unsigned long long m = __ballot(act && g == g0);
    uint32_t base = 0u;
    if (lane == l0) base = atomicAdd(&A.cnt[g0], (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, l0, 64);
    if (act && g == g0) slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    rem &= ~m;
  }
  if (act) {
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
  __shared__ unsigned long long s_w[VB_CAP];     // the sort buffer (one: every thread holds its eight words in registers while a pass scatters)
  __shared__ float4 s_pts[VB_CAP + VB_CAP / 8];
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
  const VbSeg P = A.segs[bk.seg];
  const uint32_t c = min(A.cnt[b], (uint32_t)VB_CAP);
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
#pragma unroll
    for (int k = 0; k < VB_WAVES; k++)
      if (tid < 256) s_wcnt[k][tid] = 0u;
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
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
      w[j] = i < c ? s_w[i] : ~0ull;
    }
  }
  // (npass >= 1: s_w holds the sorted words.)  Every thread takes eight CONSECUTIVE sorted elements; voxel ids go back into the
  // sort buffer's memory, the points are staged in sorted order.
  auto lp = [](uint32_t l) { return l + (l >> 3); };   // one pad slot per eight: a thread's eight consecutive elements start in distinct banks
  const uint32_t l0 = (uint32_t)tid * 8u;
  uint32_t vox[8];
  {
    unsigned long long x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t l = (uint32_t)(j * VB_THREADS + tid);
      x[j] = l < c ? s_w[l] : 0ull;
    }
    __syncthreads();
    uint32_t* s_vox = (uint32_t*)s_w;   // VB_CAP + VB_CAP / 8 words fit the 2 * VB_CAP words of the sort buffer
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t l = (uint32_t)(j * VB_THREADS + tid);
      if (l < c) {
        s_vox[lp(l)] = (uint32_t)(x[j] >> pbits);
        s_pts[lp(l)] = A.stack[sbeg + (uint32_t)(x[j] & pmask)];
      }
    }
  }
  __syncthreads();
  const uint32_t* s_vox = (const uint32_t*)s_w;
  bool head[8];
  uint32_t nh = 0u;
  uint32_t prev = (l0 == 0u || l0 > c) ? 0xffffffffu : s_vox[lp(l0 - 1u)];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t l = l0 + (uint32_t)j;
    vox[j] = l < c ? s_vox[lp(l)] : 0xffffffffu;
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
      const uint32_t l = l0 + (uint32_t)j;
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
      uint32_t cntp = 0u, q = l;
      do {
        const float4 t = s_pts[lp(q)];
        sx += t.x; sy += t.y; sz += t.z; si += t.w;
        cntp++;
        q++;
      } while (q < c && s_vox[lp(q)] == vox[j]);
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
  uint32_t nb = 0;
  for (uint32_t s = 0; s < nseg; s++) {
    const uint32_t m = h_seg_off[s + 1] - h_seg_off[s];
    nb += m ? (m + VB_T - 1) / VB_T : 1u;
  }
  nb_ = nb;
  segs_.reserve(nseg);
  buckets_.reserve(nb);
  bin2bucket_.reserve((size_t)nseg * VB_BINS);
  cnt_.reserve(nb);
  heads_.reserve(nb);
  elems_.reserve((size_t)nb * VB_CAP);
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
  a.segs = segs_.p; a.buckets = buckets_.p; a.bin2bucket = bin2bucket_.p; a.cnt = cnt_.p; a.heads = heads_.p; a.ctl = ctl_.p;
  a.h_fail = h_fail_.p; a.elems = elems_.p; a.stack = stack; a.out = out; a.out_off = d_out_off;
  a.n = n; a.nseg = nseg; a.nb = nb; a.epoch = epoch_; a.claim_base = claim_base_;
  a.inv_even = inv_even; a.inv_odd = inv_odd;
  hipLaunchKernelGGL(k_vb_plan, dim3(nseg), dim3(1024), 0, st_, a);
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
