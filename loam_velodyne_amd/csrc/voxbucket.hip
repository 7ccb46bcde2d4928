// Bucketed voxel grid of the registration's stack clouds (see voxbucket.hpp).  gfx950, wave64.
#include "voxbucket.hpp"
#include "scan.hpp"
#include "voxel.hpp"

namespace loamx {

struct VbArgs {
  const float4* in;             // concatenated input points, or
  const float4* const* src;     // one pointer per segment
  const uint32_t* seg_off;      // [nseg + 1]
  const Pose* poses;            // per sweep (segment / 2)
  VbSeg* segs;
  unsigned long long* lo;       // per bucket: smallest voxel key
  uint32_t* bseg;               // per bucket: segment
  int* box;                     // [nseg][6]
  uint32_t* cnt;
  uint32_t* heads;
  uint32_t* ctl;                // [0] fail epoch
  uint32_t* h_fail;             // pinned: [0] fail epoch, [1] timeout, [2 + r] epoch of the last run that met reason r
  uint32_t* elems;
  float4* stack;
  float4* out;
  uint32_t* out_off;
  uint32_t n, nseg, nb, epoch;
  float inv_even, inv_odd;
  unsigned long long* dbg;      // LOAMX_PROF_VB: per bucket 16 wall-clock stamps of k_vb_reduce
};

__device__ inline uint32_t vb_bits(unsigned long long v) { return v ? 64u - (uint32_t)__builtin_clzll(v) : 0u; }
// reasons (h_fail[2 + reason] = epoch, for the diagnostics of VoxBucket::why()): 0 a coordinate beyond +-2^20 voxels (or not finite),
// 1 more buckets than a segment may have, 2 segment box beyond INT_MAX voxels (PCL passes the cloud through), 3 a bucket whose own
// box needs more than 64 sort bits, 5 a bucket over capacity
__device__ inline void vb_fail(const VbArgs& A, int reason) {
  A.ctl[0] = A.epoch;      // (every writer stores the same value)
  A.h_fail[0] = A.epoch;
  A.h_fail[2 + reason] = A.epoch;
}
// voxel coordinate of one axis exactly as pcl::VoxelGrid forms it: floor(v * inverse leaf), float arithmetic
constexpr int VB_OFF = 1 << 20;
__device__ inline bool vb_voxel(float v, float inv, int& i) {
  const float f = floorf(v * inv);
  if (!(fabsf(f) < (float)VB_OFF)) return false;   // also catches NaN / inf
  i = (int)f;
  return true;
}
// voxel order = pcl::VoxelGrid's linear index order inside any box: (iz, iy, ix) lexicographically
__device__ inline unsigned long long vb_key(int ix, int iy, int iz) {
  return ((unsigned long long)(uint32_t)(iz + VB_OFF) << 42) | ((unsigned long long)(uint32_t)(iy + VB_OFF) << 21) | (unsigned long long)(uint32_t)(ix + VB_OFF);
}

// ----------------------------------------------------------------------------------------------------------------
// k_vb_plan: grid = segments, VB_SAMPLE threads
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VB_SAMPLE) void k_vb_plan(const VbArgs A) {
  __shared__ unsigned long long s_key[VB_SAMPLE], s_sorted[VB_SAMPLE];
  __shared__ uint32_t s_scan[17];
  __shared__ uint32_t s_bad;
  const int tid = (int)threadIdx.x;
  const uint32_t seg = blockIdx.x;
  const uint32_t a0 = A.seg_off[seg], ns = A.seg_off[seg + 1] - a0;
  const float4* __restrict__ pts = A.src ? A.src[seg] : A.in + a0;
  const float inv = (seg & 1) ? A.inv_odd : A.inv_even;
  // first bucket of this segment: every segment owns max(1, ceil(points / T)) buckets
  uint32_t bucket0;
  {
    uint32_t part = 0;
    for (uint32_t s = (uint32_t)tid; s < seg; s += VB_SAMPLE) {
      const uint32_t m = A.seg_off[s + 1] - A.seg_off[s];
      part += m ? (m + VB_T - 1) / VB_T : 1u;
    }
    uint32_t tot;
    (void)block_excl_scan(part, s_scan, tot);
    bucket0 = tot;
  }
  const uint32_t nbuckets = ns ? (ns + VB_T - 1) / VB_T : 1u;
  const uint32_t m = ns < (uint32_t)VB_SAMPLE ? ns : (uint32_t)VB_SAMPLE;
  if (tid == 0) s_bad = nbuckets > (uint32_t)VB_SAMPLE ? 2u : 0u;
  __syncthreads();
  unsigned long long key = ~0ull;
  if (nbuckets > 1u && (uint32_t)tid < m) {   // (one bucket needs no splitter)
    const float4 p = pts[(uint32_t)(((unsigned long long)tid * ns) / m)];
    int v[3];
    if (vb_voxel(p.x, inv, v[0]) && vb_voxel(p.y, inv, v[1]) && vb_voxel(p.z, inv, v[2])) key = vb_key(v[0], v[1], v[2]);
    else atomicOr(&s_bad, 1u);
  }
  s_key[tid] = key;
  __syncthreads();
  if (nbuckets > 1u && (uint32_t)tid < m) {   // rank among the sample (ties by sample position): every thread scans the sample through LDS broadcasts
    uint32_t r = 0;
    for (uint32_t u = 0; u < m; u++) {
      const unsigned long long k = s_key[u];
      r += (k < key || (k == key && u < (uint32_t)tid)) ? 1u : 0u;
    }
    s_sorted[r] = key;
  }
  __syncthreads();
  const uint32_t bad = s_bad;
  if (bad && tid == 0) vb_fail(A, (bad & 2u) ? 1 : 0);
  for (uint32_t k = (uint32_t)tid; k < nbuckets && k < (uint32_t)VB_SAMPLE; k += VB_SAMPLE) {
    A.lo[bucket0 + k] = (k == 0u || bad) ? 0ull : s_sorted[(uint32_t)(((unsigned long long)k * m) / nbuckets)];
    A.bseg[bucket0 + k] = seg;
    A.cnt[bucket0 + k] = 0u;
    A.heads[bucket0 + k] = 0u;
  }
  if (tid < 6) A.box[6 * seg + tid] = tid < 3 ? 2147483647 : (-2147483647 - 1);
  if (tid == 0) {
    VbSeg P;
    P.bucket0 = bucket0;
    P.nbuckets = bad ? 0u : nbuckets;   // a given-up segment takes no points (its points raise the fail word again, harmlessly)
    P.pos_bits = vb_bits(ns ? (unsigned long long)(ns - 1) : 0ull);
    if (P.pos_bits == 0) P.pos_bits = 1;
    P.pad = 0u;
    A.segs[seg] = P;
  }
}

// ----------------------------------------------------------------------------------------------------------------
// k_vb_stack: grid = ceil(n / 256), 256 threads
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vb_stack(const VbArgs A) {
  __shared__ unsigned long long s_lo[64];   // the splitters of the segment of the workgroup's first point
  __shared__ uint32_t s_tag[256], s_cnt[256], s_gbase[256];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t lead = vox_find_seg(A.seg_off, A.nseg, blockIdx.x * blockDim.x);
  const VbSeg PL = A.segs[lead];
  if (threadIdx.x < 64) s_lo[threadIdx.x] = threadIdx.x < PL.nbuckets ? A.lo[PL.bucket0 + threadIdx.x] : ~0ull;
  s_tag[threadIdx.x] = 0xffffffffu;
  s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  bool act = i < A.n;
  uint32_t g = 0u, pos = 0u;
  if (act) {
    const uint32_t seg = vox_find_seg(A.seg_off, A.nseg, i);
    const uint32_t a0 = A.seg_off[seg];
    const Pose T = A.poses[seg >> 1];
    const float4 p = A.src ? A.src[seg][i - a0] : A.in[i];
    float x = p.x, y = p.y, z = p.z;
    to_map(T, x, y, z);          // BasicLaserMapping.cpp:282-292 via :512-516
    to_be_mapped(T, x, y, z);
    A.stack[i] = make_float4(x, y, z, p.w);
    const float inv = (seg & 1) ? A.inv_odd : A.inv_even;
    int v[3];
    const VbSeg P = seg == lead ? PL : A.segs[seg];
    if (!(vb_voxel(x, inv, v[0]) && vb_voxel(y, inv, v[1]) && vb_voxel(z, inv, v[2])) || P.nbuckets == 0u) {
      vb_fail(A, 0);
      act = false;
    } else {
      const unsigned long long key = vb_key(v[0], v[1], v[2]);
      // bucket = number of splitters (of buckets 1 ..) that are <= key
      uint32_t lo = 0u, hi = P.nbuckets;   // splitter[lo] <= key < splitter[hi]  (splitter[0] = 0, splitter[nbuckets] = infinity)
      if (seg == lead && P.nbuckets <= 64u) {
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_lo[mid] <= key) lo = mid; else hi = mid; }
      } else {
        const unsigned long long* __restrict__ sp = A.lo + P.bucket0;
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (sp[mid] <= key) lo = mid; else hi = mid; }
      }
      g = P.bucket0 + lo;
      pos = i - a0;
    }
  }
  // Slots: the lanes of a wave that go to the same bucket are counted together, the waves of the workgroup meet in an LDS table
  // (256 entries, bucket % 256 with a tag: the buckets a workgroup touches are a few neighbours of one or two segments), and ONE
  // thread per touched bucket bumps the global counter — hundreds of waves hammering ~150 counters serialise in memory otherwise.
  // A bucket that finds its table entry taken by another one goes to the global counter directly.
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
    if (slot < (uint32_t)VB_CAP) A.elems[(size_t)g * VB_CAP + slot] = pos;
    else vb_fail(A, 5);
  }
}

// ----------------------------------------------------------------------------------------------------------------
// k_vb_reduce: grid = buckets, VB_CAP / 8 threads, 8 elements per thread
// ----------------------------------------------------------------------------------------------------------------
constexpr uint32_t VB_SPIN_LIMIT = 1u << 20;
#ifdef LOAMX_PROF_VB
#define VB_TS(k) do { if (threadIdx.x == 0) A.dbg[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define VB_TS(k) do { } while (0)
#endif
constexpr int VB_THREADS = VB_CAP / 8, VB_WAVES = VB_THREADS / 64;

__global__ __launch_bounds__(VB_THREADS) void k_vb_reduce(const VbArgs A) {
  // one buffer, used twice: the sort's words (8 B each; every thread holds its eight words in registers while a pass scatters), then —
  // the words consumed — the bucket's points in sorted order (16 B each) for the voxel means
  __shared__ float4 s_buf[VB_CAP + 1];
  unsigned long long* s_w = (unsigned long long*)s_buf;
  __shared__ uint32_t s_hbits[VB_CAP / 32 + 1];   // bit l: sorted element l starts a voxel
  __shared__ uint32_t s_wcnt[VB_WAVES][256];
  __shared__ uint32_t s_base[256];
  __shared__ uint32_t s_scan[17];
  __shared__ int s_box[6];
  const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // workgroups start in index order, so everything a look-back waits for is already running (or done)
  const uint32_t b = blockIdx.x;
  VB_TS(0);
  if (__hip_atomic_load(&A.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.epoch) {   // given up: an empty, well-formed result
    //  EVERY bucket publishes a count — a bucket that started before the give-up may be looking back at this one —
    // and the offsets are zeroed by the first AND the last bucket: whichever buckets ran to the end before the fail word was raised
    // may have written some, and the Gauss-Newton launches already enqueued behind this kernel read them
    if (tid == 0) __hip_atomic_store(&A.heads[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0 || b + 1u == A.nb)
      for (uint32_t s = (uint32_t)tid; s <= A.nseg; s += VB_THREADS) A.out_off[s] = 0u;
    return;
  }
  const uint32_t seg = A.bseg[b];
  const uint32_t c = min(A.cnt[b], (uint32_t)VB_CAP);
  const VbSeg P = A.segs[seg];
  const uint32_t sbeg = A.seg_off[seg];
  const float inv = (seg & 1) ? A.inv_odd : A.inv_even;
  const uint32_t* __restrict__ e = A.elems + (size_t)b * VB_CAP;
  const uint32_t pbits = P.pos_bits;
  const unsigned long long pmask = (1ull << pbits) - 1ull;
  if (tid < 6) s_box[tid] = tid < 3 ? 2147483647 : (-2147483647 - 1);
  __syncthreads();
  // ---- the elements' voxels (recomputed from the stack points: the same floor(x / leaf) k_vb_stack took) and the bucket's box
  uint32_t pos8[8];
  int vx[8], vy[8], vz[8];
  {
    float4 pt[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
      pos8[j] = i < c ? e[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
      pt[j] = i < c ? A.stack[sbeg + pos8[j]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int mn[3] = {2147483647, 2147483647, 2147483647}, mx[3] = {-2147483647 - 1, -2147483647 - 1, -2147483647 - 1};
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
      vx[j] = vy[j] = vz[j] = 0;
      if (i < c) {
        (void)vb_voxel(pt[j].x, inv, vx[j]); (void)vb_voxel(pt[j].y, inv, vy[j]); (void)vb_voxel(pt[j].z, inv, vz[j]);
        mn[0] = vx[j] < mn[0] ? vx[j] : mn[0]; mx[0] = vx[j] > mx[0] ? vx[j] : mx[0];
        mn[1] = vy[j] < mn[1] ? vy[j] : mn[1]; mx[1] = vy[j] > mx[1] ? vy[j] : mx[1];
        mn[2] = vz[j] < mn[2] ? vz[j] : mn[2]; mx[2] = vz[j] > mx[2] ? vz[j] : mx[2];
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
    if (lane == 0 && (uint32_t)(wid * 512) < c) {
#pragma unroll
      for (int a = 0; a < 3; a++) { atomicMin(&s_box[a], mn[a]); atomicMax(&s_box[3 + a], mx[a]); }
    }
  }
  __syncthreads();
  VB_TS(1);
  unsigned long long dimx = 1ull, dimy = 1ull, dimz = 1ull;
  int bx0 = 0, by0 = 0, bz0 = 0;
  if (c) {
    bx0 = s_box[0]; by0 = s_box[1]; bz0 = s_box[2];
    dimx = (unsigned long long)(s_box[3] - bx0 + 1); dimy = (unsigned long long)(s_box[4] - by0 + 1); dimz = (unsigned long long)(s_box[5] - bz0 + 1);
    if (tid < 6) {   // the segment's box, for PCL's pass-through test by the last bucket (released by the fence in front of the head count)
      if (tid < 3) atomicMin(&A.box[6 * seg + tid], s_box[tid]); else atomicMax(&A.box[6 * seg + tid], s_box[tid]);
    }
  }
  const uint32_t key_bits = vb_bits(dimx * dimy * dimz - 1ull);   // each extent < 2^21
  const uint32_t total_bits = (key_bits ? key_bits : 1u) + pbits;
  if (total_bits > 63u) {   // (block-uniform) a box this sparse cannot be sorted on one word: give up
    if (tid == 0) vb_fail(A, 3);
  }
  const uint32_t npass = total_bits > 63u ? 0u : (total_bits + 7u) / 8u;
  // sort word: voxel index inside the bucket's box << pos_bits | input position inside the segment — all distinct
  unsigned long long w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
    const unsigned long long lin = (unsigned long long)(vx[j] - bx0) + dimx * ((unsigned long long)(vy[j] - by0) + dimy * (unsigned long long)(vz[j] - bz0));
    w[j] = i < c ? ((lin << pbits) | (unsigned long long)pos8[j]) : ~0ull;
  }
  const bool wave_has = (uint32_t)(wid * 512) < c;   // wave-uniform: a wave whose slots are all beyond the end only keeps the barriers
  if (npass == 0u) {   // (given up: keep the buffer defined)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = (uint32_t)(wid * 512 + j * 64 + lane);
      if (i < c) s_w[i] = w[j];
    }
    __syncthreads();
  }
  for (uint32_t p = 0; p < npass; p++) {
    const uint32_t shift = 8u * p;
    VB_TS(2 + (p < 7u ? p : 7u));
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
  // s_w holds the sorted words.  Every thread takes eight CONSECUTIVE sorted elements and fetches their points itself (eight
  // independent gathers in flight); a voxel's mean is the sequential sum of its run in sorted = input order.  The points then go to
  // LDS in sorted order (over the words, which are in registers by now) together with one head bit per element, so that a run which
  // goes on beyond its head's own eight elements is continued from LDS.  (Round 3 continued through the words and global memory, one
  // dependent gather per point: a voxel of ~100 points — dense ground next to the sensor — kept its thread, and with it the kernel,
  // 40 us longer than every other bucket: in-kernel time stamps, profiles/r04_vb_reduce.md.)
  VB_TS(10);
  if (tid == 0) s_w[c] = ~0ull;   // sentinel behind the last element
  __syncthreads();
  const uint32_t l0 = (uint32_t)tid * 8u;
  unsigned long long vox[8];
  float4 pt[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t l = l0 + (uint32_t)j;
    const unsigned long long x = l < c ? s_w[l] : ~0ull;
    vox[j] = x >> pbits;   // (the sentinel's is larger than any index of the box)
    pt[j] = l < c ? A.stack[sbeg + (uint32_t)(x & pmask)] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  bool head[8];
  uint32_t nh = 0u, hb = 0u;
  unsigned long long prev = (l0 == 0u || l0 > c) ? ~0ull : (s_w[l0 - 1u] >> pbits);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t l = l0 + (uint32_t)j;
    head[j] = l < c && (l == 0u || vox[j] != prev);   // (a voxel never straddles two buckets: buckets are ranges of the voxel order)
    nh += head[j] ? 1u : 0u;
    hb |= (head[j] || l >= c) ? (1u << j) : 0u;       // (elements behind the last one end every run)
    prev = vox[j];
  }
  __syncthreads();   // every thread has read its words (and its predecessor's): the buffer changes hands
#pragma unroll
  for (int j = 0; j < 8; j++) s_buf[l0 + (uint32_t)j] = pt[j];
  ((unsigned char*)s_hbits)[tid] = (unsigned char)hb;   // (little-endian bytes: bit l of the table = element l)
  if (tid == 0) ((unsigned char*)s_hbits)[VB_THREADS] = 0xffu;
  __syncthreads();
  VB_TS(11);
  uint32_t tot;
  const uint32_t ex = block_excl_scan(nh, s_scan, tot);
  if (tid == 0) {
    xchg_stores_done();   // the box atomics above (this wave's lanes 0-5) have been performed before the count is published; no cache-wide fence (dev_math.hpp)
    __hip_atomic_store(&A.heads[b], tot + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
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
  VB_TS(12);
  uint32_t base;
  (void)block_excl_scan(part, s_scan, base);   // voxels emitted by all earlier buckets
  if (tid == 0) {
    if (P.bucket0 == b) A.out_off[seg] = base;
    if (b + 1u == A.nb) A.out_off[A.nseg] = base + tot;
  }
  if (b + 1u == A.nb) {   // the last bucket has seen every other bucket's count, hence every segment's final box: PCL's pass-through test
    for (uint32_t s = (uint32_t)tid; s < A.nseg; s += VB_THREADS) {
      int bb[6];
#pragma unroll
      for (int k = 0; k < 6; k++) bb[k] = __hip_atomic_load(&A.box[6 * s + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (bb[3] >= bb[0]) {
        const long long dx = (long long)bb[3] - bb[0] + 1, dy = (long long)bb[4] - bb[1] + 1, dz = (long long)bb[5] - bb[2] + 1;
        if (dx * dy > 2147483647LL || dx * dy * dz > 2147483647LL) vb_fail(A, 2);   // (each extent < 2^21)
      }
    }
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
      if (open) {   // the run reaches the end of this thread's elements: it goes on in LDS until the next head bit
        uint32_t l = l0 + 8u;
        while (open) {
          const uint32_t hbyte = ((const unsigned char*)s_hbits)[l >> 3];   // (l is a multiple of 8: one byte = the next eight elements)
          float4 ts[8];
#pragma unroll
          for (int q = 0; q < 8; q++) ts[q] = s_buf[min(l + (uint32_t)q, (uint32_t)VB_CAP)];
#pragma unroll
          for (int q = 0; q < 8; q++) {
            open = open && !((hbyte >> q) & 1u);
            if (open) { sx += ts[q].x; sy += ts[q].y; sz += ts[q].z; si += ts[q].w; cntp++; }
          }
          l += 8u;
        }
      }
#ifdef LOAMX_PROF_VB
      atomicMax((unsigned long long*)&A.dbg[(size_t)blockIdx.x * 16 + 14], (unsigned long long)cntp);
#endif
      const float cf = (float)cntp;
      A.out[pos] = make_float4(sx / cf, sy / cf, sz / cf, si / cf);
      pos++;
    }
  }
  VB_TS(13);
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
  lo_.reserve(nb);
  bseg_.reserve(nb);
  box_.reserve((size_t)6 * nseg);
  cnt_.reserve(nb);
  heads_.reserve(nb);
  elems_.reserve((size_t)nb * VB_CAP);
  if (!ctl_ready_) {
    ctl_.reserve(4);
    h_fail_.reserve(16);
    for (int k = 0; k < 16; k++) h_fail_.p[k] = 0u;
    LX_HIP(hipMemsetAsync(ctl_.p, 0, sizeof(uint32_t) * 4, st_));
    ctl_ready_ = true;
  }
  if (++epoch_ == 0u) epoch_ = 1u;
  VbArgs a;
  a.in = in; a.src = d_src; a.seg_off = d_seg_off; a.poses = d_poses;
  a.segs = segs_.p; a.lo = lo_.p; a.bseg = bseg_.p; a.box = box_.p; a.cnt = cnt_.p; a.heads = heads_.p;
  a.ctl = ctl_.p; a.h_fail = h_fail_.p; a.elems = elems_.p; a.stack = stack; a.out = out; a.out_off = d_out_off;
  a.n = n; a.nseg = nseg; a.nb = nb; a.epoch = epoch_;
  a.inv_even = inv_even; a.inv_odd = inv_odd;
  a.dbg = nullptr;
#ifdef LOAMX_PROF_VB
  static DevBuf<unsigned long long> dbg;
  dbg.reserve((size_t)nb * 16 + 16);
  LX_HIP(hipMemsetAsync(dbg.p, 0, sizeof(unsigned long long) * ((size_t)nb * 16), st_));
  a.dbg = dbg.p;
#endif
  hipLaunchKernelGGL(k_vb_plan, dim3(nseg), dim3(VB_SAMPLE), 0, st_, a);
  hipLaunchKernelGGL(k_vb_stack, dim3((n + 255) / 256), dim3(256), 0, st_, a);
  hipLaunchKernelGGL(k_vb_reduce, dim3(nb), dim3(VB_THREADS), 0, st_, a);
  LX_HIP(hipGetLastError());
#ifdef LOAMX_PROF_VB
  {
    std::vector<unsigned long long> h((size_t)nb * 16);
    LX_HIP(hipStreamSynchronize(st_));
    LX_HIP(hipMemcpy(h.data(), dbg.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (uint32_t b = 0; b < nb; b++) { if (h[16 * b]) t0 = std::min(t0, h[16 * b]); t1 = std::max(t1, h[16 * b + 13]); }
    fprintf(stderr, "[k_vb_reduce %u buckets, %.1f us first start -> last end] bucket: start | box | passes ... | sorted | heads | look-back | end (us since first start)\n", nb, (t1 - t0) * 0.01);
    uint32_t slow = 0;
    for (uint32_t b = 0; b < nb; b++) if (h[16 * b + 13] > h[16 * slow + 13]) slow = b;
    for (uint32_t b : {0u, nb / 4, nb / 2, nb - 1, slow}) {
      fprintf(stderr, "  b%-4u", b);
      for (int k = 0; k < 14; k++) if (h[16 * b + k]) fprintf(stderr, " %d:%.1f", k, (h[16 * b + k] - t0) * 0.01);
      fprintf(stderr, " longest run %llu", h[16 * b + 14]);
      fprintf(stderr, "\n");
    }
  }
#endif
}

// after failed(): which conditions the last run met (bit r = reason r of vb_fail)
uint32_t VoxBucket::why() const {
  uint32_t m = 0;
  for (int r = 0; r < 8; r++)
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
