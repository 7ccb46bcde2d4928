// Raw-sweep ingestion for gfx950 — see ingest.cuh.  The reference walks the points once, sequentially; the only
// order-dependent pieces are (a) the halfPassed flag — it flips at the FIRST kept point whose unwrapped azimuth is more
// than pi past the start and stays set, so it is "index > j*" with j* a min-reduction — and (b) the per-ring push_back,
// i.e. a stable split by ring id: per-workgroup ring histograms, a column scan over the workgroups, and a stable rank
// inside the workgroup (wave ballots per distinct ring + wave-order prefix).
#include "ingest.cuh"

namespace loamx {

namespace {

constexpr double PI_D = 3.14159265358979323846;

// atan / atan2 of floats: evaluated in double and rounded (within an ulp of the C library's float versions the
// reference calls, and reproducible)
__device__ inline float atan2_f(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ inline float atan_f(float v) { return (float)atan((double)v); }

// scan start / end orientation, MultiScanRegistration.cpp:165-173
__device__ inline void sweep_oris(const float4* __restrict__ raw, uint32_t n, float& startOri, float& endOri) {
  const float4 a = raw[0], b = raw[n - 1];
  startOri = -atan2_f(a.y, a.x);
  endOri = -atan2_f(b.y, b.x) + 2 * (float)PI_D;
  if ((double)(endOri - startOri) > 3 * PI_D) endOri = (float)((double)endOri - 2 * PI_D);
  else if ((double)(endOri - startOri) < PI_D) endOri = (float)((double)endOri + 2 * PI_D);
}

// :184-205: remapped point and ring id (-1: rejected)
__device__ inline int classify(const float4 r, const MapperParams& M, float& x, float& y, float& z) {
  x = r.y; y = r.z; z = r.x;
  if (!isfinite(x) || !isfinite(y) || !isfinite(z)) return -1;
  if ((double)(x * x + y * y + z * z) < 0.0001) return -1;
  const float angle = atan_f(y / sqrtf(x * x + z * z));
  const int id = (int)((((double)(angle * 180) / PI_D) - (double)M.lower) * (double)M.factor + 0.5);   // getRingForAngle :64-66
  return (id >= (int)M.n_rings || id < 0) ? -1 : id;
}

// the !halfPassed branch of :209-219; returns the unwrapped azimuth, `passes` = this point sets halfPassed
__device__ inline float ori_first_half(float x, float z, float startOri, bool& passes) {
  float ori = -atan2_f(x, z);
  if ((double)ori < (double)startOri - PI_D / 2) ori = (float)((double)ori + 2 * PI_D);
  else if ((double)ori > (double)startOri + PI_D * 3 / 2) ori = (float)((double)ori - 2 * PI_D);
  passes = (double)(ori - startOri) > PI_D;
  return ori;
}
// the halfPassed branch, :220-226
__device__ inline float ori_second_half(float x, float z, float endOri) {
  float ori = -atan2_f(x, z);
  ori = (float)((double)ori + 2 * PI_D);
  if ((double)ori < (double)endOri - PI_D * 3 / 2) ori = (float)((double)ori + 2 * PI_D);
  else if ((double)ori > (double)endOri + PI_D / 2) ori = (float)((double)ori - 2 * PI_D);
  return ori;
}

__global__ void k_raw_init(uint32_t* scratch) { scratch[0] = 0xffffffffu; }

__global__ __launch_bounds__(256) void k_raw_classify(const float4* __restrict__ raw, uint32_t n, MapperParams M, int* __restrict__ ring_of,
                                                      uint32_t* __restrict__ blk_cnt, uint32_t* __restrict__ jstar) {
  __shared__ uint32_t hist[RawBinner::MAX_RINGS];
  for (uint32_t r = threadIdx.x; r < M.n_rings; r += blockDim.x) hist[r] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float startOri, endOri;
    sweep_oris(raw, n, startOri, endOri);
    float x, y, z;
    const int id = classify(raw[i], M, x, y, z);
    ring_of[i] = id;
    if (id >= 0) {
      atomicAdd(&hist[id], 1u);
      bool passes;
      (void)ori_first_half(x, z, startOri, passes);
      if (passes) atomicMin(jstar, i);
    }
  }
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < M.n_rings; r += blockDim.x) blk_cnt[(size_t)blockIdx.x * M.n_rings + r] = hist[r];
}

// one workgroup per ring: exclusive scan of the ring's column of workgroup counts
__global__ __launch_bounds__(256) void k_raw_colscan(const uint32_t* __restrict__ blk_cnt, uint32_t nblk, uint32_t nrings,
                                                     uint32_t* __restrict__ blk_pre, uint32_t* __restrict__ ring_cnt) {
  __shared__ uint32_t sc[256];
  const uint32_t r = blockIdx.x;
  uint32_t base = 0;
  for (uint32_t b0 = 0; b0 < nblk; b0 += 256) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = b < nblk ? blk_cnt[(size_t)b * nrings + r] : 0u;
    sc[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {   // Hillis-Steele inclusive scan
      const uint32_t t = threadIdx.x >= d ? sc[threadIdx.x - d] : 0u;
      __syncthreads();
      sc[threadIdx.x] += t;
      __syncthreads();
    }
    if (b < nblk) blk_pre[(size_t)b * nrings + r] = base + sc[threadIdx.x] - v;
    base += sc[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) ring_cnt[r] = base;
}

__global__ __launch_bounds__(256) void k_raw_scatter(const float4* __restrict__ raw, uint32_t n, MapperParams M, float scan_period,
                                                     const int* __restrict__ ring_of, const uint32_t* __restrict__ blk_pre,
                                                     const uint32_t* __restrict__ ring_cnt, const uint32_t* __restrict__ jstar,
                                                     float4* __restrict__ out) {
  __shared__ uint32_t ring_off[RawBinner::MAX_RINGS];
  __shared__ uint32_t wcnt[4][RawBinner::MAX_RINGS];
  // ring offsets (exclusive scan of the ring totals; <= 256 rings)
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t r = 0; r < M.n_rings; r++) { ring_off[r] = acc; acc += ring_cnt[r]; }
  }
  for (uint32_t e = threadIdx.x; e < 4 * RawBinner::MAX_RINGS; e += blockDim.x) (&wcnt[0][0])[e] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int id = i < n ? ring_of[i] : -1;
  // stable rank among the same-ring points of this wave
  uint32_t rank = 0;
  unsigned long long todo = __ballot(id >= 0);
  while (todo) {
    const int src = __builtin_ctzll(todo);
    const int r0 = __shfl(id, src, 64);
    const unsigned long long m = __ballot(id == r0);
    if (id == r0) {
      rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (lane == src) wcnt[wid][r0] = (uint32_t)__popcll(m);
    }
    todo &= ~m;
  }
  __syncthreads();
  if (id < 0) return;
  for (int w = 0; w < wid; w++) rank += wcnt[w][id];
  const uint32_t pos = ring_off[id] + blk_pre[(size_t)blockIdx.x * M.n_rings + id] + rank;
  float startOri, endOri;
  sweep_oris(raw, n, startOri, endOri);
  const float4 r = raw[i];
  const float x = r.y, y = r.z, z = r.x;
  bool passes;
  float ori = ori_first_half(x, z, startOri, passes);
  if (i > *jstar) ori = ori_second_half(x, z, endOri);   // halfPassed was set by an earlier kept point
  const float relTime = scan_period * (ori - startOri) / (endOri - startOri);   // :228
  out[pos] = make_float4(x, y, z, (float)id + relTime);                          // :229
}

}  // namespace

void RawBinner::run(const float4* d_raw, uint32_t n, const MapperParams& m, float scan_period, float4* d_out, uint32_t* d_ring_cnt) {
  LX_REQUIRE(m.n_rings >= 1 && m.n_rings <= MAX_RINGS, "n_scan_rings must be in [1, 256]");
  if (n == 0) {
    LX_HIP(hipMemsetAsync(d_ring_cnt, 0, sizeof(uint32_t) * m.n_rings, st_));
    return;
  }
  const uint32_t nb = (n + 255) / 256;
  ring_of_.reserve(n + 1);
  blk_cnt_.reserve((size_t)nb * m.n_rings + 1);
  blk_pre_.reserve((size_t)nb * m.n_rings + 1);
  scratch_.reserve(4);
  hipLaunchKernelGGL(k_raw_init, dim3(1), dim3(1), 0, st_, scratch_.p);
  hipLaunchKernelGGL(k_raw_classify, dim3(nb), dim3(256), 0, st_, d_raw, n, m, ring_of_.p, blk_cnt_.p, scratch_.p);
  hipLaunchKernelGGL(k_raw_colscan, dim3(m.n_rings), dim3(256), 0, st_, blk_cnt_.p, nb, m.n_rings, blk_pre_.p, d_ring_cnt);
  hipLaunchKernelGGL(k_raw_scatter, dim3(nb), dim3(256), 0, st_, d_raw, n, m, scan_period, ring_of_.p, blk_pre_.p, d_ring_cnt, scratch_.p,
                     d_out);
  LX_HIP(hipGetLastError());
}

}  // namespace loamx
