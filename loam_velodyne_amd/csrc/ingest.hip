// Raw-sweep ingestion for gfx950 — see ingest.hpp.  The reference walks the points once, sequentially; the only
// order-dependent pieces are (a) the halfPassed flag — it flips at the FIRST kept point whose unwrapped azimuth is more
// than pi past the start and stays set, so it is "index > j*" with j* a min-reduction — and (b) the per-ring push_back,
// i.e. a stable split by ring id: per-workgroup ring histograms, a column scan over the workgroups, and a stable rank
// inside the workgroup (wave ballots per distinct ring + wave-order prefix).
#include "ingest.hpp"

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

// scratch: [0] j* (first kept point that sets halfPassed), [1] last kept point, [2] startOri, [3] endOri (float bits)
__global__ void k_raw_init(const float4* __restrict__ raw, uint32_t n, uint32_t* scratch, uint32_t* imu_first, uint32_t H) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) {
    scratch[0] = 0xffffffffu;
    scratch[1] = 0u;
    float startOri, endOri;
    sweep_oris(raw, n, startOri, endOri);
    scratch[2] = __float_as_uint(startOri);
    scratch[3] = __float_as_uint(endOri);
  }
  if (e < H) imu_first[e] = 0xffffffffu;
}

// relTime of kept point i (:209-228); x, z in the LOAM frame
__device__ inline float rel_time_of(uint32_t i, float x, float z, float startOri, float endOri, uint32_t jstar, float scan_period) {
  bool passes;
  float ori = ori_first_half(x, z, startOri, passes);
  if (i > jstar) ori = ori_second_half(x, z, endOri);   // halfPassed was set by an earlier kept point
  return scan_period * (ori - startOri) / (endOri - startOri);
}

// interpolateIMUStateFor's index search for one point on its own (:135-139): smallest j with dt[j] + relTime <= 0, else H-1
__device__ inline uint32_t imu_need(const ImuTable& I, float relTime) {
  uint32_t lo = 0, hi = I.H - 1;   // dt is decreasing in j
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (I.dt[mid] + (double)relTime > 0) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// _imuIdx only ever moves forward while the points are walked in firing order: idx(i) = max(idx0, max_{kept j <= i} need(j)).
// k_raw_imu_need records, per history index v, the first kept point that needs at least ... exactly v; the suffix minimum
// over v turns that into "first point that needs >= v", which is non-decreasing in v and binary-searchable per point.
__global__ __launch_bounds__(256) void k_raw_imu_need(const float4* __restrict__ raw, uint32_t n, float scan_period, const int* __restrict__ ring_of,
                                                      const uint32_t* __restrict__ jstar, ImuTable I, uint32_t* __restrict__ first) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n && ring_of[i] >= 0;
  uint32_t v = 0;
  if (active) {
    const float startOri = __uint_as_float(jstar[2]), endOri = __uint_as_float(jstar[3]);
    const float4 r = raw[i];
    const float relTime = rel_time_of(i, r.y, r.x, startOri, endOri, jstar[0], scan_period);
    v = imu_need(I, relTime);
  }
  // neighbouring points need the same history index: only the first lane of a run of equal v (it has the run's smallest
  // point index) issues the atomic
  const int lane = threadIdx.x & 63;
  const uint32_t pv = __shfl_up(v, 1, 64);
  const unsigned long long act = __ballot(active);
  const bool head = active && (lane == 0 || pv != v || !((act >> (lane > 0 ? lane - 1 : 0)) & 1ull));
  if (head && v > I.idx0) atomicMin(&first[v], i);
}
// first[v] = min over u >= v (one workgroup; H <= 4096: Hillis-Steele steps over LDS)
__global__ __launch_bounds__(1024) void k_raw_imu_suffix(uint32_t* __restrict__ first, uint32_t H) {
  __shared__ uint32_t a[4096];
  for (uint32_t v = threadIdx.x; v < H; v += blockDim.x) a[v] = first[v];
  __syncthreads();
  for (uint32_t d = 1; d < H; d <<= 1) {
    uint32_t t[4];
    int k = 0;
    for (uint32_t v = threadIdx.x; v < H; v += blockDim.x, k++) t[k] = v + d < H ? min(a[v], a[v + d]) : a[v];
    __syncthreads();
    k = 0;
    for (uint32_t v = threadIdx.x; v < H; v += blockDim.x, k++) a[v] = t[k];
    __syncthreads();
  }
  for (uint32_t v = threadIdx.x; v < H; v += blockDim.x) first[v] = a[v];
}

// Angle(float): cached sin / cos of a float angle (Angle.h:16-30); double-then-round, within an ulp of the host's float libm
struct DAngle { float c, s; };
__device__ inline DAngle dangle(float r) { return {(float)cos((double)r), (float)sin((double)r)}; }
__device__ inline void d_rot_x(float& y, float& z, float c, float s) { const float y0 = y; y = c * y0 - s * z; z = s * y0 + c * z; }
__device__ inline void d_rot_y(float& x, float& z, float c, float s) { const float x0 = x; x = c * x0 + s * z; z = c * z - s * x0; }
__device__ inline void d_rot_z(float& x, float& y, float c, float s) { const float x0 = x; x = c * x0 - s * y; y = s * x0 + c * y; }

// setIMUTransformFor + transformToStartIMU for kept point i (:112-131)
__device__ inline void imu_project(const ImuTable& I, const uint32_t* __restrict__ first, uint32_t i, float relTime, float& x, float& y, float& z,
                                   ImuLast* last_out) {
  // _imuIdx after this point
  uint32_t idx = I.idx0;
  {
    uint32_t lo = I.idx0, hi = I.H - 1;   // largest v in (idx0, H-1] with first[v] <= i
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (first[mid] <= i) lo = mid; else hi = mid - 1;
    }
    idx = lo;
  }
  const double timeDiff = I.dt[idx] + (double)relTime;
  float cur[9];
  if (idx == 0 || timeDiff > 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) cur[k] = I.state[9 * idx + k];
  } else {
    const float ratio = (float)(-timeDiff / I.dstamp[idx]), inv = 1 - ratio;   // IMUState::interpolate(hist[idx], hist[idx-1], ratio)
    const float* a = I.state + 9 * idx;
    const float* b = I.state + 9 * (idx - 1);
    cur[0] = a[0] * inv + b[0] * ratio;
    cur[1] = a[1] * inv + b[1] * ratio;
    if ((double)(a[2] - b[2]) > PI_D) cur[2] = (float)((double)(a[2] * inv) + ((double)b[2] + 2 * PI_D) * (double)ratio);
    else if ((double)(a[2] - b[2]) < -PI_D) cur[2] = (float)((double)(a[2] * inv) + ((double)b[2] - 2 * PI_D) * (double)ratio);
    else cur[2] = a[2] * inv + b[2] * ratio;
#pragma unroll
    for (int k = 3; k < 9; k++) cur[k] = a[k] * inv + b[k] * ratio;
  }
  const float relSweepTime = (float)(I.rel_sweep_base + (double)relTime);
  const float sx = cur[3] - I.start_pos[0] - I.start_vel[0] * relSweepTime;
  const float sy = cur[4] - I.start_pos[1] - I.start_vel[1] * relSweepTime;
  const float sz = cur[5] - I.start_pos[2] - I.start_vel[2] * relSweepTime;
  const DAngle ro = dangle(cur[0]), pi = dangle(cur[1]), ya = dangle(cur[2]);
  d_rot_z(x, y, ro.c, ro.s); d_rot_x(y, z, pi.c, pi.s); d_rot_y(x, z, ya.c, ya.s);   // rotateZXY(roll, pitch, yaw)
  x += sx; y += sy; z += sz;
  d_rot_y(x, z, I.start_c[2], -I.start_s[2]); d_rot_x(y, z, I.start_c[1], -I.start_s[1]); d_rot_z(x, y, I.start_c[0], -I.start_s[0]);   // rotateYXZ(-yaw, -pitch, -roll)
  if (last_out) {
    last_out->roll = cur[0]; last_out->pitch = cur[1]; last_out->yaw = cur[2];
    for (int k = 0; k < 3; k++) { last_out->pos[k] = cur[3 + k]; last_out->vel[k] = cur[6 + k]; }
    last_out->shift[0] = sx; last_out->shift[1] = sy; last_out->shift[2] = sz;
    last_out->idx = idx; last_out->valid = 1u;
  }
}

// per workgroup: ring histogram, first point that would set halfPassed, last kept point (k_raw_colscan reduces the two
// index arrays — thousands of atomics on ONE global word per launch serialise for tens of microseconds)
__global__ __launch_bounds__(256) void k_raw_classify(const float4* __restrict__ raw, uint32_t n, MapperParams M, int* __restrict__ ring_of,
                                                      uint32_t* __restrict__ blk_cnt, const uint32_t* __restrict__ jstar,
                                                      uint32_t* __restrict__ blk_first_pass, uint32_t* __restrict__ blk_last_kept) {
  __shared__ uint32_t hist[RawBinner::MAX_RINGS];
  __shared__ uint32_t s_first, s_last;
  if (threadIdx.x == 0) { s_first = 0xffffffffu; s_last = 0u; }
  for (uint32_t r = threadIdx.x; r < M.n_rings; r += blockDim.x) hist[r] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool kept = false, passes = false;
  if (i < n) {
    const float startOri = __uint_as_float(jstar[2]);
    float x, y, z;
    const int id = classify(raw[i], M, x, y, z);
    ring_of[i] = id;
    if (id >= 0) {
      kept = true;
      atomicAdd(&hist[id], 1u);
      (void)ori_first_half(x, z, startOri, passes);
    }
  }
  // one atomic per wave: the lowest passing lane holds the wave's smallest index, the highest kept lane its largest
  const unsigned long long mp = __ballot(passes), mk = __ballot(kept);
  const int lane = threadIdx.x & 63;
  if (mp && lane == __builtin_ctzll(mp)) atomicMin(&s_first, i);
  if (mk && lane == 63 - __builtin_clzll(mk)) atomicMax(&s_last, i + 1u);   // (+1: 0 = no kept point in this workgroup)
  __syncthreads();
  if (threadIdx.x == 0) { blk_first_pass[blockIdx.x] = s_first; blk_last_kept[blockIdx.x] = s_last; }
  for (uint32_t r = threadIdx.x; r < M.n_rings; r += blockDim.x) blk_cnt[(size_t)blockIdx.x * M.n_rings + r] = hist[r];
}

// one workgroup per ring: exclusive scan of the ring's column of workgroup counts
// (workgroup 0 also reduces the per-workgroup first-pass / last-kept indices into scratch[0] / scratch[1])
__global__ __launch_bounds__(256) void k_raw_colscan(const uint32_t* __restrict__ blk_cnt, uint32_t nblk, uint32_t nrings,
                                                     uint32_t* __restrict__ blk_pre, uint32_t* __restrict__ ring_cnt,
                                                     const uint32_t* __restrict__ blk_first_pass, const uint32_t* __restrict__ blk_last_kept,
                                                     uint32_t* __restrict__ scratch) {
  __shared__ uint32_t sc[256];
  const uint32_t r = blockIdx.x;
  if (r == 0) {
    __shared__ uint32_t s_first, s_last;
    if (threadIdx.x == 0) { s_first = 0xffffffffu; s_last = 0u; }
    __syncthreads();
    uint32_t f = 0xffffffffu, l = 0u;
    for (uint32_t b = threadIdx.x; b < nblk; b += 256) { f = min(f, blk_first_pass[b]); l = max(l, blk_last_kept[b]); }
    atomicMin(&s_first, f);
    atomicMax(&s_last, l);
    __syncthreads();
    if (threadIdx.x == 0) { scratch[0] = s_first; scratch[1] = s_last ? s_last - 1u : 0u; }
  }
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
                                                     float4* __restrict__ out, ImuTable I, const uint32_t* __restrict__ imu_first,
                                                     ImuLast* __restrict__ d_last) {
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
  const float startOri = __uint_as_float(jstar[2]), endOri = __uint_as_float(jstar[3]);
  const float4 r = raw[i];
  float x = r.y, y = r.z, z = r.x;
  const float relTime = rel_time_of(i, x, z, startOri, endOri, jstar[0], scan_period);   // :228
  if (I.H) imu_project(I, imu_first, i, relTime, x, y, z, i == jstar[1] ? d_last : nullptr);   // :231
  out[pos] = make_float4(x, y, z, (float)id + relTime);                                  // :229
}

}  // namespace

// raw records (x, y, z float32 at byte offsets 0 / 4 / 8, `stride` bytes apart) -> float4 (x, y, z, 0): the payload crosses PCIe as
// it came and is re-strided on the device
__global__ __launch_bounds__(256) void k_raw_unpack(const char* __restrict__ bytes, uint32_t stride, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = (const float*)(bytes + (size_t)i * stride);
  out[i] = make_float4(r[0], r[1], r[2], 0.f);
}
void raw_unpack(const void* d_bytes, uint32_t stride, uint32_t n, float4* d_out, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_raw_unpack, dim3((n + 255) / 256), dim3(256), 0, st, (const char*)d_bytes, stride, n, d_out);
}

void RawBinner::run(const float4* d_raw, uint32_t n, const MapperParams& m, float scan_period, float4* d_out, uint32_t* d_ring_cnt,
                    const ImuTable* imu, ImuLast* d_last) {
  LX_REQUIRE(m.n_rings >= 1 && m.n_rings <= MAX_RINGS, "n_scan_rings must be in [1, 256]");
  if (n == 0) {
    LX_HIP(hipMemsetAsync(d_ring_cnt, 0, sizeof(uint32_t) * m.n_rings, st_));
    return;
  }
  const uint32_t nb = (n + 255) / 256;
  ring_of_.reserve(n + 1);
  blk_cnt_.reserve((size_t)nb * m.n_rings + 1);
  blk_pre_.reserve((size_t)nb * m.n_rings + 1);
  scratch_.reserve(8);
  ImuTable I;
  if (imu) I = *imu;
  LX_REQUIRE(I.H <= 4096, "IMU history longer than 4096 states");
  imu_first_.reserve(I.H + 1);
  hipLaunchKernelGGL(k_raw_init, dim3((I.H + 256) / 256), dim3(256), 0, st_, d_raw, n, scratch_.p, imu_first_.p, I.H);
  blk_idx_.reserve(2 * (size_t)nb + 2);
  hipLaunchKernelGGL(k_raw_classify, dim3(nb), dim3(256), 0, st_, d_raw, n, m, ring_of_.p, blk_cnt_.p, scratch_.p, blk_idx_.p, blk_idx_.p + nb);
  hipLaunchKernelGGL(k_raw_colscan, dim3(m.n_rings), dim3(256), 0, st_, blk_cnt_.p, nb, m.n_rings, blk_pre_.p, d_ring_cnt, blk_idx_.p,
                     blk_idx_.p + nb, scratch_.p);
  if (I.H) {
    hipLaunchKernelGGL(k_raw_imu_need, dim3(nb), dim3(256), 0, st_, d_raw, n, scan_period, ring_of_.p, scratch_.p, I, imu_first_.p);
    hipLaunchKernelGGL(k_raw_imu_suffix, dim3(1), dim3(1024), 0, st_, imu_first_.p, I.H);
  }
  hipLaunchKernelGGL(k_raw_scatter, dim3(nb), dim3(256), 0, st_, d_raw, n, m, scan_period, ring_of_.p, blk_pre_.p, d_ring_cnt, scratch_.p,
                     d_out, I, imu_first_.p, d_last);
  LX_HIP(hipGetLastError());
}

}  // namespace loamx
