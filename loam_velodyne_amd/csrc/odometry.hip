// Sweep-to-sweep odometry for gfx950 — reference src/lib/BasicLaserOdometry.cpp.
//
// The problem is small (<= 768 sharp + 1536 flat features against the previous sweep's <= ~60 k feature points) and
// strictly iterative (<= 25 dependent Gauss-Newton steps), i.e. latency-bound.  One PERSISTENT workgroup per sweep runs
// the whole loop in a single launch, so there is no kernel boundary (~1.5 us each) between the 25 iterations:
//   every 5th iteration  (:250-302, :368-435)
//     phase A  thread per feature: transformToStart + exact 1-NN (5 m gate) in the previous cloud through a uniform grid
//              searched in expanding shells (replaces the kd-tree of :203-204 / :662-663)
//     phase B  wave per feature: the +-2.5-ring window scans over the ring-ordered previous cloud, lanes striding the
//              window, ballot for the loop's break, shuffle arg-min with scan-order tie-break
//   every iteration      (:304-361, :437-481, :497-559)
//     phase C  thread per feature: point-to-line / point-to-plane coefficients, Jacobian row with the de-skew chain
//              rule, J^T J / J^T r reduced with wave shuffles (double accumulators), thread 0: 6x6 pivoted QR solve,
//              degeneracy projector, update, convergence test.
#include "odometry.cuh"

namespace loamx {

constexpr int OD_THREADS = 512;
constexpr int OD_WAVES = OD_THREADS / 64;

__device__ inline void sincos_f(float a, float& s, float& c) {
  s = (float)sin((double)a);
  c = (float)cos((double)a);
}

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

__device__ inline float sqd(const float4& a, float x, float y, float z) {
  const float dx = a.x - x, dy = a.y - y, dz = a.z - z;
  return dx * dx + dy * dy + dz * dz;
}

// exact nearest neighbour with squared distance < 25 (:253-256); returns the ORIGINAL index or -1
__device__ inline int nn1(const GridDesc& g, const float4* __restrict__ sorted, const uint32_t* __restrict__ cell_start, float qx,
                          float qy, float qz) {
  float best = FLT_MAX;
  uint32_t best_id = 0xffffffffu;
  const float h = 1.0f / g.inv_h;
  const int cx = (int)floorf((qx - g.ox) * g.inv_h), cy = (int)floorf((qy - g.oy) * g.inv_h), cz = (int)floorf((qz - g.oz) * g.inv_h);
  for (int L = 0;; L++) {
    for (int dz = -L; dz <= L; dz++) {
      const int z = cz + dz;
      if (z < 0 || z >= g.nz) continue;
      for (int dy = -L; dy <= L; dy++) {
        const int y = cy + dy;
        if (y < 0 || y >= g.ny) continue;
        const bool face = (dz == -L || dz == L || dy == -L || dy == L);
        const uint32_t row = ((uint32_t)z * g.ny + y) * g.nx;
        // on a face row the whole x-run [cx-L, cx+L] is new; otherwise only its two end cells
        for (int part = 0; part < 2; part++) {
          int xa, xb;
          if (face) {
            if (part) break;
            xa = cx - L; xb = cx + L;
          } else {
            xa = xb = part ? cx + L : cx - L;
            if (L == 0 && part) break;
          }
          if (xa < 0) xa = 0;
          if (xb > g.nx - 1) xb = g.nx - 1;
          if (xa > xb) continue;
          const uint32_t beg = cell_start[row + xa], end = cell_start[row + xb + 1];
          for (uint32_t k = beg; k < end; k++) {
            const float4 p = sorted[k];
            const float dx = qx - p.x, dy2 = qy - p.y, dz2 = qz - p.z;
            const float d2 = dx * dx + dy2 * dy2 + dz2 * dz2;
            const uint32_t id = __float_as_uint(p.w);
            if (d2 < best || (d2 == best && id < best_id)) { best = d2; best_id = id; }
          }
        }
      }
    }
    const float cover = (float)L * h;   // every point within `cover` of the query has been visited
    if (best <= cover * cover) break;
    if (cover * cover >= 25.0f) break;
  }
  return (best < 25.0f && best_id != 0xffffffffu) ? (int)best_id : -1;
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

__global__ __launch_bounds__(OD_THREADS) void k_odom_lm(OdomProblem* __restrict__ probs, OdomParams P) {
  OdomProblem& pb = probs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nSharp = (int)pb.n_sharp, nFlat = (int)pb.n_flat, nFeat = nSharp + nFlat;
  const int nLC = (int)pb.n_last_corner, nLS = (int)pb.n_last_surf;
  __shared__ float T[6];
  __shared__ float trig[6];
  __shared__ double red[OD_WAVES][LX_NSUM];
  __shared__ int sh_done, sh_deg;
  __shared__ float matP[36];
  if (tid < 6) T[tid] = pb.transform[tid];
  if (tid == 0) { sh_done = 0; sh_deg = 0; pb.stats.iterations = 0; pb.stats.sel = 0; pb.stats.degenerate = 0; }
  __syncthreads();
  const GridDesc gc = *pb.lc_desc, gs = *pb.ls_desc;

  for (int iter = 0; iter < P.max_iterations; iter++) {
    if (iter % 5 == 0) {
      // ---- phase A: nearest neighbour per feature
      for (int f = tid; f < nFeat; f += OD_THREADS) {
        const bool corner = f < nSharp;
        const float4 pi = corner ? pb.sharp[f] : pb.flat[f - nSharp];
        float x, y, z;
        transform_to_start(T, P.scan_period, pi, x, y, z);
        const int c = corner ? nn1(gc, pb.lc_sorted, pb.lc_cell, x, y, z) : nn1(gs, pb.ls_sorted, pb.ls_cell, x, y, z);
        pb.ind[5 * f] = c;
        pb.ind[5 * f + 1] = -1;
        pb.ind[5 * f + 2] = -1;
      }
      __syncthreads();
      // ---- phase B: ring-window scans, one wave per feature
      for (int f = wid; f < nFeat; f += OD_WAVES) {
        const int closest = pb.ind[5 * f];
        if (closest < 0) continue;   // wave-uniform
        const bool corner = f < nSharp;
        const float4 pi = corner ? pb.sharp[f] : pb.flat[f - nSharp];
        float x, y, z;
        transform_to_start(T, P.scan_period, pi, x, y, z);
        const float4* last = corner ? pb.last_corner : pb.last_surf;
        const int nLast = corner ? nLC : nLS;
        const int nCur = corner ? nSharp : nFlat;
        const int bound = nCur < nLast ? nCur : nLast;   // forward scans are bounded by the CURRENT feature count (:262, :378)
        const int cscan = (int)last[closest].w;
        float d2 = 25.f, d3 = 25.f;
        int j2 = -1, j3 = -1, o2 = 0x7fffffff, o3 = 0x7fffffff;
        for (int base = closest + 1; base < bound; base += 64) {
          const int j = base + lane;
          const bool in = j < bound;
          const float4 q = in ? last[j] : make_float4(0.f, 0.f, 0.f, 0.f);
          const int ring = (int)q.w;
          const bool brk = in && ((double)ring > (double)cscan + 2.5);
          const unsigned long long mb = __ballot(brk);
          const int fb = mb ? __builtin_ctzll(mb) : 64;
          if (in && lane < fb) {
            const float d = sqd(q, x, y, z);
            const int order = j - (closest + 1);
            if (corner) {
              if (ring > cscan && d < d2) { d2 = d; j2 = j; o2 = order; }
            } else {
              if (ring <= cscan) { if (d < d2) { d2 = d; j2 = j; o2 = order; } }
              else { if (d < d3) { d3 = d; j3 = j; o3 = order; } }
            }
          }
          if (mb) break;
        }
        for (int base = closest - 1; base >= 0; base -= 64) {
          const int j = base - lane;
          const bool in = j >= 0;
          const float4 q = in ? last[j] : make_float4(0.f, 0.f, 0.f, 0.f);
          const int ring = (int)q.w;
          const bool brk = in && ((double)ring < (double)cscan - 2.5);
          const unsigned long long mb = __ballot(brk);
          const int fb = mb ? __builtin_ctzll(mb) : 64;
          if (in && lane < fb) {
            const float d = sqd(q, x, y, z);
            const int order = 0x40000000 + (closest - 1 - j);   // backward candidates come after all forward ones
            if (corner) {
              if (ring < cscan && d < d2) { d2 = d; j2 = j; o2 = order; }
            } else {
              if (ring >= cscan) { if (d < d2) { d2 = d; j2 = j; o2 = order; } }
              else { if (d < d3) { d3 = d; j3 = j; o3 = order; } }
            }
          }
          if (mb) break;
        }
        wave_argmin(d2, j2, o2);
        if (!corner) wave_argmin(d3, j3, o3);
        if (lane == 0) {
          pb.ind[5 * f + 1] = j2;
          pb.ind[5 * f + 2] = corner ? -1 : j3;
        }
      }
      __syncthreads();
    }

    // ---- phase C: residual rows + normal equations
    if (tid == 0) {
      trig[0] = (float)sin((double)T[0]); trig[1] = (float)cos((double)T[0]);
      trig[2] = (float)sin((double)T[1]); trig[3] = (float)cos((double)T[1]);
      trig[4] = (float)sin((double)T[2]); trig[5] = (float)cos((double)T[2]);
    }
    __syncthreads();
    double v[LX_NSUM];
#pragma unroll
    for (int k = 0; k < LX_NSUM; k++) v[k] = 0.0;
    for (int f = tid; f < nFeat; f += OD_THREADS) {
      const bool corner = f < nSharp;
      const float4 po = corner ? pb.sharp[f] : pb.flat[f - nSharp];
      const int i1 = pb.ind[5 * f], i2 = pb.ind[5 * f + 1], i3 = pb.ind[5 * f + 2];
      float cx = 0.f, cy = 0.f, cz = 0.f, ci = 0.f;
      bool sel = false;
      if (corner ? (i2 >= 0) : (i2 >= 0 && i3 >= 0)) {
        float x0, y0, z0;
        transform_to_start(T, P.scan_period, po, x0, y0, z0);
        if (corner) {
          const float4 t1 = pb.last_corner[i1], t2 = pb.last_corner[i2];
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
          const float4 t1 = pb.last_surf[i1], t2 = pb.last_surf[i2], t3 = pb.last_surf[i3];
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
#pragma unroll
    for (int t = 0; t < LX_NSUM; t++) {
      double x = v[t];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
      if (lane == 0) red[wid][t] = x;
    }
    __syncthreads();
    if (tid == 0) {
      double sums[LX_NSUM];
      for (int t = 0; t < LX_NSUM; t++) {
        double x = 0.0;
        for (int w = 0; w < OD_WAVES; w++) x += red[w][t];
        sums[t] = x;
      }
      const int sel = (int)sums[27];
      pb.stats.iterations = iter + 1;
      pb.stats.sel = sel;
      if (sel >= 10) {   // :485-488
        float AtA[36], AtB[6], X[6];
        int k = 0;
        for (int i = 0; i < 6; i++)
          for (int j = i; j < 6; j++) { AtA[i * 6 + j] = AtA[j * 6 + i] = (float)sums[k]; k++; }
        for (int i = 0; i < 6; i++) AtB[i] = (float)sums[21 + i];
        qr_solve6(AtA, AtB, X);
        if (iter == 0) {
          sh_deg = degeneracy_projector(AtA, 10.f, matP) ? 1 : 0;
          pb.stats.degenerate = sh_deg;
        }
        if (sh_deg) {
          float X2[6];
          for (int r = 0; r < 6; r++) X2[r] = X[r];
          for (int r = 0; r < 6; r++) {
            float acc = 0.f;
            for (int c = 0; c < 6; c++) acc += matP[r * 6 + c] * X2[c];
            X[r] = acc;
          }
        }
        for (int r = 0; r < 6; r++) {
          float nv = T[r] + X[r];
          if (!isfinite(nv)) nv = 0.f;   // :606-612
          T[r] = nv;
        }
        const float d0 = (float)(X[0] * 180.0 / M_PI), d1 = (float)(X[1] * 180.0 / M_PI), d2 = (float)(X[2] * 180.0 / M_PI);
        const float deltaR = (float)sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);
        const float t0 = X[3] * 100, t1 = X[4] * 100, t2 = X[5] * 100;
        const float deltaT = (float)sqrt((double)t0 * t0 + (double)t1 * t1 + (double)t2 * t2);
        if (deltaR < P.delta_r_abort && deltaT < P.delta_t_abort) sh_done = 1;
      }
    }
    __syncthreads();
    if (sh_done) break;
  }
  if (tid < 6) pb.transform[tid] = T[tid];
}

struct ToEndParams {
  float T[6];
  float sT[3], cT[3];                   // sin/cos of the transform angles (x, y, z)
  float shift[3];                       // imuShiftFromStart
  float s_start[3], c_start[3];         // imu pitch/yaw/roll start (x=pitch, y=yaw, z=roll)
  float s_end[3], c_end[3];
  float scan_period;
};

// transformToEnd (:57-87)
__global__ __launch_bounds__(256) void k_transform_to_end(float4* __restrict__ pts, uint32_t n, ToEndParams P) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
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
  pts[i] = make_float4(x, y, z, (float)(int)p.w);
}

// ----------------------------------------------------------------------------------------------------------------
Odometry::Odometry(int device) : device_(device) {
  select_device(device);
  LX_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  idx_corner_.init(st_);
  idx_surf_.init(st_);
  prob_.reserve(1);
  h_prob_.reserve(1);
}

Odometry::~Odometry() {
  if (st_) (void)hipStreamDestroy(st_);
}

void Odometry::update_imu(const float* t) {
  imu_pitch_start_ = HAngle(t[0]); imu_yaw_start_ = HAngle(t[1]); imu_roll_start_ = HAngle(t[2]);
  imu_pitch_end_ = HAngle(t[3]); imu_yaw_end_ = HAngle(t[4]); imu_roll_end_ = HAngle(t[5]);
  imu_shift_ = {t[6], t[7], t[8]};
  imu_velo_ = {t[9], t[10], t[11]};
}

void Odometry::upload_cloud(const loamx_cloud* c, DevBuf<float4>& dst) {
  check_cloud(c, false);
  h_stage_.reserve(c->count + 1);
  dst.reserve(c->count + 1);
  pack_cloud(c, h_stage_.p);
  if (c->count) LX_HIP(hipMemcpyAsync(dst.p, h_stage_.p, sizeof(float4) * c->count, hipMemcpyHostToDevice, st_));
  LX_HIP(hipStreamSynchronize(st_));   // the single staging buffer is reused
}

void Odometry::to_end_device(float4* pts, uint32_t n) {
  if (!n) return;
  ToEndParams P;
  transform_.get(P.T);
  const HAngle* ta[3] = {&transform_.rot_x, &transform_.rot_y, &transform_.rot_z};
  const HAngle* sa[3] = {&imu_pitch_start_, &imu_yaw_start_, &imu_roll_start_};
  const HAngle* ea[3] = {&imu_pitch_end_, &imu_yaw_end_, &imu_roll_end_};
  for (int k = 0; k < 3; k++) {
    P.sT[k] = ta[k]->s; P.cT[k] = ta[k]->c;
    P.s_start[k] = sa[k]->s; P.c_start[k] = sa[k]->c;
    P.s_end[k] = ea[k]->s; P.c_end[k] = ea[k]->c;
  }
  P.shift[0] = imu_shift_.x; P.shift[1] = imu_shift_.y; P.shift[2] = imu_shift_.z;
  P.scan_period = params.scan_period;
  hipLaunchKernelGGL(k_transform_to_end, dim3((n + 255) / 256), dim3(256), 0, st_, pts, n, P);
}

int Odometry::process(const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat, const loamx_cloud* less_flat) {
  LX_HIP(hipSetDevice(device_));
  upload_cloud(sharp, sharp_);
  upload_cloud(less_sharp, less_sharp_);
  upload_cloud(flat, flat_);
  upload_cloud(less_flat, less_flat_);
  const uint32_t nSharp = sharp->count, nLessSharp = less_sharp->count, nFlat = flat->count, nLessFlat = less_flat->count;

  if (!inited_) {   // :198-211
    std::swap(less_sharp_.p, last_corner_.p); std::swap(less_sharp_.cap, last_corner_.cap);
    std::swap(less_flat_.p, last_surf_.p); std::swap(less_flat_.cap, last_surf_.cap);
    n_last_corner_ = nLessSharp;
    n_last_surf_ = nLessFlat;
    idx_corner_.build(last_corner_.p, n_last_corner_);
    idx_surf_.build(last_surf_.p, n_last_surf_);
    transform_sum_.rot_x = HAngle(transform_sum_.rot_x.r + imu_pitch_start_.r);
    transform_sum_.rot_z = HAngle(transform_sum_.rot_z.r + imu_roll_start_.r);
    inited_ = true;
    LX_HIP(hipStreamSynchronize(st_));
    return LOAMX_SKIPPED;
  }
  frame_++;
  transform_.pos.x -= imu_velo_.x * params.scan_period;
  transform_.pos.y -= imu_velo_.y * params.scan_period;
  transform_.pos.z -= imu_velo_.z * params.scan_period;
  stats_ = {0, 0, (int)frame_, 0};

  if (n_last_corner_ > 10 && n_last_surf_ > 100) {
    ind_.reserve((size_t)5 * (nSharp + nFlat) + 5);
    OdomProblem& pb = *h_prob_.p;
    pb.sharp = sharp_.p; pb.n_sharp = nSharp;
    pb.flat = flat_.p; pb.n_flat = nFlat;
    pb.last_corner = last_corner_.p; pb.n_last_corner = n_last_corner_;
    pb.last_surf = last_surf_.p; pb.n_last_surf = n_last_surf_;
    pb.lc_sorted = idx_corner_.sorted(); pb.lc_cell = idx_corner_.cell_start(); pb.lc_desc = idx_corner_.desc();
    pb.ls_sorted = idx_surf_.sorted(); pb.ls_cell = idx_surf_.cell_start(); pb.ls_desc = idx_surf_.desc();
    pb.ind = ind_.p;
    transform_.get(pb.transform);
    pb.stats = {0, 0, 0, 0};
    LX_HIP(hipMemcpyAsync(prob_.p, h_prob_.p, sizeof(OdomProblem), hipMemcpyHostToDevice, st_));
    hipLaunchKernelGGL(k_odom_lm, dim3(1), dim3(OD_THREADS), 0, st_, prob_.p, params);
    LX_HIP(hipMemcpyAsync(h_prob_.p, prob_.p, sizeof(OdomProblem), hipMemcpyDeviceToHost, st_));
    LX_HIP(hipStreamSynchronize(st_));
    // _transform.rot_* = rad + x re-derives the cached sin/cos (:599-601)
    transform_.set(h_prob_.p->transform);
    stats_.iterations = h_prob_.p->stats.iterations;
    stats_.sel = h_prob_.p->stats.sel;
    stats_.degenerate = h_prob_.p->stats.degenerate;
  }

  // pose integration (:626-649)
  HAngle rx, ry, rz;
  accumulate_rotation(transform_sum_.rot_x, transform_sum_.rot_y, transform_sum_.rot_z, -transform_.rot_x,
                      HAngle((float)(-transform_.rot_y.r * 1.05)), -transform_.rot_z, rx, ry, rz);
  HVec3 v{transform_.pos.x - imu_shift_.x, transform_.pos.y - imu_shift_.y, (float)(transform_.pos.z * 1.05 - imu_shift_.z)};
  h_rot_zxy(v, rz, rx, ry);
  HVec3 trans{transform_sum_.pos.x - v.x, transform_sum_.pos.y - v.y, transform_sum_.pos.z - v.z};
  plugin_imu_rotation(rx, ry, rz, imu_pitch_start_, imu_yaw_start_, imu_roll_start_, imu_pitch_end_, imu_yaw_end_, imu_roll_end_, rx, ry, rz);
  transform_sum_.rot_x = rx; transform_sum_.rot_y = ry; transform_sum_.rot_z = rz;
  transform_sum_.pos = trans;

  // re-project to the sweep end and hand over as "last" clouds (:651-664)
  to_end_device(less_sharp_.p, nLessSharp);
  to_end_device(less_flat_.p, nLessFlat);
  std::swap(less_sharp_.p, last_corner_.p); std::swap(less_sharp_.cap, last_corner_.cap);
  std::swap(less_flat_.p, last_surf_.p); std::swap(less_flat_.cap, last_surf_.cap);
  n_last_corner_ = nLessSharp;
  n_last_surf_ = nLessFlat;
  if (n_last_corner_ > 10 && n_last_surf_ > 100) {
    idx_corner_.build(last_corner_.p, n_last_corner_);
    idx_surf_.build(last_surf_.p, n_last_surf_);
  }
  LX_HIP(hipStreamSynchronize(st_));
  return LOAMX_OK;
}

int Odometry::get_last_clouds(loamx_cloud* corner, loamx_cloud* surf) {
  LX_HIP(hipSetDevice(device_));
  int rc = LOAMX_OK;
  std::vector<float4> tmp;
  if (corner) {
    check_cloud(corner, false);
    tmp.resize(n_last_corner_);
    if (n_last_corner_) LX_HIP(hipMemcpy(tmp.data(), last_corner_.p, sizeof(float4) * n_last_corner_, hipMemcpyDeviceToHost));
    int r = unpack_cloud(tmp.data(), n_last_corner_, corner);
    if (r != LOAMX_OK) rc = r;
  }
  if (surf) {
    check_cloud(surf, false);
    tmp.resize(n_last_surf_);
    if (n_last_surf_) LX_HIP(hipMemcpy(tmp.data(), last_surf_.p, sizeof(float4) * n_last_surf_, hipMemcpyDeviceToHost));
    int r = unpack_cloud(tmp.data(), n_last_surf_, surf);
    if (r != LOAMX_OK) rc = r;
  }
  return rc;
}

int Odometry::transform_to_end(loamx_cloud* cloud) {
  LX_HIP(hipSetDevice(device_));
  check_cloud(cloud, false);
  const uint32_t n = cloud->count;
  upload_cloud(cloud, tmp_cloud_);
  to_end_device(tmp_cloud_.p, n);
  std::vector<float4> tmp(n);
  if (n) LX_HIP(hipMemcpyAsync(tmp.data(), tmp_cloud_.p, sizeof(float4) * n, hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
  return unpack_cloud(tmp.data(), n, cloud);
}

}  // namespace loamx
