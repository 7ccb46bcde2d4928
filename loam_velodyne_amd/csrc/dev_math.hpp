// Device-side scalar math of the registration path (gfx950).  float32 with the reference's double promotions;
// compiled with -ffp-contract=off so a*b+c is two roundings exactly as written in the reference expressions.
//   Pose / rotations        <-> reference include/loam_velodyne/Angle.h:16-67, src/lib/math_utils.h:129-275
//   eig3_sym (Jacobi)       <-> Eigen::SelfAdjointEigenSolver<Matrix3f> at BasicLaserMapping.cpp:695-697
//   qr_solve (col. pivoted) <-> colPivHouseholderQr().solve at BasicLaserMapping.cpp:768, :867, BasicLaserOdometry.cpp:559
//   eig6 / inverse6 / projector <-> BasicLaserMapping.cpp:869-899, BasicLaserOdometry.cpp:561-591
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>

namespace loamx {

// A pose with the cached float sin/cos the reference's Angle carries.
struct Pose {
  float rx, ry, rz, tx, ty, tz;
  float srx, crx, sry, cry, srz, crz;
};

// correctly-rounded float sin/cos through double (matches glibc sinf/cosf on all but a vanishing set of inputs)
__host__ __device__ inline void pose_set_angles(Pose& p, float rx, float ry, float rz) {
  p.rx = rx; p.ry = ry; p.rz = rz;
  p.srx = (float)sin((double)rx); p.crx = (float)cos((double)rx);
  p.sry = (float)sin((double)ry); p.cry = (float)cos((double)ry);
  p.srz = (float)sin((double)rz); p.crz = (float)cos((double)rz);
}

// rotX/rotY/rotZ with explicit (cos, sin) — math_utils.h:129-201
__host__ __device__ inline void rot_x(float& y, float& z, float c, float s) {
  float y0 = y;
  y = c * y0 - s * z;
  z = s * y0 + c * z;
}
__host__ __device__ inline void rot_y(float& x, float& z, float c, float s) {
  float x0 = x;
  x = c * x0 + s * z;
  z = c * z - s * x0;
}
__host__ __device__ inline void rot_z(float& x, float& y, float c, float s) {
  float x0 = x;
  x = c * x0 - s * y;
  y = s * x0 + c * y;
}

// pointAssociateToMap — BasicLaserMapping.cpp:207-219
__host__ __device__ inline void to_map(const Pose& T, float& x, float& y, float& z) {
  rot_z(x, y, T.crz, T.srz);
  rot_x(y, z, T.crx, T.srx);
  rot_y(x, z, T.cry, T.sry);
  x += T.tx; y += T.ty; z += T.tz;
}
// pointAssociateTobeMapped — BasicLaserMapping.cpp:223-231 (negated angles: sine flips, cosine kept)
__host__ __device__ inline void to_be_mapped(const Pose& T, float& x, float& y, float& z) {
  x = x - T.tx; y = y - T.ty; z = z - T.tz;
  rot_y(x, z, T.cry, -T.sry);
  rot_x(y, z, T.crx, -T.srx);
  rot_z(x, y, T.crz, -T.srz);
}

// ---- symmetric 3x3 eigen-decomposition, cyclic Jacobi.  In: lower triangle a00,a10,a11,a20,a21,a22.
// Out: eigenvalues ascending w0<=w1<=w2 and the unit eigenvector of the LARGEST one.
__device__ __forceinline__ void eig3_sym(float a00, float a10, float a11, float a20, float a21, float a22, float& w0, float& w1,
                                float& w2, float& vx, float& vy, float& vz) {
  float A[3][3] = {{a00, a10, a20}, {a10, a11, a21}, {a20, a21, a22}};
  float Q[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
  for (int sweep = 0; sweep < 16; sweep++) {
    float off = A[1][0] * A[1][0] + A[2][0] * A[2][0] + A[2][1] * A[2][1];
    float diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-20f * diag || off == 0.f) break;
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
      for (int q = p + 1; q < 3; q++) {
        float apq = A[p][q];
        if (apq != 0.f) {
          float theta = (A[q][q] - A[p][p]) / (2.f * apq);
          float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
          float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            float akp = A[k][p], akq = A[k][q];
            A[k][p] = c * akp - s * akq;
            A[k][q] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < 3; k++) {
            float apk = A[p][k], aqk = A[q][k];
            A[p][k] = c * apk - s * aqk;
            A[q][k] = s * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < 3; k++) {
            float qkp = Q[k][p], qkq = Q[k][q];
            Q[k][p] = c * qkp - s * qkq;
            Q[k][q] = s * qkp + c * qkq;
          }
        }
      }
    }
  }
  float e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
  float c0x = Q[0][0], c0y = Q[1][0], c0z = Q[2][0];
  float c1x = Q[0][1], c1y = Q[1][1], c1z = Q[2][1];
  float c2x = Q[0][2], c2y = Q[1][2], c2z = Q[2][2];
#define LX_CSWAP(ea, eb, ax, ay, az, bx, by, bz) \
  if (eb < ea) {                                 \
    float t_;                                    \
    t_ = ea; ea = eb; eb = t_;                   \
    t_ = ax; ax = bx; bx = t_;                   \
    t_ = ay; ay = by; by = t_;                   \
    t_ = az; az = bz; bz = t_;                   \
  }
  LX_CSWAP(e0, e1, c0x, c0y, c0z, c1x, c1y, c1z)
  LX_CSWAP(e1, e2, c1x, c1y, c1z, c2x, c2y, c2z)
  LX_CSWAP(e0, e1, c0x, c0y, c0z, c1x, c1y, c1z)
#undef LX_CSWAP
  w0 = e0; w1 = e1; w2 = e2;
  vx = c2x; vy = c2y; vz = c2z;
}

// ---- the same decomposition in closed form, double precision: eigenvalues by the trigonometric solution of the characteristic
// cubic, the eigenvector of the largest one as the best-conditioned cross product of two rows of A - w2 I.  No iteration: ~300
// dependent instructions against the ~4000 of eight Jacobi sweeps (the single-lane latency of a corner tile, k_gn_iter).  The
// reference's SelfAdjointEigenSolver<Matrix3f> (tridiagonal QL in float) and the Jacobi restatement above both approximate what
// this computes to ~1e-15: the exact eigen-pairs of the float matrix, rounded to float at the end.  The vector is only defined
// up to sign (the edge residual does not depend on it) and only used when w2 > 3 w1, where it is well conditioned.
__device__ __forceinline__ void eig3_sym_direct(float a00f, float a10f, float a11f, float a20f, float a21f, float a22f, float& w0, float& w1,
                                       float& w2, float& vx, float& vy, float& vz) {
  const double a00 = a00f, a01 = a10f, a11 = a11f, a02 = a20f, a12 = a21f, a22 = a22f;
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const double q = (a00 + a11 + a22) / 3.0;
  const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
  double e0, e1, e2;
  if (!(p2 > 0.0)) {   // a multiple of the identity (or zero)
    e0 = e1 = e2 = q;
  } else {
    const double p = sqrt(p2 / 6.0), ip = 1.0 / p;
    const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
    double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    e2 = q + 2.0 * p * cos(phi);
    e0 = q + 2.0 * p * cos(phi + 2.0943951023931954923);   // + 2 pi / 3
    e1 = 3.0 * q - e2 - e0;
  }
  w0 = (float)e0; w1 = (float)e1; w2 = (float)e2;
  // rows of A - e2 I
  const double r0x = a00 - e2, r0y = a01, r0z = a02;
  const double r1x = a01, r1y = a11 - e2, r1z = a12;
  const double r2x = a02, r2y = a12, r2z = a22 - e2;
  const double ax = r0y * r1z - r0z * r1y, ay = r0z * r1x - r0x * r1z, az = r0x * r1y - r0y * r1x;   // r0 x r1
  const double bx = r0y * r2z - r0z * r2y, by = r0z * r2x - r0x * r2z, bz = r0x * r2y - r0y * r2x;   // r0 x r2
  const double cx = r1y * r2z - r1z * r2y, cy = r1z * r2x - r1x * r2z, cz = r1x * r2y - r1y * r2x;   // r1 x r2
  const double na = ax * ax + ay * ay + az * az, nb = bx * bx + by * by + bz * bz, nc = cx * cx + cy * cy + cz * cz;
  double ux = ax, uy = ay, uz = az, nu = na;
  if (nb > nu) { ux = bx; uy = by; uz = bz; nu = nb; }
  if (nc > nu) { ux = cx; uy = cy; uz = cz; nu = nc; }
  if (nu > 0.0) {
    const double inv = 1.0 / sqrt(nu);
    vx = (float)(ux * inv); vy = (float)(uy * inv); vz = (float)(uz * inv);
  } else {
    vx = 0.f; vy = 0.f; vz = 1.f;
  }
}

// ---- column-pivoted Householder QR least squares, fully unrolled so A stays in registers.
template <int M, int N> __device__ __forceinline__ void qr_solve(float (&A)[M][N], float (&b)[M], float (&x)[N]) {
  int perm[N];
#pragma unroll
  for (int c = 0; c < N; c++) perm[c] = c;
  // (maxima as Eigen's visitor finds them: seeded with the first coefficient, replaced on `value > current` only — a NaN in front
  // survives, the threshold becomes NaN, no pivot is declared negligible and a non-finite system gets a non-finite solution)
  float maxnorm = 0.f;
#pragma unroll
  for (int c = 0; c < N; c++) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < M; r++) s += A[r][c] * A[r][c];
    const float nrm = sqrtf(s);
    maxnorm = (c == 0 || nrm > maxnorm) ? nrm : maxnorm;
  }
  const float eps = FLT_EPSILON;
  const float thr_helper = (maxnorm * eps) * (maxnorm * eps) / float(M);
  int nonzero = N;
#pragma unroll
  for (int k = 0; k < N; k++) {
    int best = k;
    float bestn = 0.f;
#pragma unroll
    for (int c = k; c < N; c++) {
      float s = 0.f;
#pragma unroll
      for (int r = k; r < M; r++) s += A[r][c] * A[r][c];
      if (c == k || s > bestn) { bestn = s; best = c; }
    }
    if (nonzero == N && bestn < thr_helper * float(M - k)) nonzero = k;
#pragma unroll
    for (int c = k + 1; c < N; c++) {
      if (best == c) {
#pragma unroll
        for (int r = 0; r < M; r++) { float t = A[r][k]; A[r][k] = A[r][c]; A[r][c] = t; }
        int t = perm[k]; perm[k] = perm[c]; perm[c] = t;
      }
    }
    float c0 = A[k][k], tail = 0.f;
#pragma unroll
    for (int r = k + 1; r < M; r++) tail += A[r][k] * A[r][k];
    float tau, beta;
    float v[M];
    if (tail <= FLT_MIN) {
      tau = 0.f;
      beta = c0;
#pragma unroll
      for (int r = k + 1; r < M; r++) v[r] = 0.f;
    } else {
      beta = sqrtf(c0 * c0 + tail);
      if (c0 >= 0.f) beta = -beta;
#pragma unroll
      for (int r = k + 1; r < M; r++) v[r] = A[r][k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    v[k] = 1.f;
    A[k][k] = beta;
#pragma unroll
    for (int r = k + 1; r < M; r++) A[r][k] = 0.f;
#pragma unroll
    for (int c = k + 1; c < N; c++) {
      float dot = 0.f;
#pragma unroll
      for (int r = k; r < M; r++) dot += v[r] * A[r][c];
      dot *= tau;
#pragma unroll
      for (int r = k; r < M; r++) A[r][c] -= dot * v[r];
    }
    {
      float dot = 0.f;
#pragma unroll
      for (int r = k; r < M; r++) dot += v[r] * b[r];
      dot *= tau;
#pragma unroll
      for (int r = k; r < M; r++) b[r] -= dot * v[r];
    }
  }
  float y[N];
#pragma unroll
  for (int c = 0; c < N; c++) y[c] = 0.f;
#pragma unroll
  for (int k = N - 1; k >= 0; k--) {
    if (k < nonzero) {
      float s = b[k];
#pragma unroll
      for (int c = k + 1; c < N; c++)
        if (c < nonzero) s -= A[k][c] * y[c];
      y[k] = s / A[k][k];
    }
  }
#pragma unroll
  for (int c = 0; c < N; c++) x[c] = 0.f;
#pragma unroll
  for (int c = 0; c < N; c++)
#pragma unroll
    for (int j = 0; j < N; j++)
      if (c < nonzero && perm[c] == j) x[j] = y[c];
}

// ---- 6x6 helpers for the once-per-iteration solve (one thread).  Work arrays live in a caller-provided LDS
// workspace `ws` (>= 216 floats): dynamically indexed private arrays would be placed in scratch (global memory), whose
// latency dominates a serial solve.
__device__ __forceinline__ void eig6_sym(const float* Ain, float* w, float* V, float* ws) {
  float* A = ws;        // 36
  float* Q = ws + 36;   // 36
  for (int r = 0; r < 6; r++)
    for (int c = 0; c <= r; c++) A[r * 6 + c] = A[c * 6 + r] = Ain[r * 6 + c];
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) Q[r * 6 + c] = (r == c) ? 1.f : 0.f;
  for (int sweep = 0; sweep < 16; sweep++) {
    float off = 0.f, diag = 0.f;
    for (int r = 0; r < 6; r++) {
      diag += A[r * 6 + r] * A[r * 6 + r];
      for (int c = 0; c < r; c++) off += A[r * 6 + c] * A[r * 6 + c];
    }
    if (off <= 1e-20f * diag || off == 0.f) break;
    for (int p = 0; p < 5; p++)
      for (int q = p + 1; q < 6; q++) {
        float apq = A[p * 6 + q];
        if (apq == 0.f) continue;
        float theta = (A[q * 6 + q] - A[p * 6 + p]) / (2.f * apq);
        float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
        float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
        for (int k = 0; k < 6; k++) {
          float akp = A[k * 6 + p], akq = A[k * 6 + q];
          A[k * 6 + p] = c * akp - s * akq;
          A[k * 6 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 6; k++) {
          float apk = A[p * 6 + k], aqk = A[q * 6 + k];
          A[p * 6 + k] = c * apk - s * aqk;
          A[q * 6 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 6; k++) {
          float qkp = Q[k * 6 + p], qkq = Q[k * 6 + q];
          Q[k * 6 + p] = c * qkp - s * qkq;
          Q[k * 6 + q] = s * qkp + c * qkq;
        }
      }
  }
  int order[6];
#pragma unroll
  for (int k = 0; k < 6; k++) order[k] = k;
  // insertion sort ascending, stable (small fixed network, compile-time indices)
#pragma unroll
  for (int a_ = 1; a_ < 6; a_++) {
#pragma unroll
    for (int b_ = 5; b_ > 0; b_--) {
      if (b_ <= a_) {
        const float eb = A[order[b_] * 6 + order[b_]], ea = A[order[b_ - 1] * 6 + order[b_ - 1]];
        if (eb < ea) { int t = order[b_]; order[b_] = order[b_ - 1]; order[b_ - 1] = t; }
      }
    }
  }
  for (int k = 0; k < 6; k++) {
    w[k] = A[order[k] * 6 + order[k]];
    for (int r = 0; r < 6; r++) V[r * 6 + k] = Q[r * 6 + order[k]];
  }
}

__device__ __forceinline__ bool inverse6(const float* Ain, float* inv, float* ws) {
  float* A = ws;   // 6 x 12
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      A[r * 12 + c] = Ain[r * 6 + c];
      A[r * 12 + 6 + c] = (r == c) ? 1.f : 0.f;
    }
  for (int k = 0; k < 6; k++) {
    int piv = k;
    for (int r = k + 1; r < 6; r++)
      if (fabsf(A[r * 12 + k]) > fabsf(A[piv * 12 + k])) piv = r;
    if (A[piv * 12 + k] == 0.f) return false;
    if (piv != k)
      for (int c = 0; c < 12; c++) { float t = A[k * 12 + c]; A[k * 12 + c] = A[piv * 12 + c]; A[piv * 12 + c] = t; }
    float d = 1.f / A[k * 12 + k];
    for (int c = 0; c < 12; c++) A[k * 12 + c] *= d;
    for (int r = 0; r < 6; r++)
      if (r != k) {
        float f = A[r * 12 + k];
        if (f != 0.f)
          for (int c = 0; c < 12; c++) A[r * 12 + c] -= f * A[k * 12 + c];
      }
  }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) inv[r * 6 + c] = A[r * 12 + 6 + c];
  return true;
}

// P = V^-1 * V2, V2 = V with ROW i zeroed while ascending eigenvalue i < thr (BasicLaserMapping.cpp:875-898)
// Certificate that every eigenvalue of the symmetric 6x6 AtA exceeds thr*(1+1e-3): a Cholesky factorisation of
// AtA - thr*(1+1e-3)*I in double succeeds with pivots well above rounding noise.  When it holds the reference's loop
// (BasicLaserMapping.cpp:883-897) zeroes nothing, isDegenerate stays false and matP is never used, so the 6x6
// eigen-decomposition can be skipped; borderline and degenerate cases take the full path below.
__device__ __forceinline__ bool certainly_not_degenerate(const float* AtA, float thr) {
  double L[6][6];
  const double shift = (double)thr * 1.001;
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) scale = fmax(scale, fabs((double)AtA[i * 6 + i]));
  const double tiny = 1e-6 * scale;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = (double)AtA[j * 6 + j] - shift;
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < j) dj -= L[j][k] * L[j][k];
    if (!(dj > tiny)) return false;
    const double lj = sqrt(dj);
    L[j][j] = lj;
#pragma unroll
    for (int i = 0; i < 6; i++)
      if (i > j) {
        double s = (double)AtA[i * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (k < j) s -= L[i][k] * L[j][k];
        L[i][j] = s / lj;
      }
  }
  return true;
}

__device__ __forceinline__ bool degeneracy_projector_full(const float* AtA, float thr, float* P, float* ws);
__device__ __forceinline__ bool degeneracy_projector(const float* AtA, float thr, float* P, float* ws) {
  if (certainly_not_degenerate(AtA, thr)) return false;
  return degeneracy_projector_full(AtA, thr, P, ws);
}
// the reference's computation itself (the caller has found no certificate)
__device__ __forceinline__ bool degeneracy_projector_full(const float* AtA, float thr, float* P, float* ws) {
  float w[6];
  float* V = ws + 72;     // 36
  float* V2 = ws + 108;   // 36
  float* Vi = ws + 144;   // 36
  eig6_sym(AtA, w, V, ws);
  for (int k = 0; k < 36; k++) V2[k] = V[k];
  bool degenerate = false;
  {
    bool go = true;   // ascending order: stop at the first eigenvalue that reaches the threshold
#pragma unroll
    for (int i = 0; i < 6; i++) {
      if (go && w[i] < thr) {
        for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0.f;
        degenerate = true;
      } else {
        go = false;
      }
    }
  }
  if (!inverse6(V, Vi, ws)) {
    for (int k = 0; k < 36; k++) P[k] = (k % 7 == 0) ? 1.f : 0.f;
    return degenerate;
  }
  for (int r = 0; r < 6; r++)
    for (int c = 0; c < 6; c++) {
      float s = 0.f;
      for (int k = 0; k < 6; k++) s += Vi[r * 6 + k] * V2[k * 6 + c];
      P[r * 6 + c] = s;
    }
  return degenerate;
}

// 6x6 column-pivoted Householder QR solve spread over the lanes of a wave: lane c < 6 owns COLUMN c of A, lane 6 owns the
// right-hand side; every arithmetic operation is the one the scalar qr_solve<6,6> performs on that element, in the same
// order, so the result is bit-identical (tests/test_gpu_parity_hooks: loamx_debug_qr6) — only the serial dependency chain shrinks.
// Values travel between lanes with v_readlane (a few cycles) instead of ds_bpermute (~60 cycles each on the critical path).
// Must be called by ALL 64 lanes of wave 0 of the workgroup; AtA/AtB/X are in LDS or global memory.
__device__ __forceinline__ float lane_get(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }

// qr_solve6_lanes: the solution stays in the lanes — lane c < 6 returns x[c], the others 0 (k_odom_lm goes on from there in registers)
__device__ __forceinline__ float qr_solve6_lanes(const float* AtA, const float* AtB) {
  const int gl = (int)(threadIdx.x & 63);
  float a[6];
#pragma unroll
  for (int r = 0; r < 6; r++) a[r] = gl < 6 ? AtA[r * 6 + gl] : (gl == 6 ? AtB[r] : 0.f);
  // Columns never move between lanes: L[c] = the lane (= original column) that stands at POSITION c of the pivoted matrix.  A column
  // swap is a swap of two entries of L — uniform integers — instead of twelve lane reads and selects, and everything a lane derives
  // from its own column (remaining norm, Householder vector) is computed before the pivot is known, off the pivot search's chain.
  int L[6] = {0, 1, 2, 3, 4, 5};
  bool pivoted = false;   // this lane's column has been a pivot (its entries from the diagonal down are final)
  // largest initial column norm
  float s0 = 0.f;
#pragma unroll
  for (int r = 0; r < 6; r++) s0 += a[r] * a[r];
  const float nrm = gl < 6 ? sqrtf(s0) : 0.f;
  float maxnorm = lane_get(nrm, 0);   // (seeded with the first, replaced on `>` only: Eigen's visitor, see qr_solve)
#pragma unroll
  for (int c = 1; c < 6; c++) {
    const float v = lane_get(nrm, c);
    maxnorm = v > maxnorm ? v : maxnorm;
  }
  const float thr_helper = (maxnorm * FLT_EPSILON) * (maxnorm * FLT_EPSILON) / 6.0f;
  int nonzero = 6;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    // every lane, for its own column: the remaining squared norm (rows k..5) and the Householder vector from row k
    float s = 0.f, tail = 0.f;
#pragma unroll
    for (int r = k; r < 6; r++) s += a[r] * a[r];
#pragma unroll
    for (int r = k + 1; r < 6; r++) tail += a[r] * a[r];
    const float c0 = a[k];
    float tau, beta, v[6];
    if (tail <= FLT_MIN) {
      tau = 0.f;
      beta = c0;
#pragma unroll
      for (int r = 0; r < 6; r++) v[r] = 0.f;
    } else {
      beta = sqrtf(c0 * c0 + tail);
      if (c0 >= 0.f) beta = -beta;
#pragma unroll
      for (int r = 0; r < 6; r++) v[r] = r > k ? a[r] / (c0 - beta) : 0.f;
      tau = (beta - c0) / beta;
    }
    v[k] = 1.f;
    // pivot: first position (from k upwards) whose column has the largest remaining squared norm
    int best = k;
    float bestn = lane_get(s, L[k]);
#pragma unroll
    for (int c = k + 1; c < 6; c++) {
      const float sc = lane_get(s, L[c]);
      if (sc > bestn) { bestn = sc; best = c; }
    }
    best = __builtin_amdgcn_readfirstlane(best);
    if (nonzero == 6 && bestn < thr_helper * float(6 - k)) nonzero = k;
    {   // positions k and best change places
      int lb = L[k];
#pragma unroll
      for (int c = k + 1; c < 6; c++) lb = best == c ? L[c] : lb;
#pragma unroll
      for (int c = k + 1; c < 6; c++) L[c] = best == c ? L[k] : L[c];
      L[k] = __builtin_amdgcn_readfirstlane(lb);
    }
    const int pl = L[k];   // the pivot column's lane: its Householder vector is the step's
    tau = lane_get(tau, pl);
    beta = lane_get(beta, pl);
#pragma unroll
    for (int r = k + 1; r < 6; r++) v[r] = lane_get(v[r], pl);
    if (gl == pl) {
      pivoted = true;
      a[k] = beta;
#pragma unroll
      for (int r = k + 1; r < 6; r++) a[r] = 0.f;
    } else if (gl <= 6 && !pivoted) {   // remaining columns and the right-hand side
      float dot = 0.f;
#pragma unroll
      for (int r = k; r < 6; r++) dot += v[r] * a[r];
      dot *= tau;
#pragma unroll
      for (int r = k; r < 6; r++) a[r] -= dot * v[r];
    }
  }
  // back substitution on the leading nonzero x nonzero block: y[k] ends up in the lane at position k
  float y = 0.f;
#pragma unroll
  for (int k = 5; k >= 0; k--) {
    float sacc = lane_get(a[k], 6);   // b[k]
#pragma unroll
    for (int c = k + 1; c < 6; c++) {
      const float term = lane_get(a[k] * y, L[c]);   // A[k][c] * y[c] from the lane whose column stands at position c
      if (c < nonzero) sacc -= term;
    }
    const float diag = lane_get(a[k], L[k]);
    if (gl == L[k] && k < nonzero) y = sacc / diag;
  }
  return gl < 6 ? y : 0.f;   // (a lane is its original column; positions >= nonzero never received a value: 0)
}
__device__ __forceinline__ void qr_solve6_coop(const float* AtA, const float* AtB, float* X) {
  const float y = qr_solve6_lanes(AtA, AtB);
  if ((threadIdx.x & 63) < 6) X[threadIdx.x & 63] = y;
}

// ---- exchange between workgroups without cache-wide fences (round 4).  __threadfence() on gfx950 is buffer_wbl2 sc1 + buffer_inv sc1:
// it writes back and INVALIDATES the XCD's whole L2 — the L2 the other HIP streams' kernels are living in (k_gn_iter's map, the
// odometry's clouds).  k_gn_iter executed it ~2,400 times per launch, k_odom_lm ~700.  What an exchange of a few dozen words needs is
// much less: the producer stores its words with agent-scope (sc1, write-through) stores and waits for them to complete before it bumps
// the arrival counter; the consumer reads them with agent-scope loads (which do not stop at the non-coherent L2).
__device__ __forceinline__ void xchg_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xchg_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // (gfx9: vmcnt counts stores too)
// agent-scope load issued WITHOUT a wait: the compiler emits atomic loads one at a time (a wait behind each); n of these followed by
// one xchg_loads_done() are n loads in flight
__device__ __forceinline__ double xchg_load_nowait(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// A double with a tag on both sides in ONE aligned 16-byte record, written by one store and read by one load: the reader takes the value
// when BOTH tags are the one it waits for.  An aligned 16-byte access never leaves its cache line; should the memory system ever
// split it, it splits at 8 bytes, and each half carries a tag — so a record with two matching tags holds both halves of the new
// value.  The exchange then needs no separate flag, no counter and no wait on the producer's side (k_odom_lm).
typedef unsigned xrec_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void xrec_store(xrec_t* p, double v, unsigned tag) {
  xrec_t r;
  r.x = tag; r.y = (unsigned)__double2loint(v); r.z = (unsigned)__double2hiint(v); r.w = tag;
  // (s_nop 1: a store of more than 64 bits must not be followed at once by a write to its data registers — the compiler pads its own
  // stores, it cannot see into this statement)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ xrec_t xrec_load(const xrec_t* p) {
  xrec_t r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ double xrec_value(const xrec_t& r) { return __hiloint2double((int)r.z, (int)r.y); }
__device__ __forceinline__ void xchg_loads_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... and every loaded value goes through this once, after xchg_loads_done(): it ties the value's uses behind the wait (the compiler
// sees no dependence between the load statement's output and the wait statement otherwise)
__device__ __forceinline__ void xchg_loaded(double& v) { asm volatile("" : "+v"(v)); }

}  // namespace loamx
