// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the arithmetic the loam_velodyne registration hot path relies on.
// Nothing under oracle/ is part of the shipped product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may build, link, import or execute it, and only as the checker.
//
// PARITY PINNING (tests/test_ref_pinning.py, tests/test_oracle_primitives.py):
//   * The reference repository holds NO golden vectors / known-answer tests for any function on this path (its sole test
//     needs ROS + a downloaded bag).  The pin is the REFERENCE ITSELF RUN HERE: oracle/Makefile (target `ref`)
//     compiles the reference's own translation units where they lie — MultiScanRegistration.cpp, BasicScanRegistration.cpp,
//     BasicLaserOdometry.cpp, BasicLaserMapping.cpp, BasicTransformMaintenance.cpp, the vendored nanoflann.hpp — into
//     oracle/_ref/, and every restatement in this directory is compared with them BIT FOR BIT over multi-sweep runs; the
//     committed golden fixtures (tests/golden) are reproduced by the reference code as well.
//   * PCL, Eigen, boost and ROS are absent from this image, so those translation units compile against the stand-ins in
//     oracle/ref_stubs.  Containers, points, shared pointers and the ROS plumbing are inert.  FIVE third-party operations are
//     NOT inert and are forwarded to the restatements in this directory: pcl::VoxelGrid::filter (oracle_cloud.hpp voxel_grid),
//     Eigen's matrix product, colPivHouseholderQr().solve, SelfAdjointEigenSolver and inverse() (this header).  THOSE FIVE
//     REMAIN UNPINNED by the reference: they restate the libraries' published algorithms and are checked against NumPy
//     (tests/test_oracle_primitives.py) only.  Everything else on the path is pinned.
//
// This header: value types and small dense linear algebra.
//   Angle / Twist            -> include/loam_velodyne/Angle.h:16-67, Twist.h:15-27
//   rot*/rotateZXY/rotateYXZ -> src/lib/math_utils.h:129-275
//   sq_diff / pt_dist        -> src/lib/math_utils.h:68-121
//   eig_sym_jacobi           -> stands in for Eigen::SelfAdjointEigenSolver (not in /root/reference;
//                               Eigen3 is un-vendored, no version stated, CMakeLists.txt:14).  Contract restated:
//                               symmetric input read from the LOWER triangle, eigenvalues ascending,
//                               unit eigenvectors in columns, sign unspecified.
//   colpiv_qr_solve          -> stands in for Eigen::ColPivHouseholderQR::solve (Eigen 3.2/3.3 published
//                               algorithm: Householder reflections, column pivoting on the largest
//                               remaining column norm, solution restricted to the non-zero pivots).
//   inverse_lu               -> stands in for Eigen's general inverse of a 6x6 (partial-pivot LU).
#pragma once
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <limits>

namespace loam_oracle {

struct Pt {
  float x, y, z, i;
};
using Cloud = std::vector<Pt>;

// Angle.h:16-67 — float radian with cached float sin/cos; unary minus flips the sine only.
struct Angle {
  float r = 0.f, c = 1.f, s = 0.f;
  Angle() = default;
  Angle(float rad) : r(rad), c(std::cos(rad)), s(std::sin(rad)) {}
  Angle operator-() const {
    Angle o;
    o.r = -r;
    o.c = c;
    o.s = -s;
    return o;
  }
  void operator+=(float v) { *this = Angle(r + v); }
  void operator-=(float v) { *this = Angle(r - v); }
  float rad() const { return r; }
  float cos() const { return c; }
  float sin() const { return s; }
};

struct Vec3 {
  float x = 0.f, y = 0.f, z = 0.f;
};
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

// Twist.h:15-27
struct Twist {
  Angle rot_x, rot_y, rot_z;
  Vec3 pos;
};

// math_utils.h:129-201 (same operand order; the temporaries matter for float results)
template <class P> inline void rotX(P& p, const Angle& a) {
  float y = p.y;
  p.y = a.cos() * y - a.sin() * p.z;
  p.z = a.sin() * y + a.cos() * p.z;
}
template <class P> inline void rotY(P& p, const Angle& a) {
  float x = p.x;
  p.x = a.cos() * x + a.sin() * p.z;
  p.z = a.cos() * p.z - a.sin() * x;
}
template <class P> inline void rotZ(P& p, const Angle& a) {
  float x = p.x;
  p.x = a.cos() * x - a.sin() * p.y;
  p.y = a.sin() * x + a.cos() * p.y;
}
// math_utils.h:212-275
template <class P> inline void rotateZXY(P& p, const Angle& az, const Angle& ax, const Angle& ay) {
  rotZ(p, az);
  rotX(p, ax);
  rotY(p, ay);
}
template <class P> inline void rotateYXZ(P& p, const Angle& ay, const Angle& ax, const Angle& az) {
  rotY(p, ay);
  rotX(p, ax);
  rotZ(p, az);
}

// math_utils.h:68-95
template <class A, class B> inline float sq_diff(const A& a, const B& b) {
  float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;
}
template <class A, class B> inline float sq_diff_w(const A& a, const B& b, float wb) {
  float dx = a.x - b.x * wb, dy = a.y - b.y * wb, dz = a.z - b.z * wb;
  return dx * dx + dy * dy + dz * dz;
}
// math_utils.h:103-121
template <class P> inline float pt_dist(const P& p) { return std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z); }
template <class P> inline float sq_pt_dist(const P& p) { return p.x * p.x + p.y * p.y + p.z * p.z; }
// math_utils.h:30-33 — float -> double -> float
inline float rad2deg_f(float r) { return (float)(r * 180.0 / M_PI); }

// ---------------------------------------------------------------------------------------------
// Symmetric eigen-decomposition, cyclic Jacobi in float.  A is N x N row-major; only the lower
// triangle is read.  w ascending, V(:,k) = k-th unit eigenvector (row-major V[r*N+c]).
template <int N> inline void eig_sym_jacobi(const float* Ain, float* w, float* V) {
  float A[N][N];
  for (int r = 0; r < N; r++)
    for (int c = 0; c <= r; c++) A[r][c] = A[c][r] = Ain[r * N + c];
  float Q[N][N];
  for (int r = 0; r < N; r++)
    for (int c = 0; c < N; c++) Q[r][c] = (r == c) ? 1.f : 0.f;
  for (int sweep = 0; sweep < 16; sweep++) {
    float off = 0.f, diag = 0.f;
    for (int r = 0; r < N; r++) {
      diag += A[r][r] * A[r][r];
      for (int c = 0; c < r; c++) off += A[r][c] * A[r][c];
    }
    if (off <= 1e-20f * diag || off == 0.f) break;
    for (int p = 0; p < N - 1; p++)
      for (int q = p + 1; q < N; q++) {
        float apq = A[p][q];
        if (apq == 0.f) continue;
        float theta = (A[q][q] - A[p][p]) / (2.f * apq);
        float t = (theta >= 0.f ? 1.f : -1.f) / (std::fabs(theta) + std::sqrt(theta * theta + 1.f));
        float c = 1.f / std::sqrt(t * t + 1.f), s = t * c;
        for (int k = 0; k < N; k++) {  // A <- A J
          float akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; k++) {  // A <- J^T A
          float apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; k++) {
          float qkp = Q[k][p], qkq = Q[k][q];
          Q[k][p] = c * qkp - s * qkq;
          Q[k][q] = s * qkp + c * qkq;
        }
      }
  }
  int order[N];
  for (int k = 0; k < N; k++) order[k] = k;
  for (int a = 1; a < N; a++)  // insertion sort ascending, stable
    for (int b = a; b > 0 && A[order[b]][order[b]] < A[order[b - 1]][order[b - 1]]; b--) std::swap(order[b], order[b - 1]);
  for (int k = 0; k < N; k++) {
    w[k] = A[order[k]][order[k]];
    for (int r = 0; r < N; r++) V[r * N + k] = Q[r][order[k]];
  }
}

// Column-pivoted Householder QR least-squares solve of A(MxN, row-major) x = b.  M >= N.
template <int M, int N> inline void colpiv_qr_solve(const float* Ain, const float* bin, float* x) {
  float A[M][N], b[M];
  for (int r = 0; r < M; r++) {
    b[r] = bin[r];
    for (int c = 0; c < N; c++) A[r][c] = Ain[r * N + c];
  }
  int perm[N];
  for (int c = 0; c < N; c++) perm[c] = c;
  // (Eigen finds maxima with a visitor that is seeded with the FIRST coefficient and replaces it on `value > current` only — which
  // matters for nothing but a non-finite system: a NaN in front survives every comparison, the threshold becomes NaN, no pivot is
  // ever declared negligible and the solution comes out NaN, which is what BasicLaserOdometry.cpp:606-612 exists for)
  float maxnorm = 0.f;
  for (int c = 0; c < N; c++) {
    float s = 0.f;
    for (int r = 0; r < M; r++) s += A[r][c] * A[r][c];
    const float nrm = std::sqrt(s);
    if (c == 0 || nrm > maxnorm) maxnorm = nrm;
  }
  const float eps = std::numeric_limits<float>::epsilon();
  const float thr_helper = (maxnorm * eps) * (maxnorm * eps) / float(M);
  int nonzero = N;
  float maxpivot = 0.f;
  for (int k = 0; k < N; k++) {
    int best = k;
    float bestn = 0.f;
    for (int c = k; c < N; c++) {
      float s = 0.f;
      for (int r = k; r < M; r++) s += A[r][c] * A[r][c];
      if (c == k || s > bestn) {   // (seeded with the first remaining column, see above)
        bestn = s;
        best = c;
      }
    }
    if (nonzero == N && bestn < thr_helper * float(M - k)) nonzero = k;
    if (best != k) {
      for (int r = 0; r < M; r++) std::swap(A[r][k], A[r][best]);
      std::swap(perm[k], perm[best]);
    }
    // Householder on column k, rows k..M-1
    float c0 = A[k][k], tail = 0.f;
    for (int r = k + 1; r < M; r++) tail += A[r][k] * A[r][k];
    float tau, beta;
    float v[M];
    if (tail <= std::numeric_limits<float>::min()) {
      tau = 0.f;
      beta = c0;
      for (int r = k + 1; r < M; r++) v[r] = 0.f;
    } else {
      beta = std::sqrt(c0 * c0 + tail);
      if (c0 >= 0.f) beta = -beta;
      for (int r = k + 1; r < M; r++) v[r] = A[r][k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    v[k] = 1.f;
    A[k][k] = beta;
    for (int r = k + 1; r < M; r++) A[r][k] = 0.f;
    if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
    for (int c = k + 1; c < N; c++) {
      float dot = 0.f;
      for (int r = k; r < M; r++) dot += v[r] * A[r][c];
      dot *= tau;
      for (int r = k; r < M; r++) A[r][c] -= dot * v[r];
    }
    {
      float dot = 0.f;
      for (int r = k; r < M; r++) dot += v[r] * b[r];
      dot *= tau;
      for (int r = k; r < M; r++) b[r] -= dot * v[r];
    }
  }
  float y[N];
  for (int c = 0; c < N; c++) y[c] = 0.f;
  for (int k = nonzero - 1; k >= 0; k--) {
    float s = b[k];
    for (int c = k + 1; c < nonzero; c++) s -= A[k][c] * y[c];
    y[k] = s / A[k][k];
  }
  for (int c = 0; c < N; c++) x[c] = 0.f;
  for (int c = 0; c < nonzero; c++) x[perm[c]] = y[c];
}

// General inverse by partial-pivot Gauss-Jordan (float).  Returns false if singular.
template <int N> inline bool inverse_lu(const float* Ain, float* inv) {
  float A[N][2 * N];
  for (int r = 0; r < N; r++)
    for (int c = 0; c < N; c++) {
      A[r][c] = Ain[r * N + c];
      A[r][N + c] = (r == c) ? 1.f : 0.f;
    }
  for (int k = 0; k < N; k++) {
    int piv = k;
    for (int r = k + 1; r < N; r++)
      if (std::fabs(A[r][k]) > std::fabs(A[piv][k])) piv = r;
    if (A[piv][k] == 0.f) return false;
    if (piv != k)
      for (int c = 0; c < 2 * N; c++) std::swap(A[k][c], A[piv][c]);
    float d = 1.f / A[k][k];
    for (int c = 0; c < 2 * N; c++) A[k][c] *= d;
    for (int r = 0; r < N; r++)
      if (r != k) {
        float f = A[r][k];
        if (f != 0.f)
          for (int c = 0; c < 2 * N; c++) A[r][c] -= f * A[k][c];
      }
  }
  for (int r = 0; r < N; r++)
    for (int c = 0; c < N; c++) inv[r * N + c] = A[r][N + c];
  return true;
}

// Degeneracy projector shared by mapping and odometry (BasicLaserMapping.cpp:869-899,
// BasicLaserOdometry.cpp:561-591): eigen-decompose AtA, zero ROW i of the eigenvector matrix
// (eigenvectors are its columns) while the ascending eigenvalue i is below thr, P = V^-1 * V2.
inline bool degeneracy_projector(const float* AtA, float thr, float* P) {
  float w[6], V[36], V2[36], Vi[36];
  eig_sym_jacobi<6>(AtA, w, V);
  std::memcpy(V2, V, sizeof(V));
  bool degenerate = false;
  for (int i = 0; i < 6; i++) {
    if (w[i] < thr) {
      for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0.f;
      degenerate = true;
    } else
      break;
  }
  if (!inverse_lu<6>(V, Vi)) {
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

}  // namespace loam_oracle
