// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT PCL and NOT Eigen: the smallest declarations that let the reference's own
// Vector3.h / Twist.h / math_utils.h / BasicTransformMaintenance.{h,cpp} compile where they lie (oracle/Makefile, target
// `ref`), so that the oracle's restatements of those files can be pinned against the reference code itself.  Nothing of
// PCL's or Eigen's behaviour is restated here — the compiled reference functions only move floats through these types
// (element access, construction, assignment).
#pragma once

namespace Eigen {
template <class Derived> struct MatrixBase {
  const Derived& derived() const { return static_cast<const Derived&>(*this); }
};
struct Vector4f : MatrixBase<Vector4f> {
  float v[4];
  Vector4f() : v{0.f, 0.f, 0.f, 0.f} {}
  Vector4f(float a, float b, float c, float d) : v{a, b, c, d} {}
  Vector4f(const Vector4f&) = default;
  Vector4f& operator=(const Vector4f&) = default;
  template <class D> Vector4f(const MatrixBase<D>& o) { for (int k = 0; k < 4; k++) v[k] = o.derived()(k); }
  template <class D> Vector4f& operator=(const MatrixBase<D>& o) { for (int k = 0; k < 4; k++) v[k] = o.derived()(k); return *this; }
  float operator()(int i) const { return v[i]; }
  float& operator()(int i) { return v[i]; }
};
// coefficient-wise arithmetic, evaluated eagerly: the same float operations, in the same order per coefficient, as the
// expression templates of the real library perform (build with -ffp-contract=off)
inline Vector4f operator+(const Vector4f& a, const Vector4f& b) { return Vector4f(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]); }
inline Vector4f operator-(const Vector4f& a, const Vector4f& b) { return Vector4f(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]); }
inline Vector4f operator*(const Vector4f& a, float s) { return Vector4f(a.v[0] * s, a.v[1] * s, a.v[2] * s, a.v[3] * s); }
inline Vector4f operator*(float s, const Vector4f& a) { return Vector4f(s * a.v[0], s * a.v[1], s * a.v[2], s * a.v[3]); }
inline Vector4f operator*(double s, const Vector4f& a) { return (float)s * a; }   // a literal scalar is converted to the vector's scalar type
inline Vector4f operator*(const Vector4f& a, double s) { return a * (float)s; }
}  // namespace Eigen

namespace pcl {
struct PointXYZ { float x, y, z; };
struct PointXYZI { float x, y, z, intensity; };
}  // namespace pcl
