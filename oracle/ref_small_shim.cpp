// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over the parts of the reference that compile without ROS / PCL / Eigen
// (oracle/Makefile target `ref`, output oracle/_ref/libref_loam_small.so):
//   src/lib/BasicTransformMaintenance.cpp      (compiled as is, next to this file)
//   src/lib/math_utils.h, include/loam_velodyne/{Angle,Vector3,Twist,CircularBuffer}.h   (inline / header-only)
// <pcl/point_types.h> resolves to oracle/ref_stubs/pcl/point_types.h (type declarations only, see there).
#include "loam_velodyne/BasicTransformMaintenance.h"
#include "loam_velodyne/CircularBuffer.h"
#include "math_utils.h"

extern "C" {

// BasicTransformMaintenance: updateOdometry / updateMappingTransform / transformAssociateToMap -> transformMapped()
void ref_tm_associate(const float* sum6, const float* bef6, const float* aft6, float* mapped6) {
  loam::BasicTransformMaintenance tm;
  tm.updateOdometry(sum6[0], sum6[1], sum6[2], sum6[3], sum6[4], sum6[5]);
  tm.updateMappingTransform(aft6[0], aft6[1], aft6[2], aft6[3], aft6[4], aft6[5], bef6[0], bef6[1], bef6[2], bef6[3], bef6[4], bef6[5]);
  tm.transformAssociateToMap();
  for (int k = 0; k < 6; k++) mapped6[k] = tm.transformMapped()[k];
}

// Angle: value, cached cos / sin of a (possibly negated, possibly incremented) angle
void ref_angle(float rad, int negate, float add, float* out3) {
  loam::Angle a(rad);
  if (add != 0.f) a += add;
  const loam::Angle b = negate ? -a : a;
  out3[0] = b.rad(); out3[1] = b.cos(); out3[2] = b.sin();
}

// math_utils.h: rotateZXY / rotateYXZ on a Vector3, rotX / rotY / rotZ individually (which: 0 ZXY, 1 YXZ, 2 X, 3 Y, 4 Z)
void ref_rotate(int which, float* p3, float a0, float a1, float a2) {
  loam::Vector3 v(p3[0], p3[1], p3[2]);
  switch (which) {
    case 0: loam::rotateZXY(v, loam::Angle(a0), loam::Angle(a1), loam::Angle(a2)); break;
    case 1: loam::rotateYXZ(v, loam::Angle(a0), loam::Angle(a1), loam::Angle(a2)); break;
    case 2: loam::rotX(v, loam::Angle(a0)); break;
    case 3: loam::rotY(v, loam::Angle(a0)); break;
    default: loam::rotZ(v, loam::Angle(a0)); break;
  }
  p3[0] = v.x(); p3[1] = v.y(); p3[2] = v.z();
}
float ref_rad2deg(float r) { return loam::rad2deg(r); }
float ref_deg2rad(float d) { return loam::deg2rad(d); }

// CircularBuffer<int>: push a sequence into a buffer of the given capacity, read back size / first / last / [i]
int ref_circular(int capacity, const int* values, int n, int* out, int cap_out) {
  loam::CircularBuffer<int> b((size_t)capacity);
  for (int i = 0; i < n; i++) b.push(values[i]);
  const int sz = (int)b.size();
  for (int i = 0; i < sz && i < cap_out; i++) out[i] = b[(size_t)i];
  return sz;
}

}  // extern "C"
