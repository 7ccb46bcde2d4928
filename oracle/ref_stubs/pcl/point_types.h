// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT PCL: the two point structs the reference's sources name (x, y, z [, intensity], with
// the data[] alias nanoflann_pcl.h reads coordinates through) and the pcl_isfinite macro.  The float4 vector behind
// loam::Vector3 lives in the Eigen stand-in (Eigen/Core in this directory).
#pragma once
#include <cmath>
#include <Eigen/Core>

#define pcl_isfinite(x) std::isfinite(x)

namespace pcl {
// the records have PCL's own size and alignment (16 / 32 bytes, 16-byte aligned: x y z 1 | intensity pad pad pad)
struct alignas(16) PointXYZ {
  union { float data[4]; struct { float x, y, z; }; };
  PointXYZ() : data{0.f, 0.f, 0.f, 1.f} {}
};
struct alignas(16) PointXYZI {
  union { float data[4]; struct { float x, y, z; }; };
  union { struct { float intensity; }; float data_c[4]; };
  PointXYZI() : data{0.f, 0.f, 0.f, 1.f}, data_c{0.f, 0.f, 0.f, 0.f} {}
};
}  // namespace pcl
