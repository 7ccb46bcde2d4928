// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over the reference's own BasicLaserOdometry, compiled from
// src/lib/BasicLaserOdometry.cpp WHERE IT LIES together with the reference's vendored nanoflann (oracle/Makefile target `ref`,
// output oracle/_ref/libref_odometry.so).  PCL / Eigen / boost are absent from this image: <pcl/...>, <Eigen/...> and
// <boost/shared_ptr.hpp> resolve to oracle/ref_stubs.  What is the reference's code running unchanged: the whole translation
// unit (correspondence search over the real nanoflann kd-trees, residual coefficients, Jacobian rows, pose update and abort
// test, IMU plug-in, transformToStart / transformToEnd).  What is NOT: the matrix product, colPivHouseholderQr, the
// self-adjoint eigen solver and the 6x6 inverse, which the Eigen stand-in forwards to the oracle's restatements.
#include "loam_velodyne/BasicLaserOdometry.h"

using namespace loam;

namespace {
void to_twist(Twist& t, const float* v) {
  t.rot_x = v[0]; t.rot_y = v[1]; t.rot_z = v[2];
  t.pos = Vector3(v[3], v[4], v[5]);
}
void from_twist(const Twist& t, float* v) {
  v[0] = t.rot_x.rad(); v[1] = t.rot_y.rad(); v[2] = t.rot_z.rad();
  v[3] = t.pos.x(); v[4] = t.pos.y(); v[5] = t.pos.z();
}
void fill(pcl::PointCloud<pcl::PointXYZI>& c, const float* pts, int n) {
  c.clear();
  for (int i = 0; i < n; i++) {
    pcl::PointXYZI p;
    p.x = pts[4 * i]; p.y = pts[4 * i + 1]; p.z = pts[4 * i + 2]; p.intensity = pts[4 * i + 3];
    c.push_back(p);
  }
}
}  // namespace

extern "C" {

void* ref_odom_create(float scanPeriod, int maxIterations, float deltaTAbort, float deltaRAbort) {
  auto* o = new BasicLaserOdometry(scanPeriod, (size_t)maxIterations);
  o->setDeltaTAbort(deltaTAbort);
  o->setDeltaRAbort(deltaRAbort);
  return o;
}
void ref_odom_destroy(void* h) { delete (BasicLaserOdometry*)h; }
// which: 0 laserCloud, 1 sharp, 2 lessSharp, 3 flat, 4 lessFlat
void ref_odom_set_cloud(void* h, int which, const float* pts, int n) {
  auto* o = (BasicLaserOdometry*)h;
  pcl::PointCloud<pcl::PointXYZI>::Ptr* c[5] = {&o->laserCloud(), &o->cornerPointsSharp(), &o->cornerPointsLessSharp(), &o->surfPointsFlat(),
                                                 &o->surfPointsLessFlat()};
  fill(**c[which], pts, n);
}
void ref_odom_update_imu(void* h, const float* t12) {
  pcl::PointCloud<pcl::PointXYZ> t;
  for (int k = 0; k < 4; k++) {
    pcl::PointXYZ p;
    p.x = t12[3 * k]; p.y = t12[3 * k + 1]; p.z = t12[3 * k + 2];
    t.push_back(p);
  }
  ((BasicLaserOdometry*)h)->updateIMU(t);
}
// tests seed the motion estimate (the member is private; the accessor hands out a reference to it)
void ref_odom_set_transform(void* h, const float* t6) { to_twist(const_cast<Twist&>(((BasicLaserOdometry*)h)->transform()), t6); }
void ref_odom_set_transform_sum(void* h, const float* t6) { to_twist(const_cast<Twist&>(((BasicLaserOdometry*)h)->transformSum()), t6); }
void ref_odom_process(void* h) { ((BasicLaserOdometry*)h)->process(); }
void ref_odom_get_transform(void* h, float* t6) { from_twist(((BasicLaserOdometry*)h)->transform(), t6); }
void ref_odom_get_transform_sum(void* h, float* t6) { from_twist(((BasicLaserOdometry*)h)->transformSum(), t6); }
void ref_odom_transform_full_to_end(void* h) { auto* o = (BasicLaserOdometry*)h; o->transformToEnd(o->laserCloud()); }
// which: 0 lastCornerCloud, 1 lastSurfaceCloud, 2 laserCloud
int ref_odom_get_cloud(void* h, int which, float* out, int cap) {
  auto* o = (BasicLaserOdometry*)h;
  const pcl::PointCloud<pcl::PointXYZI>& c = which == 0 ? *o->lastCornerCloud() : which == 1 ? *o->lastSurfaceCloud() : *o->laserCloud();
  const int n = (int)c.size();
  for (int i = 0; i < n && i < cap; i++) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].intensity; }
  return n;
}
long ref_odom_frame_count(void* h) { return ((BasicLaserOdometry*)h)->frameCount(); }

}  // extern "C"
