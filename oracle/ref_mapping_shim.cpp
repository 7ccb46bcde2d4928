// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over the reference's own BasicLaserMapping, compiled from
// src/lib/BasicLaserMapping.cpp WHERE IT LIES together with the reference's vendored nanoflann (oracle/Makefile target `ref`,
// output oracle/_ref/libref_mapping.so).  PCL / Eigen / boost are absent from this image: <pcl/...>, <Eigen/...> and
// <boost/shared_ptr.hpp> resolve to oracle/ref_stubs.  What is the reference's code running unchanged: the whole translation
// unit (pose prediction, stacking, the rolling cube window, sub-map selection by field of view, the 5-NN searches over the
// real nanoflann, edge / plane residuals, Jacobian rows, pose update and abort test, map insertion, transformUpdate with its
// IMU blend).  What is NOT: pcl::VoxelGrid (the oracle's voxel grid behind PCL's interface) and the Eigen operations (matrix
// product, colPivHouseholderQr, self-adjoint eigen solver, 6x6 inverse -> the oracle's restatements, ref_stubs/Eigen/Core).
#include <chrono>
#include <memory>
#include <vector>
#include <cmath>
#include <Eigen/Core>
#include <pcl/point_cloud.h>
#include <pcl/filters/voxel_grid.h>
#include <boost/shared_ptr.hpp>
// the shim reads the private cube arrays and poses to compare them with the oracle's; access control does not change the layout
#define private public
#include "loam_velodyne/BasicLaserMapping.h"
#undef private

using namespace loam;

namespace {
Time t_of(double sec) { return Time(std::chrono::duration_cast<Time::duration>(std::chrono::duration<double>(sec))); }
void to_twist(Twist& t, const float* v) {
  t.rot_x = v[0]; t.rot_y = v[1]; t.rot_z = v[2];
  t.pos = Vector3(v[3], v[4], v[5]);
}
void from_twist(const Twist& t, float* v) {
  v[0] = t.rot_x.rad(); v[1] = t.rot_y.rad(); v[2] = t.rot_z.rad();
  v[3] = t.pos.x(); v[4] = t.pos.y(); v[5] = t.pos.z();
}
int dump(const pcl::PointCloud<pcl::PointXYZI>& c, float* out, int cap, int at = 0) {
  const int n = (int)c.size();
  for (int i = 0; i < n && at + i < cap; i++) {
    float* o = out + 4 * (size_t)(at + i);
    o[0] = c[i].x; o[1] = c[i].y; o[2] = c[i].z; o[3] = c[i].intensity;
  }
  return n;
}
}  // namespace

extern "C" {

void* ref_map_create(float scanPeriod, int maxIterations, float deltaTAbort, float deltaRAbort, float cornerLeaf, float surfLeaf) {
  auto* m = new BasicLaserMapping(scanPeriod, (size_t)maxIterations);
  m->setDeltaTAbort(deltaTAbort);
  m->setDeltaRAbort(deltaRAbort);
  m->downSizeFilterCorner().setLeafSize(cornerLeaf, cornerLeaf, cornerLeaf);
  m->downSizeFilterSurf().setLeafSize(surfLeaf, surfLeaf, surfLeaf);
  return m;
}
void ref_map_destroy(void* h) { delete (BasicLaserMapping*)h; }
// which: 0 cornerLast, 1 surfLast, 2 fullRes
void ref_map_set_cloud(void* h, int which, const float* pts, int n) {
  auto* m = (BasicLaserMapping*)h;
  pcl::PointCloud<pcl::PointXYZI>& c = which == 0 ? m->laserCloudCornerLast() : which == 1 ? m->laserCloudSurfLast() : m->laserCloud();
  c.clear();
  for (int i = 0; i < n; i++) {
    pcl::PointXYZI p;
    p.x = pts[4 * i]; p.y = pts[4 * i + 1]; p.z = pts[4 * i + 2]; p.intensity = pts[4 * i + 3];
    c.push_back(p);
  }
}
void ref_map_update_odometry(void* h, const float* t6) {
  Twist t;
  to_twist(t, t6);
  ((BasicLaserMapping*)h)->updateOdometry(t);
}
int ref_map_process(void* h, double t) { return ((BasicLaserMapping*)h)->process(t_of(t)) ? 1 : 0; }
void ref_map_update_imu(void* h, double stamp, float roll, float pitch) {
  IMUState2 s;
  s.stamp = t_of(stamp);
  s.roll = roll;
  s.pitch = pitch;
  ((BasicLaserMapping*)h)->updateIMU(s);
}
// which: 0 aft, 1 bef, 2 tobe, 3 sum
void ref_map_get_transform(void* h, int which, float* t6) {
  auto* m = (BasicLaserMapping*)h;
  const Twist* t[4] = {&m->_transformAftMapped, &m->_transformBefMapped, &m->_transformTobeMapped, &m->_transformSum};
  from_twist(*t[which], t6);
}
void ref_map_set_transform(void* h, int which, const float* t6) {
  auto* m = (BasicLaserMapping*)h;
  Twist* t[4] = {&m->_transformAftMapped, &m->_transformBefMapped, &m->_transformTobeMapped, &m->_transformSum};
  to_twist(*t[which], t6);
}
// which: 0 fullRes(registered), 1 surroundDS, 2 cornerFromMap, 3 surfFromMap, 4 cornerStackDS, 5 surfStackDS, 6 all corner cubes, 7 all surf cubes
int ref_map_get_cloud(void* h, int which, float* out, int cap) {
  auto* m = (BasicLaserMapping*)h;
  if (which >= 6) {
    const auto& arr = which == 6 ? m->_laserCloudCornerArray : m->_laserCloudSurfArray;
    int at = 0;
    for (const auto& c : arr) at += dump(*c, out, cap, at);
    return at;
  }
  const pcl::PointCloud<pcl::PointXYZI>* c[6] = {m->_laserCloudFullRes.get(), m->_laserCloudSurroundDS.get(), m->_laserCloudCornerFromMap.get(),
                                                  m->_laserCloudSurfFromMap.get(), m->_laserCloudCornerStackDS.get(), m->_laserCloudSurfStackDS.get()};
  return dump(*c[which], out, cap);
}
// Registration of one sweep against a caller-provided sub-map — the unit the batched mode shards (oracle_mapping.hpp
// register_frozen).  Built ONLY from the reference's own members, in the order process() runs them (:282-292, :511-531):
// pointAssociateToMap into the stacks, pointAssociateTobeMapped back, the two down-sizing filters, optimizeTransformTobeMapped.
void ref_map_register_frozen(void* h, const float* corner_map, int ncm, const float* surf_map, int nsm, const float* guess6, float* pose6) {
  auto* m = (BasicLaserMapping*)h;
  auto load = [](pcl::PointCloud<pcl::PointXYZI>& c, const float* pts, int n) {
    c.clear();
    for (int i = 0; i < n; i++) {
      pcl::PointXYZI p;
      p.x = pts[4 * i]; p.y = pts[4 * i + 1]; p.z = pts[4 * i + 2]; p.intensity = pts[4 * i + 3];
      c.push_back(p);
    }
  };
  load(*m->_laserCloudCornerFromMap, corner_map, ncm);
  load(*m->_laserCloudSurfFromMap, surf_map, nsm);
  to_twist(m->_transformTobeMapped, guess6);
  pcl::PointXYZI pointSel;
  for (auto const& pt : m->_laserCloudCornerLast->points) { m->pointAssociateToMap(pt, pointSel); m->_laserCloudCornerStack->push_back(pointSel); }
  for (auto const& pt : m->_laserCloudSurfLast->points) { m->pointAssociateToMap(pt, pointSel); m->_laserCloudSurfStack->push_back(pointSel); }
  for (auto& pt : *m->_laserCloudCornerStack) m->pointAssociateTobeMapped(pt, pt);
  for (auto& pt : *m->_laserCloudSurfStack) m->pointAssociateTobeMapped(pt, pt);
  m->_laserCloudCornerStackDS->clear();
  m->_downSizeFilterCorner.setInputCloud(m->_laserCloudCornerStack);
  m->_downSizeFilterCorner.filter(*m->_laserCloudCornerStackDS);
  m->_laserCloudSurfStackDS->clear();
  m->_downSizeFilterSurf.setInputCloud(m->_laserCloudSurfStack);
  m->_downSizeFilterSurf.filter(*m->_laserCloudSurfStackDS);
  m->_laserCloudCornerStack->clear();
  m->_laserCloudSurfStack->clear();
  m->optimizeTransformTobeMapped();
  from_twist(m->_transformTobeMapped, pose6);
}
// transformAssociateToMap (:101-157) with the current sum / bef / aft; returns the predicted transformTobeMapped
void ref_map_associate(void* h, float* tobe6) {
  auto* m = (BasicLaserMapping*)h;
  m->transformAssociateToMap();
  from_twist(m->_transformTobeMapped, tobe6);
}
int ref_map_has_fresh_map(void* h) { return ((BasicLaserMapping*)h)->hasFreshMap() ? 1 : 0; }
void ref_map_grid_center(void* h, int* c3) {
  auto* m = (BasicLaserMapping*)h;
  c3[0] = m->_laserCloudCenWidth; c3[1] = m->_laserCloudCenHeight; c3[2] = m->_laserCloudCenDepth;
}

}  // extern "C"
