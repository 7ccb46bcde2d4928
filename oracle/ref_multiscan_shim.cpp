// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over the reference's own MultiScanRegistration (the sweep ingestion:
// vertical-angle ring binning, start / end orientation, the half-sweep logic, relative times, IMU projection, then the feature
// extraction of BasicScanRegistration), compiled from src/lib/MultiScanRegistration.cpp and src/lib/BasicScanRegistration.cpp
// WHERE THEY LIE (oracle/Makefile target `ref`, output oracle/_ref/libref_multiscan.so).  ROS / PCL / Eigen are absent from
// this image: their headers resolve to the inert stand-ins in oracle/ref_stubs.  The four ROS-plumbing members of the
// ScanRegistration base (src/lib/ScanRegistration.cpp: parameter parsing, topic set-up, the IMU message handler, publishing)
// are defined empty HERE — they are not on the path; IMU states enter through BasicScanRegistration::updateIMUData.
#include <chrono>
#include <memory>
#include <vector>
#include <string>
#include <cmath>
#include <ros/ros.h>
#include <sensor_msgs/Imu.h>
#include <pcl_conversions/pcl_conversions.h>
#include <pcl/filters/voxel_grid.h>
// process() is private and the registration state sits in a protected base: the shim reads them; access control does not change layout
#define private public
#define protected public
#include "loam_velodyne/MultiScanRegistration.h"
#undef private
#undef protected

namespace loam {
bool ScanRegistration::parseParams(const ros::NodeHandle&, RegistrationParams&) { return true; }
bool ScanRegistration::setupROS(ros::NodeHandle&, ros::NodeHandle&, RegistrationParams&) { return true; }
void ScanRegistration::handleIMUMessage(const sensor_msgs::Imu::ConstPtr&) {}
void ScanRegistration::publishResult() {}
}  // namespace loam

using namespace loam;

namespace {
Time t_of(double sec) { return Time(std::chrono::duration_cast<Time::duration>(std::chrono::duration<double>(sec))); }
}

extern "C" {

void* ref_ms_create(float lower_deg, float upper_deg, int n_rings, float scanPeriod, int imuHistorySize, int nFeatureRegions, int curvatureRegion,
                    int maxCornerSharp, int maxSurfaceFlat, float lessFlatFilterSize, float surfaceCurvatureThreshold) {
  auto* h = new MultiScanRegistration(MultiScanMapper(lower_deg, upper_deg, (uint16_t)n_rings));
  h->configure(RegistrationParams(scanPeriod, imuHistorySize, nFeatureRegions, curvatureRegion, maxCornerSharp, maxSurfaceFlat, lessFlatFilterSize,
                                  surfaceCurvatureThreshold));
  return h;
}
void ref_ms_destroy(void* h) { delete (MultiScanRegistration*)h; }
int ref_ms_ring_for_angle(float lower_deg, float upper_deg, int n_rings, float angle_rad) {
  MultiScanMapper m(lower_deg, upper_deg, (uint16_t)n_rings);
  return m.getRingForAngle(angle_rad);
}
void ref_ms_update_imu(void* h, double stamp, float roll, float pitch, float yaw, float ax, float ay, float az) {
  Vector3 acc(ax, ay, az);
  IMUState st;
  st.stamp = t_of(stamp);
  st.roll = roll; st.pitch = pitch; st.yaw = yaw;
  st.acceleration = acc;
  ((MultiScanRegistration*)h)->updateIMUData(acc, st);
}
// MultiScanRegistration::process(laserCloudIn, scanTime): raw = n x (x, y, z) in the sensor's axes, firing order
void ref_ms_process(void* h, const float* raw, int n, double scan_time) {
  pcl::PointCloud<pcl::PointXYZ> in;
  for (int i = 0; i < n; i++) {
    pcl::PointXYZ p;
    p.x = raw[3 * i]; p.y = raw[3 * i + 1]; p.z = raw[3 * i + 2];
    in.push_back(p);
  }
  ((MultiScanRegistration*)h)->process(in, t_of(scan_time));
}
// the message entry (handleCloudMessage, :143-156) with its start-up delay counter; returns 1 when the message was processed
int ref_ms_handle_message(void* h, const float* raw, int n, unsigned sec, unsigned nsec) {
  auto* m = (MultiScanRegistration*)h;
  auto msg = std::make_shared<sensor_msgs::PointCloud2>();
  msg->header.stamp.sec = sec; msg->header.stamp.nsec = nsec;
  msg->data.assign(raw, raw + 3 * (size_t)n);
  msg->floats_per_point = 3;
  const int before = m->_systemDelay;
  m->handleCloudMessage(msg);
  return before > 0 ? 0 : 1;
}
// which: 0 laserCloud, 1 sharp, 2 lessSharp, 3 flat, 4 lessFlat; returns the cloud's size
int ref_ms_get(void* h, int which, float* out, int cap) {
  auto* s = (MultiScanRegistration*)h;
  const pcl::PointCloud<pcl::PointXYZI>* c[5] = {&s->laserCloud(), &s->cornerPointsSharp(), &s->cornerPointsLessSharp(), &s->surfacePointsFlat(),
                                                  &s->surfacePointsLessFlat()};
  const int n = (int)c[which]->size();
  for (int i = 0; i < n && i < cap; i++) {
    const pcl::PointXYZI& p = (*c[which])[i];
    out[4 * i] = p.x; out[4 * i + 1] = p.y; out[4 * i + 2] = p.z; out[4 * i + 3] = p.intensity;
  }
  return n;
}
// per-ring sizes of the binned sweep (the _laserCloudScans of the last process())
void ref_ms_ring_sizes(void* h, int* sizes, int n_rings) {
  auto* s = (MultiScanRegistration*)h;
  for (int r = 0; r < n_rings && r < (int)s->_laserCloudScans.size(); r++) sizes[r] = (int)s->_laserCloudScans[r].size();
}
void ref_ms_imu_trans(void* h, float* out12) {
  const auto& t = ((MultiScanRegistration*)h)->imuTransform();
  for (int k = 0; k < 4; k++) { out12[3 * k] = t[k].x; out12[3 * k + 1] = t[k].y; out12[3 * k + 2] = t[k].z; }
}

}  // extern "C"
