// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over the reference's own BasicScanRegistration (feature extraction and the
// IMU bookkeeping of the scan registration), compiled from src/lib/BasicScanRegistration.cpp WHERE IT LIES (oracle/Makefile
// target `ref`, output oracle/_ref/libref_scanreg.so).  PCL / Eigen are absent from this image: <pcl/...> resolves to
// oracle/ref_stubs (a std::vector container, coefficient-wise float4 arithmetic, and pcl::VoxelGrid's interface over the
// ORACLE's voxel grid — so agreement on the down-sampled less-flat cloud pins nothing, everything else in the translation unit
// is the reference's code running unchanged).
#include "loam_velodyne/BasicScanRegistration.h"
#include <chrono>

using namespace loam;

namespace {
Time t_of(double sec) { return Time(std::chrono::duration_cast<Time::duration>(std::chrono::duration<double>(sec))); }
}

extern "C" {

void* ref_sr_create(float scanPeriod, int imuHistorySize, int nFeatureRegions, int curvatureRegion, int maxCornerSharp, int maxSurfaceFlat,
                    float lessFlatFilterSize, float surfaceCurvatureThreshold) {
  auto* h = new BasicScanRegistration();
  h->configure(RegistrationParams(scanPeriod, imuHistorySize, nFeatureRegions, curvatureRegion, maxCornerSharp, maxSurfaceFlat,
                                  lessFlatFilterSize, surfaceCurvatureThreshold));
  return h;
}
void ref_sr_destroy(void* h) { delete (BasicScanRegistration*)h; }
// what ScanRegistration::parseParams does for the "maxCornerLessSharp" parameter (ScanRegistration.cpp:100-109): the member is set
// on its own and the object re-configured
void ref_sr_set_less_sharp(void* h, int maxCornerLessSharp) {
  auto* r = (BasicScanRegistration*)h;
  RegistrationParams p = r->config();
  p.maxCornerLessSharp = maxCornerLessSharp;
  r->configure(p);
}

// updateIMUData(acc, newState)
void ref_sr_update_imu(void* h, double stamp, float roll, float pitch, float yaw, float ax, float ay, float az) {
  Vector3 acc(ax, ay, az);
  IMUState st;
  st.stamp = t_of(stamp);
  st.roll = roll; st.pitch = pitch; st.yaw = yaw;
  st.acceleration = acc;
  ((BasicScanRegistration*)h)->updateIMUData(acc, st);
}
// projectPointToStartOfSweep(point, relTime) on (x, y, z, intensity)
void ref_sr_project(void* h, float* p4, float relTime) {
  pcl::PointXYZI p;
  p.x = p4[0]; p.y = p4[1]; p.z = p4[2]; p.intensity = p4[3];
  ((BasicScanRegistration*)h)->projectPointToStartOfSweep(p, relTime);
  p4[0] = p.x; p4[1] = p.y; p4[2] = p.z; p4[3] = p.intensity;
}
// processScanlines(scanTime, laserCloudScans): pts = rings concatenated (4 floats per point)
void ref_sr_process(void* h, double scan_time, const float* pts, const int* ring_sizes, int n_rings) {
  std::vector<pcl::PointCloud<pcl::PointXYZI>> scans(n_rings);
  size_t off = 0;
  for (int r = 0; r < n_rings; r++)
    for (int i = 0; i < ring_sizes[r]; i++, off++) {
      pcl::PointXYZI p;
      p.x = pts[4 * off]; p.y = pts[4 * off + 1]; p.z = pts[4 * off + 2]; p.intensity = pts[4 * off + 3];
      scans[r].push_back(p);
    }
  ((BasicScanRegistration*)h)->processScanlines(t_of(scan_time), scans);
}
// which: 0 laserCloud, 1 sharp, 2 lessSharp, 3 flat, 4 lessFlat; returns the cloud's size
int ref_sr_get(void* h, int which, float* out, int cap) {
  auto* s = (BasicScanRegistration*)h;
  const pcl::PointCloud<pcl::PointXYZI>* c[5] = {&s->laserCloud(), &s->cornerPointsSharp(), &s->cornerPointsLessSharp(), &s->surfacePointsFlat(),
                                                  &s->surfacePointsLessFlat()};
  const int n = (int)c[which]->size();
  for (int i = 0; i < n && i < cap; i++) {
    const pcl::PointXYZI& p = (*c[which])[i];
    out[4 * i] = p.x; out[4 * i + 1] = p.y; out[4 * i + 2] = p.z; out[4 * i + 3] = p.intensity;
  }
  return n;
}
void ref_sr_imu_trans(void* h, float* out12) {
  const auto& t = ((BasicScanRegistration*)h)->imuTransform();
  for (int k = 0; k < 4; k++) { out12[3 * k] = t[k].x; out12[3 * k + 1] = t[k].y; out12[3 * k + 2] = t[k].z; }
}

}  // extern "C"
