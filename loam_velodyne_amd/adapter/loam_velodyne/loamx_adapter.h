// Header-only C++ adapter: the reference's ROS-free class surface on top of the libloamx C-ABI.
//
// A maintainer drops these three classes in place of the reference's
//   loam::BasicScanRegistration  (include/loam_velodyne/BasicScanRegistration.h:135-163)
//   loam::BasicLaserOdometry     (include/loam_velodyne/BasicLaserOdometry.h:16-48)
//   loam::BasicLaserMapping      (include/loam_velodyne/BasicLaserMapping.h:80-111)
// and the L2 ROS wrappers (ScanRegistration / LaserOdometry / LaserMapping) compile against them unchanged: same member
// names, same argument meaning, same return conventions.  With PCL present define LOAMX_USE_PCL before including this
// header and the clouds are real pcl::PointCloud<pcl::PointXYZI>; without it a layout-compatible stand-in is used
// (32-byte records: x,y,z,1 | intensity,pad[3] — the layout PCL serialises with point_step 32).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "loamx.h"

// LOAMX_REFERENCE_TYPES (implies LOAMX_USE_PCL): the build sits inside the reference's tree and keeps the reference's own
// value-type headers — Angle.h, Vector3.h, Twist.h, time_utils.h — so that the wrappers and math_utils.h see exactly the
// types they were written against; the adapter then adds the reference's time-stamped overloads (IMUState, IMUState2,
// processScanlines(Time, ...), process(Time)).  oracle/dropin_check.sh compiles the reference's wrapper sources this way.
#ifdef LOAMX_REFERENCE_TYPES
#ifndef LOAMX_USE_PCL
#define LOAMX_USE_PCL
#endif
#endif
#ifdef LOAMX_USE_PCL
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace loamx_pcl = pcl;
#else
namespace loamx_pcl {
struct alignas(16) PointXYZI {
  float x = 0.f, y = 0.f, z = 0.f, w = 1.f;
  float intensity = 0.f, pad[3] = {0.f, 0.f, 0.f};
};
struct alignas(16) PointXYZ {
  float x = 0.f, y = 0.f, z = 0.f, w = 1.f;
};
template <class P> struct PointCloud {
  std::vector<P> points;
  using Ptr = std::shared_ptr<PointCloud<P>>;
  size_t size() const { return points.size(); }
  void clear() { points.clear(); }
  void push_back(const P& p) { points.push_back(p); }
  P& operator[](size_t i) { return points[i]; }
  const P& operator[](size_t i) const { return points[i]; }
};
}  // namespace loamx_pcl
#endif

namespace loam {

using CloudXYZI = loamx_pcl::PointCloud<loamx_pcl::PointXYZI>;
static_assert(sizeof(loamx_pcl::PointXYZI) == 32, "PointXYZI must be the 32-byte PCL record");

namespace detail {
inline loamx_cloud in_cloud(const CloudXYZI& c) {
  return loamx_cloud{(void*)c.points.data(), (uint32_t)c.points.size(), 32u, 16u, 0u};
}
// call fn(loamx_cloud*) with a growing output buffer until it fits
template <class F> inline void out_cloud(CloudXYZI& c, size_t hint, F&& fn) {
  size_t cap = hint ? hint : 1024;
  for (;;) {
    c.points.resize(cap);
    loamx_cloud d{(void*)c.points.data(), (uint32_t)cap, 32u, 16u, 0u};
    int rc = fn(&d);
    if (rc == LOAMX_E_CAPACITY) { cap = d.count + 16; continue; }
    if (rc < 0) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
    c.points.resize(d.count);
    return;
  }
}
inline void check(int rc) {
  if (rc < 0) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
}
}  // namespace detail

}  // namespace loam
#ifdef LOAMX_REFERENCE_TYPES
#include "loam_velodyne/Angle.h"
#include "loam_velodyne/Vector3.h"
#include "loam_velodyne/Twist.h"
#include "loam_velodyne/time_utils.h"
namespace loam {
#else
namespace loam {
// Angle / Twist value types with the accessors the wrappers use (Angle.h:16-67, Twist.h:15-27)
struct Angle {
  float _rad = 0.f;
  Angle() = default;
  Angle(float r) : _rad(r) {}
  float rad() const { return _rad; }
  float deg() const { return float(_rad * 180 / 3.14159265358979323846); }
};
struct Vector3 {
  float v[3] = {0.f, 0.f, 0.f};
  Vector3() = default;
  Vector3(float x_, float y_, float z_) : v{x_, y_, z_} {}
  float x() const { return v[0]; }
  float y() const { return v[1]; }
  float z() const { return v[2]; }
};
struct Twist {
  Angle rot_x, rot_y, rot_z;
  Vector3 pos;
};
#endif

namespace detail {
inline Twist twist_from(const float* t) {
  Twist w;
  w.rot_x = t[0]; w.rot_y = t[1]; w.rot_z = t[2];
  w.pos = Vector3(t[3], t[4], t[5]);
  return w;
}
inline void twist_to(const Twist& w, float* t) {
  t[0] = w.rot_x.rad(); t[1] = w.rot_y.rad(); t[2] = w.rot_z.rad(); t[3] = w.pos.x(); t[4] = w.pos.y(); t[5] = w.pos.z();
}
}  // namespace detail

// RegistrationParams (BasicScanRegistration.h:34-72)
struct RegistrationParams {
  RegistrationParams() = default;
  RegistrationParams(const float& scanPeriod_, const int& imuHistorySize_ = 200, const int& nFeatureRegions_ = 6, const int& curvatureRegion_ = 5,
                     const int& maxCornerSharp_ = 2, const int& maxSurfaceFlat_ = 4, const float& lessFlatFilterSize_ = 0.2f,
                     const float& surfaceCurvatureThreshold_ = 0.1f)
      : scanPeriod(scanPeriod_), imuHistorySize(imuHistorySize_), nFeatureRegions(nFeatureRegions_), curvatureRegion(curvatureRegion_),
        maxCornerSharp(maxCornerSharp_), maxCornerLessSharp(10 * maxCornerSharp_), maxSurfaceFlat(maxSurfaceFlat_),
        lessFlatFilterSize(lessFlatFilterSize_), surfaceCurvatureThreshold(surfaceCurvatureThreshold_) {}
  float scanPeriod = 0.1f;
  int imuHistorySize = 200;
  int nFeatureRegions = 6;
  int curvatureRegion = 5;
  int maxCornerSharp = 2;
  int maxCornerLessSharp = 20;
  int maxSurfaceFlat = 4;
  float lessFlatFilterSize = 0.2f;
  float surfaceCurvatureThreshold = 0.1f;
};

#ifdef LOAMX_REFERENCE_TYPES
// IMUState (BasicScanRegistration.h:76-132): the record the wrapper's IMU handler fills.  The history, its interpolation and the
// integration of position / velocity live behind loamx_scanreg_update_imu.
struct IMUState {
  Time stamp;
  Angle roll, pitch, yaw;
  Vector3 position, velocity, acceleration;
};
// IMUState2 (BasicLaserMapping.h:47-76)
struct IMUState2 {
  Time stamp;
  Angle roll, pitch;
};
namespace detail {
inline double seconds(const Time& t) { return toSec(t.time_since_epoch()); }
}
#endif

class BasicScanRegistration {
 public:
  BasicScanRegistration() = default;
  ~BasicScanRegistration() { loamx_scanreg_destroy(_h); }
  BasicScanRegistration(const BasicScanRegistration&) = delete;              // the object owns a device handle
  BasicScanRegistration& operator=(const BasicScanRegistration&) = delete;
#ifdef LOAMX_REFERENCE_TYPES
  // the reference's own signatures (BasicScanRegistration.h:140-150): time points instead of seconds
  void processScanlines(const Time& scanTime, std::vector<CloudXYZI> const& laserCloudScans) {
    setScanTime(detail::seconds(scanTime));
    _sweepStart = scanTime;
    processScanlines<int>(0, laserCloudScans);
  }
  void updateIMUData(Vector3& acc, IMUState& newState) {
    updateIMUData(detail::seconds(newState.stamp), newState.roll.rad(), newState.pitch.rad(), newState.yaw.rad(), acc.x(), acc.y(), acc.z());
  }
  auto const& sweepStart() { return _sweepStart; }
  // MultiScanRegistration::process(laserCloudIn, scanTime) on the device, with the reference's time point
  template <class CloudXYZ> void processRawSweep(const Time& scanTime, CloudXYZ const& laserCloudIn, float lowerBoundDeg, float upperBoundDeg, uint16_t nScanRings) {
    setScanTime(detail::seconds(scanTime));
    _sweepStart = scanTime;
    processRawSweep<int, CloudXYZ>(0, laserCloudIn, lowerBoundDeg, upperBoundDeg, nScanRings);
  }
#endif
  // configure (BasicScanRegistration.cpp:49-53): every RegistrationParams member is forwarded — maxCornerLessSharp and
  // imuHistorySize too — and an existing handle keeps its IMU history and sweep state, as the reference's object does
  bool configure(const RegistrationParams& config = RegistrationParams()) {
    _config = config;
    loamx_scanreg_config c;
    loamx_scanreg_default_config(&c);
    c.scan_period = config.scanPeriod;
    c.n_feature_regions = config.nFeatureRegions;
    c.curvature_region = config.curvatureRegion;
    c.max_corner_sharp = config.maxCornerSharp;
    c.max_corner_less_sharp = config.maxCornerLessSharp;
    c.max_surface_flat = config.maxSurfaceFlat;
    c.less_flat_filter_size = config.lessFlatFilterSize;
    c.surface_curvature_threshold = config.surfaceCurvatureThreshold;
    c.imu_history_size = config.imuHistorySize;
    if (_h) return loamx_scanreg_configure(_h, &c) == LOAMX_OK;
    _h = loamx_scanreg_create(&c);
    return _h != nullptr;
  }
  // processScanlines(scanTime, laserCloudScans) — BasicScanRegistration.cpp:28-46
  template <class TimeT> void processScanlines(const TimeT&, std::vector<CloudXYZI> const& laserCloudScans) {
    if (!_h && !configure(_config)) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
    _laserCloud.clear();
    std::vector<uint32_t> ring(laserCloudScans.size());
    for (size_t i = 0; i < laserCloudScans.size(); i++) {
      ring[i] = (uint32_t)laserCloudScans[i].size();
      _laserCloud.points.insert(_laserCloud.points.end(), laserCloudScans[i].points.begin(), laserCloudScans[i].points.end());
    }
    const size_t n = _laserCloud.size() + 16;
    _cornerPointsSharp.points.resize(n); _cornerPointsLessSharp.points.resize(n);
    _surfacePointsFlat.points.resize(n); _surfacePointsLessFlat.points.resize(n);
    loamx_cloud in = detail::in_cloud(_laserCloud);
    loamx_cloud o[4] = {{_cornerPointsSharp.points.data(), (uint32_t)n, 32, 16, 0}, {_cornerPointsLessSharp.points.data(), (uint32_t)n, 32, 16, 0},
                        {_surfacePointsFlat.points.data(), (uint32_t)n, 32, 16, 0}, {_surfacePointsLessFlat.points.data(), (uint32_t)n, 32, 16, 0}};
    detail::check(loamx_scanreg_process(_h, &in, ring.data(), (uint32_t)ring.size(), &o[0], &o[1], &o[2], &o[3]));
    _cornerPointsSharp.points.resize(o[0].count); _cornerPointsLessSharp.points.resize(o[1].count);
    _surfacePointsFlat.points.resize(o[2].count); _surfacePointsLessFlat.points.resize(o[3].count);
  }
  // The body of MultiScanRegistration::process(laserCloudIn, scanTime) — src/lib/MultiScanRegistration.cpp:160-238 — plus the
  // processScanlines it ends with, in one call: laserCloudIn is the raw /velodyne_points payload (pcl::PointXYZ records in
  // sensor axes, firing order); ring binning, relTime and feature extraction run on the device.
  template <class TimeT, class CloudXYZ>
  void processRawSweep(const TimeT&, CloudXYZ const& laserCloudIn, float lowerBoundDeg, float upperBoundDeg, uint16_t nScanRings) {
    if (!_h && !configure(_config)) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
    const size_t n = laserCloudIn.points.size() + 16;
    _laserCloud.points.resize(n);
    _cornerPointsSharp.points.resize(n); _cornerPointsLessSharp.points.resize(n);
    _surfacePointsFlat.points.resize(n); _surfacePointsLessFlat.points.resize(n);
    loamx_cloud full = {_laserCloud.points.data(), (uint32_t)n, 32, 16, 0};
    loamx_cloud o[4] = {{_cornerPointsSharp.points.data(), (uint32_t)n, 32, 16, 0}, {_cornerPointsLessSharp.points.data(), (uint32_t)n, 32, 16, 0},
                        {_surfacePointsFlat.points.data(), (uint32_t)n, 32, 16, 0}, {_surfacePointsLessFlat.points.data(), (uint32_t)n, 32, 16, 0}};
    const loamx_multiscan_mapper m = {lowerBoundDeg, upperBoundDeg, nScanRings};
    _ringSizes.assign(nScanRings, 0);
    detail::check(loamx_scanreg_process_raw(_h, &m, laserCloudIn.points.data(), (uint32_t)laserCloudIn.points.size(),
                                            (uint32_t)sizeof(laserCloudIn.points[0]), &full, _ringSizes.data(), &o[0], &o[1], &o[2], &o[3]));
    _laserCloud.points.resize(full.count);
    _cornerPointsSharp.points.resize(o[0].count); _cornerPointsLessSharp.points.resize(o[1].count);
    _surfacePointsFlat.points.resize(o[2].count); _surfacePointsLessFlat.points.resize(o[3].count);
  }
  auto const& ringSizes() { return _ringSizes; }      // points per scan ring of the last raw sweep
  // updateIMUData(acc, newState) — BasicScanRegistration.cpp:82-98 — with the state spelled out: stamp in seconds on the
  // clock of the scan times, roll / pitch / yaw, local acceleration (gravity removed, axes remapped: ScanRegistration.cpp:171-174)
  void updateIMUData(double stampSec, float roll, float pitch, float yaw, float accX, float accY, float accZ) {
    if (!_h && !configure(_config)) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
    const float acc[3] = {accX, accY, accZ};
    detail::check(loamx_scanreg_update_imu(_h, stampSec, roll, pitch, yaw, acc));
  }
  // the scanTime of the next processScanlines / processRawSweep call, in seconds (only the IMU path looks at it)
  void setScanTime(double scanTimeSec) {
    if (!_h && !configure(_config)) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
    detail::check(loamx_scanreg_set_time(_h, scanTimeSec));
  }
  auto const& imuTransform() {   // updateIMUTransform (:258-281), a cloud of 4 points; zeros without IMU data
    float t[12] = {0};
    if (_h) detail::check(loamx_scanreg_get_imu_trans(_h, t));
    _imuTrans.points.resize(4);
    for (int k = 0; k < 4; k++) { _imuTrans.points[k].x = t[3 * k]; _imuTrans.points[k].y = t[3 * k + 1]; _imuTrans.points[k].z = t[3 * k + 2]; }
    return _imuTrans;
  }
  auto const& laserCloud() { return _laserCloud; }
  auto const& cornerPointsSharp() { return _cornerPointsSharp; }
  auto const& cornerPointsLessSharp() { return _cornerPointsLessSharp; }
  auto const& surfacePointsFlat() { return _surfacePointsFlat; }
  auto const& surfacePointsLessFlat() { return _surfacePointsLessFlat; }
  auto const& config() { return _config; }

 private:
  loamx_scanreg* _h = nullptr;
  RegistrationParams _config;
  CloudXYZI _laserCloud, _cornerPointsSharp, _cornerPointsLessSharp, _surfacePointsFlat, _surfacePointsLessFlat;
  loamx_pcl::PointCloud<loamx_pcl::PointXYZ> _imuTrans;
#ifdef LOAMX_REFERENCE_TYPES
  Time _sweepStart;
#endif
  std::vector<uint32_t> _ringSizes;
};

class BasicLaserOdometry {
 public:
  explicit BasicLaserOdometry(float scanPeriod = 0.1f, size_t maxIterations = 25)
      : _cornerPointsSharp(new CloudXYZI), _cornerPointsLessSharp(new CloudXYZI), _surfPointsFlat(new CloudXYZI),
        _surfPointsLessFlat(new CloudXYZI), _laserCloud(new CloudXYZI), _lastCornerCloud(new CloudXYZI), _lastSurfaceCloud(new CloudXYZI) {
    loamx_odom_default_config(&_cfg);
    _cfg.scan_period = scanPeriod;
    _cfg.max_iterations = (int)maxIterations;
  }
  ~BasicLaserOdometry() { loamx_odom_destroy(_h); }
  BasicLaserOdometry(const BasicLaserOdometry&) = delete;                    // the object owns a device handle
  BasicLaserOdometry& operator=(const BasicLaserOdometry&) = delete;
  void setScanPeriod(float v) { _cfg.scan_period = v; reset_handle(); }
  void setMaxIterations(size_t v) { _cfg.max_iterations = (int)v; reset_handle(); }
  void setDeltaTAbort(float v) { _cfg.delta_t_abort = v; reset_handle(); }
  void setDeltaRAbort(float v) { _cfg.delta_r_abort = v; reset_handle(); }
  auto& cornerPointsSharp() { return _cornerPointsSharp; }
  auto& cornerPointsLessSharp() { return _cornerPointsLessSharp; }
  auto& surfPointsFlat() { return _surfPointsFlat; }
  auto& surfPointsLessFlat() { return _surfPointsLessFlat; }
  auto& laserCloud() { return _laserCloud; }
  auto frameCount() const { return _frameCount; }
  // updateIMU(imuTrans) — BasicLaserOdometry.cpp:181-194: 4 points = start angles, end angles, shift, velocity
  template <class CloudXYZ> void updateIMU(CloudXYZ const& imuTrans) {
    float t[12];
    for (int k = 0; k < 4; k++) { t[3 * k] = imuTrans.points[k].x; t[3 * k + 1] = imuTrans.points[k].y; t[3 * k + 2] = imuTrans.points[k].z; }
    ensure();
    detail::check(loamx_odom_update_imu(_h, t));
  }
  void process() {   // BasicLaserOdometry.cpp:196-666
    ensure();
    loamx_cloud a = detail::in_cloud(*_cornerPointsSharp), b = detail::in_cloud(*_cornerPointsLessSharp),
                c = detail::in_cloud(*_surfPointsFlat), d = detail::in_cloud(*_surfPointsLessFlat);
    int rc = loamx_odom_process(_h, &a, &b, &c, &d);
    detail::check(rc);
    if (rc == LOAMX_OK) _frameCount++;
    float t[6];
    detail::check(loamx_odom_get_transform(_h, t)); _transform = detail::twist_from(t);
    detail::check(loamx_odom_get_transform_sum(_h, t)); _transformSum = detail::twist_from(t);
    detail::out_cloud(*_lastCornerCloud, _cornerPointsLessSharp->size() + 16, [&](loamx_cloud* o) { return loamx_odom_get_last_clouds(_h, o, nullptr); });
    detail::out_cloud(*_lastSurfaceCloud, _surfPointsLessFlat->size() + 16, [&](loamx_cloud* o) { return loamx_odom_get_last_clouds(_h, nullptr, o); });
  }
  auto const& transformSum() { return _transformSum; }
  auto const& transform() { return _transform; }
  auto const& lastCornerCloud() { return _lastCornerCloud; }
  auto const& lastSurfaceCloud() { return _lastSurfaceCloud; }
  size_t transformToEnd(CloudXYZI::Ptr& cloud) {   // BasicLaserOdometry.cpp:57-87
    ensure();
    loamx_cloud c = detail::in_cloud(*cloud);
    detail::check(loamx_odom_transform_to_end(_h, &c));
    return cloud->size();
  }

 private:
  loamx_odom* _h = nullptr;
  loamx_odom_config _cfg;
  long _frameCount = 0;
  Twist _transform, _transformSum;
  CloudXYZI::Ptr _cornerPointsSharp, _cornerPointsLessSharp, _surfPointsFlat, _surfPointsLessFlat, _laserCloud, _lastCornerCloud, _lastSurfaceCloud;
  void reset_handle() { loamx_odom_destroy(_h); _h = nullptr; }
  void ensure() {
    if (!_h) _h = loamx_odom_create(&_cfg);
    if (!_h) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
  }
};

class BasicLaserMapping {
 public:
  // leaf-size proxy with the setLeafSize() the wrappers call (LaserMapping.cpp:112-138)
  struct FilterProxy {
    float* leaf;
    BasicLaserMapping* owner;
    void setLeafSize(float x, float, float) { *leaf = x; owner->reset_handle(); }
  };
  explicit BasicLaserMapping(const float& scanPeriod = 0.1f, const size_t& maxIterations = 10) {
    loamx_map_default_config(&_cfg);
    _cfg.scan_period = scanPeriod;
    _cfg.max_iterations = (int)maxIterations;
  }
  ~BasicLaserMapping() { loamx_map_destroy(_h); }
  BasicLaserMapping(const BasicLaserMapping&) = delete;                      // the object owns a device handle
  BasicLaserMapping& operator=(const BasicLaserMapping&) = delete;
#ifdef LOAMX_REFERENCE_TYPES
  // the reference's own signatures (BasicLaserMapping.h:85-86)
  void updateIMU(IMUState2 const& newState) { updateIMU(detail::seconds(newState.stamp), newState.roll.rad(), newState.pitch.rad()); }
  bool process(Time const& laserOdometryTime) { return processAt(detail::seconds(laserOdometryTime)); }
#endif
  // updateIMU(IMUState2) (:602-605): stamp in seconds on the clock of the process() times
  void updateIMU(double stampSec, float roll, float pitch) { ensure(); detail::check(loamx_map_update_imu(_h, stampSec, roll, pitch)); }
  // process(laserOdometryTime) with the time as seconds (needed by the IMU blend of transformUpdate only)
  bool processAt(double laserOdometryTimeSec) {
    ensure();
    detail::check(loamx_map_set_time(_h, laserOdometryTimeSec));
    return process<int>(0);
  }
  template <class TimeT> bool process(TimeT const&) {   // BasicLaserMapping.cpp:266-599
    ensure();
    loamx_cloud a = detail::in_cloud(_laserCloudCornerLast), b = detail::in_cloud(_laserCloudSurfLast), f = detail::in_cloud(_laserCloudFullRes);
    int rc = loamx_map_process(_h, &a, &b, &f);
    detail::check(rc);
    float t[6];
    detail::check(loamx_map_get_transform(_h, 0, t)); _transformAftMapped = detail::twist_from(t);
    detail::check(loamx_map_get_transform(_h, 1, t)); _transformBefMapped = detail::twist_from(t);
    if (loamx_map_has_fresh_map(_h))
      detail::out_cloud(_laserCloudSurroundDS, 1 << 16, [&](loamx_cloud* o) { return loamx_map_get_surround(_h, o); });
    return rc == LOAMX_OK;
  }
  void updateOdometry(double pitch, double yaw, double roll, double x, double y, double z) {   // :607-616
    const float t[6] = {(float)pitch, (float)yaw, (float)roll, (float)x, (float)y, (float)z};
    ensure();
    detail::check(loamx_map_update_odometry(_h, t));
  }
  void updateOdometry(Twist const& twist) {
    float t[6];
    detail::twist_to(twist, t);
    ensure();
    detail::check(loamx_map_update_odometry(_h, t));
  }
  auto& laserCloud() { return _laserCloudFullRes; }
  auto& laserCloudCornerLast() { return _laserCloudCornerLast; }
  auto& laserCloudSurfLast() { return _laserCloudSurfLast; }
  void setScanPeriod(float v) { _cfg.scan_period = v; reset_handle(); }
  void setMaxIterations(size_t v) { _cfg.max_iterations = (int)v; reset_handle(); }
  void setDeltaTAbort(float v) { _cfg.delta_t_abort = v; reset_handle(); }
  void setDeltaRAbort(float v) { _cfg.delta_r_abort = v; reset_handle(); }
  FilterProxy downSizeFilterCorner() { return {&_cfg.corner_filter_size, this}; }
  FilterProxy downSizeFilterSurf() { return {&_cfg.surf_filter_size, this}; }
  FilterProxy downSizeFilterMap() { return {&_cfg.map_filter_size, this}; }
  auto scanPeriod() const { return _cfg.scan_period; }
  auto maxIterations() const { return (size_t)_cfg.max_iterations; }
  auto deltaTAbort() const { return _cfg.delta_t_abort; }
  auto deltaRAbort() const { return _cfg.delta_r_abort; }
  auto const& transformAftMapped() const { return _transformAftMapped; }
  auto const& transformBefMapped() const { return _transformBefMapped; }
  auto const& laserCloudSurroundDS() const { return _laserCloudSurroundDS; }
  bool hasFreshMap() const { return _h && loamx_map_has_fresh_map(_h); }

 private:
  friend struct FilterProxy;
  loamx_map* _h = nullptr;
  loamx_map_config _cfg;
  Twist _transformAftMapped, _transformBefMapped;
  CloudXYZI _laserCloudCornerLast, _laserCloudSurfLast, _laserCloudFullRes, _laserCloudSurroundDS;
  // parameters may only change before the first process(): the live map lives in the handle
  void reset_handle() {
    if (_h) throw std::runtime_error("loamx: mapping parameters must be set before the first process()");
  }
  void ensure() {
    if (!_h) _h = loamx_map_create(&_cfg);
    if (!_h) throw std::runtime_error(std::string("loamx: ") + loamx_last_error());
  }
};

// BasicTransformMaintenance (include/loam_velodyne/BasicTransformMaintenance.h:44-66) over loamx_tm_*
class BasicTransformMaintenance {
 public:
  BasicTransformMaintenance() : _h(loamx_tm_create()) {}
  ~BasicTransformMaintenance() { loamx_tm_destroy(_h); }
  BasicTransformMaintenance(const BasicTransformMaintenance&) = delete;
  BasicTransformMaintenance& operator=(const BasicTransformMaintenance&) = delete;
  void updateOdometry(double pitch, double yaw, double roll, double x, double y, double z) {
    const float t[6] = {(float)pitch, (float)yaw, (float)roll, (float)x, (float)y, (float)z};
    detail::check(loamx_tm_update_odometry(_h, t));
  }
  void updateMappingTransform(double pitch, double yaw, double roll, double x, double y, double z, double twist_rot_x, double twist_rot_y,
                              double twist_rot_z, double twist_pos_x, double twist_pos_y, double twist_pos_z) {
    const float a[6] = {(float)pitch, (float)yaw, (float)roll, (float)x, (float)y, (float)z};
    const float b[6] = {(float)twist_rot_x, (float)twist_rot_y, (float)twist_rot_z, (float)twist_pos_x, (float)twist_pos_y, (float)twist_pos_z};
    detail::check(loamx_tm_update_mapping_transform(_h, a, b));
  }
  void updateMappingTransform(Twist const& transformAftMapped, Twist const& transformBefMapped) {
    float a[6], b[6];
    detail::twist_to(transformAftMapped, a);
    detail::twist_to(transformBefMapped, b);
    detail::check(loamx_tm_update_mapping_transform(_h, a, b));
  }
  void transformAssociateToMap() {
    detail::check(loamx_tm_associate_to_map(_h));
    detail::check(loamx_tm_get_mapped(_h, _transformMapped));
  }
  auto const& transformMapped() const { return _transformMapped; }

 private:
  loamx_tm* _h;
  float _transformMapped[6]{};
};

}  // namespace loam
