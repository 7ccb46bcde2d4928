// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp for the rules and for what is / is not pinned by the reference's own code).
//
// Per-sweep feature extraction, restating BasicScanRegistration (IMU-less path):
//   process_scanlines   -> src/lib/BasicScanRegistration.cpp:28-46
//   extract_features    -> :155-254
//   set_scan_buffers    -> :321-363   (occlusion / parallel-beam masks)
//   set_region_buffers  -> :284-318   (curvature + stable ascending order)
//   mark_as_picked      -> :367-386
//   RegistrationParams  -> include/loam_velodyne/BasicScanRegistration.h:34-72, .cpp:9-26
//   PointLabel          -> .h:24-30
// IMU state:
//   update_imu_data / project_point_to_start_of_sweep / set_imu_transform_for / transform_to_start_imu /
//   interpolate_imu_state_for -> :82-152;  reset -> :55-79;  update_imu_transform -> :258-281;
//   IMUState + interpolate -> include/loam_velodyne/BasicScanRegistration.h:77-132;  CircularBuffer.h:111-119 (push).
// Times are double seconds on any common clock (the reference's Time / toSec, time_utils.h).  NOTE the reference's
// order of events, reproduced as is: MultiScanRegistration::process projects the points of a sweep (:231) BEFORE
// processScanlines calls reset(scanTime) (:31), so a sweep is de-skewed with the scan time / start state the PREVIOUS
// sweep's reset left behind (a default-constructed Time and state for the very first sweep).
#pragma once
#include "oracle_cloud.hpp"
#include <deque>

namespace loam_oracle {

struct RegistrationParams {
  float scanPeriod = 0.1f;
  int imuHistorySize = 200;
  int nFeatureRegions = 6;
  int curvatureRegion = 5;
  int maxCornerSharp = 2;
  int maxCornerLessSharp = 20;   // 10 * maxCornerSharp (.cpp:22)
  int maxSurfaceFlat = 4;
  float lessFlatFilterSize = 0.2f;
  float surfaceCurvatureThreshold = 0.1f;
};

enum PointLabel { CORNER_SHARP = 2, CORNER_LESS_SHARP = 1, SURFACE_LESS_FLAT = 0, SURFACE_FLAT = -1 };

class ScanRegistration {
 public:
  RegistrationParams cfg;
  Cloud laserCloud, cornerSharp, cornerLessSharp, surfFlat, surfLessFlat;
  std::vector<std::pair<size_t, size_t>> scanIndices;  // inclusive [first, second]
  float imuTrans[12] = {0};

  // ---- IMU state
  struct IMUState {
    double stamp = 0;
    Angle roll, pitch, yaw;
    Vec3 position, velocity, acceleration;
  };
  std::deque<IMUState> imuHistory;   // CircularBuffer<IMUState>, capacity max(200, cfg.imuHistorySize) (configure :48-53, see update_imu_data)
  size_t imuIdx = 0;
  IMUState imuStart, imuCur;
  Vec3 imuPositionShift;
  double scanTime = 0, sweepStart = 0;
  bool has_imu() const { return !imuHistory.empty(); }

  // updateIMUData(acc, newState) :82-98 — acc in the local frame, newState with stamp / roll / pitch / yaw / acceleration set
  void update_imu_data(Vec3 acc, IMUState newState) {
    if (!imuHistory.empty()) {
      rotateZXY(acc, newState.roll, newState.pitch, newState.yaw);
      const IMUState& prev = imuHistory.back();
      const float timeDiff = (float)(newState.stamp - prev.stamp);
      newState.position = {prev.position.x + prev.velocity.x * timeDiff + 0.5f * acc.x * timeDiff * timeDiff,
                           prev.position.y + prev.velocity.y * timeDiff + 0.5f * acc.y * timeDiff * timeDiff,
                           prev.position.z + prev.velocity.z * timeDiff + 0.5f * acc.z * timeDiff * timeDiff};
      newState.velocity = {prev.velocity.x + acc.x * timeDiff, prev.velocity.y + acc.y * timeDiff, prev.velocity.z + acc.z * timeDiff};
    }
    // capacity: the buffer is constructed with 200 slots and configure() only ever GROWS it (ensureCapacity, CircularBuffer.h:
    // 53-66: `_capacity < reqCapacity`), so an imuHistorySize below 200 has no effect
    if (imuHistory.size() >= (size_t)std::max(200, cfg.imuHistorySize)) imuHistory.pop_front();
    imuHistory.push_back(newState);
  }
  static void interpolate(const IMUState& start, const IMUState& end, float ratio, IMUState& result) {   // .h:107-131
    const float invRatio = 1 - ratio;
    result.roll = Angle(start.roll.rad() * invRatio + end.roll.rad() * ratio);
    result.pitch = Angle(start.pitch.rad() * invRatio + end.pitch.rad() * ratio);
    if (start.yaw.rad() - end.yaw.rad() > M_PI) {
      result.yaw = Angle((float)(start.yaw.rad() * invRatio + (end.yaw.rad() + 2 * M_PI) * ratio));
    } else if (start.yaw.rad() - end.yaw.rad() < -M_PI) {
      result.yaw = Angle((float)(start.yaw.rad() * invRatio + (end.yaw.rad() - 2 * M_PI) * ratio));
    } else {
      result.yaw = Angle(start.yaw.rad() * invRatio + end.yaw.rad() * ratio);
    }
    result.velocity = {start.velocity.x * invRatio + end.velocity.x * ratio, start.velocity.y * invRatio + end.velocity.y * ratio,
                       start.velocity.z * invRatio + end.velocity.z * ratio};
    result.position = {start.position.x * invRatio + end.position.x * ratio, start.position.y * invRatio + end.position.y * ratio,
                       start.position.z * invRatio + end.position.z * ratio};
  }
  void interpolate_imu_state_for(float relTime, IMUState& out) {   // :133-147
    double timeDiff = (scanTime - imuHistory[imuIdx].stamp) + relTime;
    while (imuIdx < imuHistory.size() - 1 && timeDiff > 0) {
      imuIdx++;
      timeDiff = (scanTime - imuHistory[imuIdx].stamp) + relTime;
    }
    if (imuIdx == 0 || timeDiff > 0) {
      out = imuHistory[imuIdx];
    } else {
      const float ratio = (float)(-timeDiff / (imuHistory[imuIdx].stamp - imuHistory[imuIdx - 1].stamp));
      interpolate(imuHistory[imuIdx], imuHistory[imuIdx - 1], ratio, out);
    }
  }
  void project_point_to_start_of_sweep(Pt& point, float relTime) {   // :101-131
    if (!has_imu()) return;
    interpolate_imu_state_for(relTime, imuCur);                       // setIMUTransformFor :112-118
    const float relSweepTime = (float)((scanTime - sweepStart) + relTime);
    imuPositionShift = {imuCur.position.x - imuStart.position.x - imuStart.velocity.x * relSweepTime,
                        imuCur.position.y - imuStart.position.y - imuStart.velocity.y * relSweepTime,
                        imuCur.position.z - imuStart.position.z - imuStart.velocity.z * relSweepTime};
    rotateZXY(point, imuCur.roll, imuCur.pitch, imuCur.yaw);          // transformToStartIMU :122-131
    point.x += imuPositionShift.x;
    point.y += imuPositionShift.y;
    point.z += imuPositionShift.z;
    rotateYXZ(point, -imuStart.yaw, -imuStart.pitch, -imuStart.roll);
  }
  void reset(double t) {   // :55-79
    scanTime = t;
    imuIdx = 0;
    if (has_imu()) interpolate_imu_state_for(0, imuStart);
    sweepStart = t;
  }
  void update_imu_transform() {   // :258-281
    imuTrans[0] = imuStart.pitch.rad(); imuTrans[1] = imuStart.yaw.rad(); imuTrans[2] = imuStart.roll.rad();
    imuTrans[3] = imuCur.pitch.rad(); imuTrans[4] = imuCur.yaw.rad(); imuTrans[5] = imuCur.roll.rad();
    Vec3 shift = imuPositionShift;
    rotateYXZ(shift, -imuStart.yaw, -imuStart.pitch, -imuStart.roll);
    imuTrans[6] = shift.x; imuTrans[7] = shift.y; imuTrans[8] = shift.z;
    Vec3 vel = {imuCur.velocity.x - imuStart.velocity.x, imuCur.velocity.y - imuStart.velocity.y, imuCur.velocity.z - imuStart.velocity.z};
    rotateYXZ(vel, -imuStart.yaw, -imuStart.pitch, -imuStart.roll);
    imuTrans[9] = vel.x; imuTrans[10] = vel.y; imuTrans[11] = vel.z;
  }
  // processScanlines(scanTime, laserCloudScans) :28-46 with the IMU bookkeeping around it
  void process_scanlines_at(double t, const std::vector<Cloud>& rings) {
    reset(t);
    process_scanlines(rings);
    update_imu_transform();
  }

  // rings: one cloud per scan ring, already in the LOAM camera frame, intensity = ring + relTime.
  void process_scanlines(const std::vector<Cloud>& rings) {
    laserCloud.clear(); cornerSharp.clear(); cornerLessSharp.clear(); surfFlat.clear(); surfLessFlat.clear();
    scanIndices.clear();
    size_t cloudSize = 0;
    for (const Cloud& r : rings) {
      laserCloud.insert(laserCloud.end(), r.begin(), r.end());
      size_t first = cloudSize;
      cloudSize += r.size();
      scanIndices.emplace_back(first, cloudSize > 0 ? cloudSize - 1 : 0);
    }
    extract_features();
  }

 private:
  std::vector<float> regionCurvature_;
  std::vector<int> regionLabel_;
  std::vector<size_t> regionSort_;
  std::vector<int> picked_;

  void extract_features() {
    const size_t cr = (size_t)cfg.curvatureRegion;
    const size_t nreg = (size_t)cfg.nFeatureRegions;
    for (size_t i = 0; i < scanIndices.size(); i++) {
      Cloud lessFlatScan;
      const size_t s0 = scanIndices[i].first, e0 = scanIndices[i].second;
      if (e0 <= s0 + 2 * cr) continue;
      set_scan_buffers(s0, e0);
      for (size_t j = 0; j < nreg; j++) {
        size_t sp = ((s0 + cr) * (nreg - j) + (e0 - cr) * j) / nreg;
        size_t ep = ((s0 + cr) * (nreg - 1 - j) + (e0 - cr) * (j + 1)) / nreg - 1;
        if (ep <= sp) continue;
        const size_t regionSize = ep - sp + 1;
        set_region_buffers(sp, ep);

        int largestPicked = 0;
        for (size_t k = regionSize; k > 0 && largestPicked < cfg.maxCornerLessSharp;) {
          size_t idx = regionSort_[--k];
          size_t scanIdx = idx - s0, regionIdx = idx - sp;
          if (picked_[scanIdx] == 0 && regionCurvature_[regionIdx] > cfg.surfaceCurvatureThreshold) {
            largestPicked++;
            if (largestPicked <= cfg.maxCornerSharp) {
              regionLabel_[regionIdx] = CORNER_SHARP;
              cornerSharp.push_back(laserCloud[idx]);
            } else {
              regionLabel_[regionIdx] = CORNER_LESS_SHARP;
            }
            cornerLessSharp.push_back(laserCloud[idx]);
            mark_as_picked(idx, scanIdx);
          }
        }
        int smallestPicked = 0;
        for (size_t k = 0; k < regionSize && smallestPicked < cfg.maxSurfaceFlat; k++) {
          size_t idx = regionSort_[k];
          size_t scanIdx = idx - s0, regionIdx = idx - sp;
          if (picked_[scanIdx] == 0 && regionCurvature_[regionIdx] < cfg.surfaceCurvatureThreshold) {
            smallestPicked++;
            regionLabel_[regionIdx] = SURFACE_FLAT;
            surfFlat.push_back(laserCloud[idx]);
            mark_as_picked(idx, scanIdx);
          }
        }
        for (size_t k = 0; k < regionSize; k++)
          if (regionLabel_[k] <= SURFACE_LESS_FLAT) lessFlatScan.push_back(laserCloud[sp + k]);
      }
      Cloud ds;
      voxel_grid(lessFlatScan, cfg.lessFlatFilterSize, ds);
      surfLessFlat.insert(surfLessFlat.end(), ds.begin(), ds.end());
    }
  }

  void set_region_buffers(size_t startIdx, size_t endIdx) {
    const size_t n = endIdx - startIdx + 1;
    regionCurvature_.resize(n);
    regionSort_.resize(n);
    regionLabel_.assign(n, SURFACE_LESS_FLAT);
    const float w = -2 * cfg.curvatureRegion;
    for (size_t i = startIdx, r = 0; i <= endIdx; i++, r++) {
      float dx = w * laserCloud[i].x, dy = w * laserCloud[i].y, dz = w * laserCloud[i].z;
      for (int j = 1; j <= cfg.curvatureRegion; j++) {
        dx += laserCloud[i + j].x + laserCloud[i - j].x;
        dy += laserCloud[i + j].y + laserCloud[i - j].y;
        dz += laserCloud[i + j].z + laserCloud[i - j].z;
      }
      regionCurvature_[r] = dx * dx + dy * dy + dz * dz;
      regionSort_[r] = i;
    }
    // the reference's quadratic exchange pass (:311-317): only strictly smaller neighbours move ahead,
    // so equal curvatures keep their cloud order (stable, ascending).
    for (size_t i = 1; i < n; i++)
      for (size_t j = i; j >= 1; j--)
        if (regionCurvature_[regionSort_[j] - startIdx] < regionCurvature_[regionSort_[j - 1] - startIdx])
          std::swap(regionSort_[j], regionSort_[j - 1]);
  }

  void set_scan_buffers(size_t startIdx, size_t endIdx) {
    const size_t cr = (size_t)cfg.curvatureRegion;
    picked_.assign(endIdx - startIdx + 1, 0);
    for (size_t i = startIdx + cr; i < endIdx - cr; i++) {
      const Pt& prev = laserCloud[i - 1];
      const Pt& pt = laserCloud[i];
      const Pt& next = laserCloud[i + 1];
      float diffNext = sq_diff(next, pt);
      if (diffNext > 0.1) {   // double comparison, as in the source
        float depth1 = pt_dist(pt), depth2 = pt_dist(next);
        if (depth1 > depth2) {
          float wd = std::sqrt(sq_diff_w(next, pt, depth2 / depth1)) / depth2;
          if (wd < 0.1) {
            std::fill_n(&picked_[i - startIdx - cr], cr + 1, 1);
            continue;
          }
        } else {
          float wd = std::sqrt(sq_diff_w(pt, next, depth1 / depth2)) / depth1;
          if (wd < 0.1) std::fill_n(&picked_[i - startIdx + 1], cr + 1, 1);
        }
      }
      float diffPrev = sq_diff(pt, prev);
      float dis = sq_pt_dist(pt);
      if (diffNext > 0.0002 * dis && diffPrev > 0.0002 * dis) picked_[i - startIdx] = 1;
    }
  }

  void mark_as_picked(size_t cloudIdx, size_t scanIdx) {
    picked_[scanIdx] = 1;
    for (int i = 1; i <= cfg.curvatureRegion; i++) {
      if (sq_diff(laserCloud[cloudIdx + i], laserCloud[cloudIdx + i - 1]) > 0.05) break;
      picked_[scanIdx + i] = 1;
    }
    for (int i = 1; i <= cfg.curvatureRegion; i++) {
      if (sq_diff(laserCloud[cloudIdx - i], laserCloud[cloudIdx - i + 1]) > 0.05) break;
      picked_[scanIdx - i] = 1;
    }
  }
};

}  // namespace loam_oracle
