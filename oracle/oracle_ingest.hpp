// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp for the rules and for what is / is not pinned by the reference's own code).
//
// Raw-sweep ingestion, restating the ROS-free part of MultiScanRegistration:
//   MultiScanMapper::set / getRingForAngle -> src/lib/MultiScanRegistration.cpp:41-66,
//                                             include/loam_velodyne/MultiScanRegistration.h:83-89 (presets .h:60-75)
//   bin_sweep                               -> MultiScanRegistration::process, src/lib/MultiScanRegistration.cpp:160-238
// Input: the raw points of one revolution in sensor axes (x forward, y left, z up) in firing order; output: one cloud per
// scan ring in the LOAM camera frame (x = y_in, y = z_in, z = x_in), intensity = ring + relTime.
// projectPointToStartOfSweep (:231, BasicScanRegistration.cpp:101-109) de-skews with the IMU state of the scan
// registration passed in (identity without IMU data).
// Types follow the reference: float everywhere, with the double promotions that the M_PI / 0.0001 / 0.5 literals cause.
#pragma once
#include "oracle_features.hpp"
#include <cmath>

namespace loam_oracle {

struct MultiScanMapper {
  float lowerBound = -15, upperBound = 15;
  uint16_t nScanRings = 16;
  float factor = (16 - 1) / (15.0f - (-15.0f));
  void set(float lo, float hi, uint16_t n) {   // :41-50
    lowerBound = lo; upperBound = hi; nScanRings = n;
    factor = (n - 1) / (hi - lo);
  }
  int getRingForAngle(float angle) const {     // :64-66 (float * int -> float, / M_PI -> double from there on)
    return int(((angle * 180 / M_PI) - lowerBound) * factor + 0.5);
  }
};

// raw: n records of (x, y, z); returns one cloud per ring
// sr (optional): the scan registration whose IMU state de-skews every kept point (projectPointToStartOfSweep :231)
// trace (optional, tests): the kept points in FIRING order before the IMU projection — (x, y, z, intensity), ring, relTime
struct BinTrace {
  Cloud points;
  std::vector<int> ring;
  std::vector<float> relTime;
};
inline std::vector<Cloud> bin_sweep(const float* raw, size_t n, const MultiScanMapper& mapper, float scanPeriod, ScanRegistration* sr = nullptr,
                                    BinTrace* trace = nullptr) {
  std::vector<Cloud> scans(mapper.nScanRings);
  if (n == 0) return scans;
  // scan start and end orientations (:165-173)
  float startOri = -std::atan2(raw[1], raw[0]);
  float endOri = -std::atan2(raw[3 * (n - 1) + 1], raw[3 * (n - 1)]) + 2 * float(M_PI);
  if (endOri - startOri > 3 * M_PI) {
    endOri -= 2 * M_PI;
  } else if (endOri - startOri < M_PI) {
    endOri += 2 * M_PI;
  }
  bool halfPassed = false;
  for (size_t i = 0; i < n; i++) {
    Pt point;
    point.x = raw[3 * i + 1];   // :184-186 axis remap
    point.y = raw[3 * i + 2];
    point.z = raw[3 * i];
    if (!std::isfinite(point.x) || !std::isfinite(point.y) || !std::isfinite(point.z)) continue;   // :189-193
    if (point.x * point.x + point.y * point.y + point.z * point.z < 0.0001) continue;             // :196-198
    float angle = std::atan(point.y / std::sqrt(point.x * point.x + point.z * point.z));           // :201
    int scanID = mapper.getRingForAngle(angle);
    if (scanID >= mapper.nScanRings || scanID < 0) continue;                                        // :203-205
    float ori = -std::atan2(point.x, point.z);                                                      // :208
    if (!halfPassed) {
      if (ori < startOri - M_PI / 2) {
        ori += 2 * M_PI;
      } else if (ori > startOri + M_PI * 3 / 2) {
        ori -= 2 * M_PI;
      }
      if (ori - startOri > M_PI) halfPassed = true;
    } else {
      ori += 2 * M_PI;
      if (ori < endOri - M_PI * 3 / 2) {
        ori += 2 * M_PI;
      } else if (ori > endOri + M_PI / 2) {
        ori -= 2 * M_PI;
      }
    }
    float relTime = scanPeriod * (ori - startOri) / (endOri - startOri);   // :228
    point.i = scanID + relTime;                                             // :229
    if (trace) { trace->points.push_back(point); trace->ring.push_back(scanID); trace->relTime.push_back(relTime); }
    if (sr) sr->project_point_to_start_of_sweep(point, relTime);            // :231
    scans[scanID].push_back(point);
  }
  return scans;
}

}  // namespace loam_oracle
