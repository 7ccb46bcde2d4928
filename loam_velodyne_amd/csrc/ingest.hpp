// Raw-sweep ingestion: MultiScanRegistration::process, src/lib/MultiScanRegistration.cpp:160-238 —
// axis remap, NaN / zero / out-of-field rejection, vertical angle -> ring (MultiScanMapper :41-66), azimuth -> relTime
// with the halfPassed unwrapping, stable split into per-ring clouds.
#pragma once
#include "common.h"

namespace loamx {

struct MapperParams {
  float lower, upper, factor;   // degrees, degrees, (n_rings - 1) / (upper - lower) as a float (MultiScanRegistration.cpp:41-50)
  uint32_t n_rings;
};

// IMU de-skew of the kept points (projectPointToStartOfSweep, src/lib/BasicScanRegistration.cpp:101-147): the host keeps the
// IMU state machine and hands the device one table per sweep.
struct ImuTable {
  uint32_t H = 0;                // history length; 0 = no IMU data (projection is the identity)
  uint32_t idx0 = 0;             // _imuIdx when the sweep's loop starts
  const double* dt = nullptr;    // [H] toSec(_scanTime - history[j].stamp)
  const double* dstamp = nullptr;   // [H] toSec(history[j].stamp - history[j-1].stamp), j >= 1
  const float* state = nullptr;  // [H][9] roll, pitch, yaw, position xyz, velocity xyz
  float start_c[3] = {1, 1, 1}, start_s[3] = {0, 0, 0};   // cos / sin of _imuStart roll, pitch, yaw
  float start_pos[3] = {0, 0, 0}, start_vel[3] = {0, 0, 0};
  double rel_sweep_base = 0;     // toSec(_scanTime - _sweepStart)
};
// _imuCur / _imuPositionShift as the LAST kept point of the sweep leaves them (they feed updateIMUTransform, :258-281)
struct ImuLast {
  float roll, pitch, yaw, pos[3], vel[3], shift[3];
  uint32_t idx, valid;
};

// device-side re-striding of a raw payload (x, y, z at 0 / 4 / 8 of every `stride` bytes) into float4 records
void raw_unpack(const void* d_bytes, uint32_t stride, uint32_t n, float4* d_out, hipStream_t st);

class RawBinner {
 public:
  static constexpr uint32_t MAX_RINGS = 256;
  void init(hipStream_t st) { st_ = st; }
  // d_raw: n records (x, y, z, unused) in sensor axes and firing order.  d_out (capacity n): the kept points in the LOAM
  // frame, rings concatenated, intensity = ring + relTime.  d_ring_cnt[n_rings]: points per ring.  Asynchronous.
  void run(const float4* d_raw, uint32_t n, const MapperParams& m, float scan_period, float4* d_out, uint32_t* d_ring_cnt,
           const ImuTable* imu = nullptr, ImuLast* d_last = nullptr);

 private:
  hipStream_t st_ = nullptr;
  DevBuf<int> ring_of_;
  DevBuf<uint32_t> blk_cnt_, blk_pre_, scratch_, imu_first_, blk_idx_;
};

}  // namespace loamx
