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

class RawBinner {
 public:
  static constexpr uint32_t MAX_RINGS = 256;
  void init(hipStream_t st) { st_ = st; }
  // d_raw: n records (x, y, z, unused) in sensor axes and firing order.  d_out (capacity n): the kept points in the LOAM
  // frame, rings concatenated, intensity = ring + relTime.  d_ring_cnt[n_rings]: points per ring.  Asynchronous.
  void run(const float4* d_raw, uint32_t n, const MapperParams& m, float scan_period, float4* d_out, uint32_t* d_ring_cnt);

 private:
  hipStream_t st_ = nullptr;
  DevBuf<int> ring_of_;
  DevBuf<uint32_t> blk_cnt_, blk_pre_, scratch_;
};

}  // namespace loamx
