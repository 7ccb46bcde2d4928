// Sweep-to-sweep registration (BasicLaserOdometry) for n independent streams; loamx_odom_* is the 1-stream case.
#pragma once
#include "common.h"
#include "host_math.h"
#include "registration.cuh"

namespace loamx {

struct OdomParams {
  float scan_period = 0.1f;
  int max_iterations = 25;
  float delta_t_abort = 0.1f, delta_r_abort = 0.1f;
};

struct OdomStats {
  int iterations, sel, frame, degenerate;
};

// device-side description of one stream's odometry problem
struct OdomProblem {
  const float4* sharp; uint32_t n_sharp;
  const float4* flat; uint32_t n_flat;
  const float4* last_corner; uint32_t n_last_corner;     // ring-ordered clouds (scan windows walk these)
  const float4* last_surf; uint32_t n_last_surf;
  const float4* lc_sorted; const uint32_t* lc_cell; const GridDesc* lc_desc;   // grid index over last_corner
  const float4* ls_sorted; const uint32_t* ls_cell; const GridDesc* ls_desc;   // grid index over last_surf
  int* ind;            // 5 ints per feature: corner (ind1, ind2, -, -, -) / surf (ind1, ind2, ind3, -, -)
  float transform[6];  // in: initial _transform, out: optimised
  OdomStats stats;
  int done;
  float matP[36];
};

// device pointers to one sweep's four feature clouds
struct OdomInput {
  const float4* sharp; uint32_t n_sharp;
  const float4* less_sharp; uint32_t n_less_sharp;
  const float4* flat; uint32_t n_flat;
  const float4* less_flat; uint32_t n_less_flat;
};

struct OdomStream {
  bool inited = false;
  long frame = 0;
  HTwist transform, transform_sum;
  HAngle imu_roll_start, imu_pitch_start, imu_yaw_start, imu_roll_end, imu_pitch_end, imu_yaw_end;
  HVec3 imu_shift, imu_velo;
  OdomStats stats = {0, 0, 0, 0};
  DevBuf<float4> cur_corner, cur_surf, last_corner, last_surf;   // less-sharp / less-flat of the current and previous sweep
  uint32_t n_last_corner = 0, n_last_surf = 0;
  SubMapIndex idx_corner, idx_surf;
  DevBuf<int> ind;
};

class OdometryBatch {
 public:
  OdometryBatch(int device, uint32_t n_streams, hipStream_t shared_stream = nullptr);
  ~OdometryBatch();
  OdomParams params;
  uint32_t n_streams() const { return (uint32_t)streams_.size(); }
  OdomStream& stream_state(uint32_t s) { return *streams_[s]; }
  hipStream_t stream() const { return st_; }
  void update_imu(uint32_t s, const float* t12);
  // one sweep per stream, inputs already on the device (same HIP stream or synchronised).  Synchronous.
  // rc[s] = LOAMX_SKIPPED for a stream's first (initialising) sweep.
  void process(const OdomInput* in, int* rc);
  // host-cloud convenience for stream s only (other streams untouched)
  int process_host(uint32_t s, const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat,
                   const loamx_cloud* less_flat);
  int get_last_clouds(uint32_t s, loamx_cloud* corner, loamx_cloud* surf);
  int transform_to_end_host(uint32_t s, loamx_cloud* cloud);
  // in place on device points with stream s's current transform (async on the stream)
  void to_end_device(uint32_t s, float4* pts, uint32_t n);

 private:
  int device_;
  hipStream_t st_ = nullptr;
  bool own_stream_ = false;
  std::vector<OdomStream*> streams_;
  DevBuf<OdomProblem> prob_;
  PinBuf<OdomProblem> h_prob_;
  PinBuf<float4> h_stage_;
  DevBuf<float4> up_[4], tmp_cloud_;
  void process_subset(const std::vector<uint32_t>& which, const OdomInput* in, int* rc);
};

}  // namespace loamx
