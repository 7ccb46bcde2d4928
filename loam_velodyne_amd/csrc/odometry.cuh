// Sweep-to-sweep registration (BasicLaserOdometry) host class.
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

// device-side description of one sweep's odometry problem
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
};

class Odometry {
 public:
  explicit Odometry(int device);
  ~Odometry();
  OdomParams params;
  void update_imu(const float* t12);
  int process(const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat, const loamx_cloud* less_flat);
  void get_transform(float* t6) const { transform_.get(t6); }
  void get_transform_sum(float* t6) const { transform_sum_.get(t6); }
  void set_transform(const float* t6) { transform_.set(t6); }
  void set_transform_sum(const float* t6) { transform_sum_.set(t6); }
  int get_last_clouds(loamx_cloud* corner, loamx_cloud* surf);
  int transform_to_end(loamx_cloud* cloud);
  OdomStats stats() const { return stats_; }

 private:
  int device_;
  hipStream_t st_ = nullptr;
  bool inited_ = false;
  long frame_ = 0;
  HTwist transform_, transform_sum_;
  HAngle imu_roll_start_, imu_pitch_start_, imu_yaw_start_, imu_roll_end_, imu_pitch_end_, imu_yaw_end_;
  HVec3 imu_shift_, imu_velo_;
  OdomStats stats_ = {0, 0, 0, 0};

  PinBuf<float4> h_stage_;
  DevBuf<float4> sharp_, flat_, less_sharp_, less_flat_, last_corner_, last_surf_, tmp_cloud_;
  uint32_t n_last_corner_ = 0, n_last_surf_ = 0;
  SubMapIndex idx_corner_, idx_surf_;
  DevBuf<int> ind_;
  DevBuf<OdomProblem> prob_;
  PinBuf<OdomProblem> h_prob_;

  void upload_cloud(const loamx_cloud* c, DevBuf<float4>& dst);
  void to_end_device(float4* pts, uint32_t n);
};

}  // namespace loamx
