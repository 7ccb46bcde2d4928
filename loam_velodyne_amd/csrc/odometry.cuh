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

struct ToEndParams;

// device-side description of one stream's odometry problem
struct OdomProblem {
  const float4* sharp; uint32_t n_sharp;
  const float4* flat; uint32_t n_flat;
  const float4* last_corner; uint32_t n_last_corner;     // ring-ordered clouds (scan windows walk these)
  const float4* last_surf; uint32_t n_last_surf;
  const float4* sorted;            // concatenated cell-sorted points of the batch index
  const uint32_t* cell_table;      // concatenated cell tables
  const GridDescB* lc_desc;        // grid over last_corner
  const GridDescB* ls_desc;        // grid over last_surf
  int* ind;            // 5 ints per feature: corner (ind1, ind2, -, -, -) / surf (ind1, ind2, ind3, -, -)
  float transform[6];  // in: initial _transform, out: optimised
  OdomStats stats;
  int done;
  float matP[36];
  int stream_id;          // index of the stream this problem belongs to
  ToEndParams* te_out;        // re-projection parameters of this stream (completed by k_odom_lm)
  OdomProblem* host_mirror;   // pinned host copy that k_odom_lm fills with transform / stats / done (no D2H copy on the stream)
  unsigned ticket;        // k_odom_lm: workgroup arrivals since the problem was set up (per-stream barrier)
  double* part;           // k_odom_lm: [2][16][LX_NSUM] partial normal equations of the stream's workgroups
  const uint32_t* rf_corner;   // ring-first tables of last_corner / last_surf (k_odom_corr_lds: where to expect the ring windows — a hint)
  const uint32_t* rf_surf;
  uint32_t rf_epoch;
  uint32_t* err_word;     // pinned host word raised when k_odom_lm's exchange times out (checked by the host after the pose event)
};

// one sweep's four feature clouds on the device.  less_sharp / less_flat of ALL streams must be contiguous in stream
// order (that is how the feature extractor emits them): stream s's cloud starts where stream s-1's ends.
struct OdomInput {
  const float4* sharp; uint32_t n_sharp;
  const float4* less_sharp; uint32_t n_less_sharp;
  const float4* flat; uint32_t n_flat;
  const float4* less_flat; uint32_t n_less_flat;
};

struct ToEndParams {
  float T[6];
  float sT[3], cT[3];                   // sin/cos of the transform angles (x, y, z)
  float shift[3];                       // imuShiftFromStart
  float s_start[3], c_start[3];         // imu pitch/yaw/roll start (x=pitch, y=yaw, z=roll)
  float s_end[3], c_end[3];
  float scan_period;
  int enabled;
};

struct OdomStream {
  bool inited = false;
  long frame = 0;
  HTwist transform, transform_sum;
  HAngle imu_roll_start, imu_pitch_start, imu_yaw_start, imu_roll_end, imu_pitch_end, imu_yaw_end;
  HVec3 imu_shift, imu_velo;
  OdomStats stats = {0, 0, 0, 0};
  uint32_t n_last_corner = 0, n_last_surf = 0;
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
  // one sweep for EVERY stream, inputs already on the device (same HIP stream or synchronised).  Synchronous.
  // rc[s] = LOAMX_SKIPPED for a stream's first (initialising) sweep.
  // defer_tail: return once the poses are known; the re-projected clouds / their index are ready at tail_event()
  void process(const OdomInput* in, int* rc, bool defer_tail = false);
  hipEvent_t tail_event() const { return tail_pending_ ? ev_tail_ : nullptr; }
  // host-cloud convenience (single-stream handles)
  int process_host(const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat, const loamx_cloud* less_flat);
  int get_last_clouds(uint32_t s, loamx_cloud* corner, loamx_cloud* surf);
  int transform_to_end_host(uint32_t s, loamx_cloud* cloud);
  // in place on device points with stream s's current transform (async on the stream)
  void to_end_device(uint32_t s, float4* pts, uint32_t n);
  // dst[h_off[k] .. h_off[k+1]) = transformToEnd(src[k]) with seg_params[k]; one launch on `stream`
  void to_end_gather(float4* dst, const uint32_t* h_off, const float4* const* src, const ToEndParams* seg_params, uint32_t K,
                     hipStream_t stream);
  ToEndParams to_end_params(uint32_t s, bool enabled) const;
  // device views of the re-projected clouds handed on to mapping
  const float4* d_last_corner(uint32_t s) const { return last_.p + h_last_off_[s]; }
  const float4* d_last_surf(uint32_t s) const { return last_.p + h_last_off_[n_streams() + s]; }

 private:
  int device_;
  hipStream_t st_ = nullptr;
  bool own_stream_ = false;
  std::vector<OdomStream*> streams_;
  // clouds of all streams, concatenated: [corner_0 .. corner_{ns-1} | surf_0 .. surf_{ns-1}], offsets 2*ns+1
  DevBuf<float4> cur_, last_, prev_;   // being written | handed on by the last call | handed on by the call before (still read by its consumer)
  std::vector<uint32_t> h_cur_off_, h_last_off_;
  SubMapIndexBatch index_;
  DevBuf<int> ind_;
  DevBuf<double> part_;
  PinBuf<OdomProblem> h_mirror_;
  PinBuf<uint32_t> h_err_;
  DevBuf<uint32_t> rf_;   // [2 * n_streams][OD_RF_N] ring-first tables of the clouds handed on by the last call (entries tagged with rf_epoch_)
  uint32_t rf_epoch_ = 0;
  // what a call sends up before its first kernel — the problems, the re-projection parameters, the cloud offsets — is ONE block in
  // pinned memory and ONE copy (three copies were three ~7 us commands at the head of the odometry chain, the pipeline's longest)
  template <class T> struct View { T* p = nullptr; };
  DevBuf<char> up_dev_;
  PinBuf<char> up_host_;
  size_t up_bytes_ = 0;
  View<OdomProblem> prob_, h_prob_;
  View<ToEndParams> te_, h_te_;
  View<uint32_t> d_cur_off_, h_off_pin_;
  PinBuf<float4> h_stage_;
  DevBuf<float4> up_[4], tmp_cloud_;
  uint32_t lm_slots_[2] = {0, 0};   // workgroups of k_odom_lm<1> / <2> the device holds at once (occupancy x CUs)
  hipEvent_t ev_tail_ = nullptr, ev_pose_ = nullptr, ev_up_ = nullptr;
  bool tail_pending_ = false, up_pending_ = false;
  PinBuf<char> h_gather_;
  DevBuf<char> d_gather_;
};

}  // namespace loamx
