// Per-sweep feature extraction (BasicScanRegistration::extractFeatures, IMU-less) for a batch of sweeps.
#pragma once
#include "common.h"
#include "host_math.h"
#include "ingest.hpp"
#include <deque>
#include "voxel.hpp"

namespace loamx {

struct FeatParams {
  float scan_period = 0.1f;
  int n_regions = 6;
  int curv_region = 5;
  int max_sharp = 2;
  int max_less_sharp = 20;
  int max_flat = 4;
  float less_flat_leaf = 0.2f;
  float curv_thr = 0.1f;
};

// The IMU state machine of BasicScanRegistration for ONE sensor stream, on the host (src/lib/BasicScanRegistration.cpp:55-152,
// :258-281): history with accumulated position / velocity (updateIMUData :82-98), interpolation (interpolateIMUStateFor
// :133-147), reset(scanTime) :55-79 and updateIMUTransform :258-281.  The per-point projection (projectPointToStartOfSweep
// :101-131) runs on the device from a table this class fills (ingest.hpp ImuTable); the state the projection loop leaves behind
// comes back as ImuLast.  Used by FeatureExtractor (single-sweep entry points) and, one per stream, by the batched pipeline.
class ImuTracker {
 public:
  struct State {
    double stamp = 0;
    HAngle roll, pitch, yaw;
    HVec3 position, velocity, acceleration;
  };
  int history_size = 200;                                   // max(200, RegistrationParams::imuHistorySize): the buffer never shrinks
  void update(double stamp, float roll, float pitch, float yaw, const float acc[3]);   // updateIMUData :82-98
  void set_scan_time(double t) { next_scan_time_ = t; }     // the scanTime argument of the next process call
  void begin_sweep();                                       // reset(scanTime) + updateIMUTransform() around processScanlines
  const float* imu_trans() const { return imu_trans_; }     // imuTransform(): 4 x (x, y, z)
  uint32_t size() const { return (uint32_t)hist_.size(); }
  // the table of the next projection loop: h_d[2 H] (dt, dstamp), h_f[9 H] (states); I gets everything but the device pointers
  void fill_table(double* h_d, float* h_f, ImuTable& I) const;
  void apply_last(const ImuLast& L);                        // _imuCur / _imuPositionShift / _imuIdx as the last kept point left them

 private:
  std::deque<State> hist_;
  size_t idx_ = 0;
  State start_, cur_;
  HVec3 shift_;
  double scan_time_ = 0, sweep_start_ = 0, next_scan_time_ = 0;
  float imu_trans_[12] = {0};
  void interpolate_for_(float rel_time, State& out);
};

class FeatureExtractor {
 public:
  FeatureExtractor(int device, hipStream_t shared_stream = nullptr);
  ~FeatureExtractor();
  FeatParams params;
  hipStream_t stream() const { return st_; }
  int device() const { return device_; }

  // stage nsw sweeps: cloud[s] = rings concatenated, ring_size[s][0..n_rings[s])
  void upload(uint32_t nsw, const loamx_cloud* clouds, const uint32_t* const* ring_size, const uint32_t* n_rings, bool allow_direct = false);
  // the same without blocking: every copy goes to `copy_stream` (packed float4 clouds straight from the caller's memory —
  // pinned memory makes that a true DMA — other layouts through this object's pinned staging), `done` is recorded behind
  // them; run_async() must be ordered behind `done`.  The caller's buffers are read until `done` has completed.
  // big_copy (optional): asked to carry each block copy itself (dst, src, bytes) — true: taken (the caller orders run_async() behind it by
  // its own means), false: the copy goes to copy_stream as usual
  void upload_async(uint32_t nsw, const loamx_cloud* clouds, const uint32_t* const* ring_size, const uint32_t* n_rings, hipStream_t copy_stream,
                    hipEvent_t done, const std::function<bool(void*, const void*, size_t)>& big_copy = nullptr);
  // the same from DEVICE memory (e.g. the output of the raw-sweep binning): sweep s = d_src + src_off[s], rings concatenated;
  // copies device-to-device on `copy_stream`, `done` recorded behind them
  void upload_device(uint32_t nsw, const float4* d_src, const uint32_t* src_off, const uint32_t* const* ring_size, const uint32_t* n_rings,
                     hipStream_t copy_stream, hipEvent_t done);
  // one raw revolution in sensor axes / firing order (MultiScanRegistration::process): binned into rings on the device and,
  // with IMU data, de-skewed point by point (projectPointToStartOfSweep)
  void upload_raw(const void* raw_xyz, uint32_t count, uint32_t stride, float lower_deg, float upper_deg, uint32_t n_scan_rings);
  // ---- IMU state machine of BasicScanRegistration (src/lib/BasicScanRegistration.cpp:55-152, :258-281); times in seconds
  void update_imu_data(double stamp, float roll, float pitch, float yaw, const float acc[3]) { imu_.history_size = imu_history_size; imu_.update(stamp, roll, pitch, yaw, acc); }
  void set_scan_time(double t) { imu_.set_scan_time(t); }   // the scanTime argument of the next process call
  void begin_sweep() { imu_.begin_sweep(); }                // reset(scanTime) + updateIMUTransform() around processScanlines
  const float* imu_trans() const { return imu_.imu_trans(); }   // imuTransform(): 4 x (x, y, z)
  int imu_history_size = 200;                               // RegistrationParams::imuHistorySize
  int download_cloud(uint32_t sweep, loamx_cloud* full, uint32_t* ring_size_out);
  // mirror_offsets: the offset tables also go to pinned host memory, with an event behind each compaction (device_results_front / _lf)
  void run_async(bool mirror_offsets = false);
  void sync();
  // after a synchronisation point behind run_async(): throws LOAMX_E_INVALID when k_feat_point met a non-finite coordinate (and clears the mark)
  void check_finite_input();
  // host copies of one sweep's outputs (after sync); any pointer may be NULL
  int download(uint32_t sweep, loamx_cloud* sharp, loamx_cloud* less_sharp, loamx_cloud* flat, loamx_cloud* less_flat);

  // the same clouds left where they are: waits for the run (and raises what download() would raise), then hands out the device views
  // and sizes of one sweep's sharp / less sharp / flat / less flat clouds (valid until the next upload)
  void device_results(uint32_t sweep, const float4* ptr[4], uint32_t count[4]);
  // the same in two steps behind a run_async(true) whose rings fit the LDS voxel grid (split_results()): sharp / less sharp / flat as soon
  // as they are compacted (raises what device_results() would raise about the input), the less-flat cloud when its voxel grid is done —
  // the odometry's iterations run in between (OdometryBatch::process_linked)
  bool split_results() const { return split_; }
  void device_results_front(uint32_t sweep, const float4* ptr[3], uint32_t count[3]);
  void device_results_lf(uint32_t sweep, const float4*& ptr, uint32_t& count);

  // device-side results for chaining (valid after run_async on the same stream):
  //   kind 0 sharp, 1 less_sharp, 2 flat: compact arrays + per-sweep offsets [nsw+1]
  //   less_flat: voxel-filtered per ring; d_less_flat_ring_off [total_rings+1]; sweep s owns rings [ring_base(s), ring_base(s+1))
  const float4* d_feat(int kind) const { return out_[kind].p; }
  const uint32_t* d_feat_off(int kind) const { return offs_.p + (size_t)kind * off_stride_; }
  // all four offset tables back to back: [kind][n_sweeps + 1] x 3, then the less-flat ring offsets [total_rings + 1]
  const uint32_t* d_offsets() const { return offs_.p; }
  uint32_t n_offsets() const { return 3 * off_stride_ + nring_ + 1; }
  const float4* d_less_flat() const { return lf_out_.p; }
  const uint32_t* d_less_flat_ring_off() const { return offs_.p + (size_t)3 * off_stride_; }
  const float4* d_cloud() const { return cloud_.p; }
  uint32_t n_points() const { return n_; }
  uint32_t n_sweeps() const { return nsw_; }
  uint32_t ring_base(uint32_t s) const { return h_ring_base_[s]; }
  uint32_t point_base(uint32_t s) const { return h_pt_base_[s]; }
  uint32_t total_rings() const { return nring_; }

 private:
  ImuTracker imu_;
  DevBuf<double> imu_dt_, imu_dstamp_;
  DevBuf<float> imu_state_;
  DevBuf<ImuLast> imu_last_;
  PinBuf<double> h_imu_d_;
  PinBuf<float> h_imu_f_;
  PinBuf<ImuLast> h_imu_last_;
  void check_params_() const;
  void layout_(uint32_t nsw, const uint32_t* const* ring_size, const uint32_t* n_rings);
  void allocate_(hipStream_t table_stream = nullptr);
  PinBuf<uint32_t> h_tab_;
  RawBinner binner_;
  DevBuf<float4> raw_;
  DevBuf<uint32_t> raw_ring_cnt_;
  PinBuf<uint32_t> h_raw_ring_cnt_;
  int device_;
  hipStream_t st_ = nullptr;
  bool own_stream_ = false;
  uint32_t nsw_ = 0, n_ = 0, nring_ = 0, max_ring_len_ = 0;
  std::vector<uint32_t> h_ring_off_, h_ring_base_, h_pt_base_, h_ring_sweep_base_;
  PinBuf<float4> h_cloud_;
  DevBuf<float4> cloud_;
  // the layout tables: ring_off_[nring+1] global point offsets | sweep base offset per ring | first ring of every sweep — views into ONE
  // device block (tab_dev_), uploaded in one copy and only when the layout differs from the one the block holds (tab_last_)
  struct TabView { uint32_t* p = nullptr; };
  TabView ring_off_, ring_sweep_base_, sweep_ring_base_;
  DevBuf<uint32_t> tab_dev_;
  std::vector<uint32_t> tab_last_;
  DevBuf<uint8_t> lf_valid_;   // (curvature, masks and gaps live in k_feat_ring's LDS since round 5)
  DevBuf<float4> slots_[3];       // per-ring fixed-capacity pick slots (sharp / less sharp / flat)
  DevBuf<uint32_t> slot_cnt_[3];  // per-ring counts
  DevBuf<float4> out_[3];         // the compact sharp / less-sharp / flat clouds (k_feat_compact)
  DevBuf<uint32_t> offs_;   // offs_: [3][nsw + 1] per-sweep offsets | [nring + 1] less-flat ring offsets
  uint32_t off_stride_ = 1;
  uint32_t* lf_off_() const { return offs_.p + (size_t)3 * off_stride_; }
  DevBuf<float4> lf_out_, lf_slots_;
  DevBuf<uint32_t> lf_cnt_;
  VoxelPipeline vox_;
  PinBuf<uint32_t> h_off_, h_link_off_;
  PinBuf<uint32_t> h_mirror_off_;   // run_async(true): the offset tables as the compaction kernels wrote them (layout of offs_)
  hipEvent_t ev_front_ = nullptr, ev_lf_ = nullptr;   // behind k_feat_compact / behind k_feat_lf_compact
  bool split_ = false;
  PinBuf<uint32_t> h_bad_;   // pinned word raised by k_feat_point on a non-finite input coordinate
  PinBuf<float4> h_pack_;   // download(): [header | sharp | less sharp | flat | less flat] of one sweep, written by k_feat_pack_host
};

}  // namespace loamx
