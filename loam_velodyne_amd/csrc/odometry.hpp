// Sweep-to-sweep registration (BasicLaserOdometry) for n independent streams; loamx_odom_* is the 1-stream case.
#pragma once
#include "common.h"
#include "host_math.h"
#include "registration.hpp"
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

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
  unsigned ticket;        // (unused; kept zeroed)
  unsigned xchg_epoch;    // k_odom_lm: number of this sweep (24 bits, never 0) in the tags of the records its workgroups exchange
  double* part;           // k_odom_lm: [2][16][LX_NSUM] partial normal equations of the stream's workgroups, one tagged 16-byte record each
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
  // how many further process() calls the handed-on clouds (d_last_corner / d_last_surf) survive: 2 by default, up to MAX_KEEP
  void set_keep(int n) { keep_ = n < 2 ? 2 : (n > MAX_KEEP ? MAX_KEEP : n); }
  // host-cloud convenience (single-stream handles)
  int process_host(const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat, const loamx_cloud* less_flat);
  // the same sweep from DEVICE clouds (the feature extractor's own buffers, read until this call returns), plus the sweep's
  // full-resolution cloud: re-projected to the sweep end (LaserOdometry.cpp:326) into a buffer of this object behind the tail.
  // link_ready() is recorded behind everything a consumer on another stream reads: d_last_corner / d_last_surf / d_link_full
  // less_flat_late (optional): the less-flat cloud is asked for only when the tail needs it — the callback blocks until it exists and
  // returns its device view and size (a subset of the sweep); the iterations start without it (process(): late_less_flat)
  int process_linked(const float4* const feat[4], const uint32_t n_feat[4], const float4* d_full, uint32_t n_full,
                     const std::function<void(const float4*&, uint32_t&)>& less_flat_late = nullptr);
  hipEvent_t link_ready() const { return ev_link_; }
  const float4* d_link_full() const { return link_full_.p; }
  uint32_t n_link_full() const { return n_link_full_; }
  bool link_valid() const { return link_valid_; }
  int device() const { return device_; }
  int get_last_clouds(uint32_t s, loamx_cloud* corner, loamx_cloud* surf);
  int transform_to_end_host(uint32_t s, loamx_cloud* cloud);
  // in place on device points with stream s's current transform (async on the stream)
  void to_end_device(uint32_t s, float4* pts, uint32_t n);
  // dst[h_off[k] .. h_off[k+1]) = transformToEnd(src[k]) with seg_params[k]; one launch on `stream`
  void to_end_gather(float4* dst, const uint32_t* h_off, const float4* const* src, const ToEndParams* seg_params, uint32_t K,
                     hipStream_t stream);
  ToEndParams to_end_params(uint32_t s, bool enabled) const;
  // Launch timing (measurement only; bench.py's roofline / latency model of the odometry pair): while on, every process() call brackets
  // its k_odom_corr_grid and k_odom_lm launches with HIP events on this object's stream; the elapsed times are taken lazily, once a
  // call's events have completed, and added up together with what the host knows of every launch afterwards: how many iterations its
  // slowest stream ran in it (0: a launch over converged streams) and the algorithmic bytes of the streams that were still iterating.
  struct LaunchTotals {
    double lm_ms = 0, lm_noop_ms = 0, corr_ms = 0, corr_noop_ms = 0;   // launches that iterated / that found every stream converged
    uint64_t lm_launches = 0, lm_noop_launches = 0, lm_iterations = 0, corr_launches = 0, corr_noop_launches = 0;
    uint64_t lm_bytes = 0, corr_features = 0;                          // 48 B x features of the iterating streams; features searched
  };
  void set_launch_timing(bool on) { launch_timing_.store(on, std::memory_order_relaxed); }
  LaunchTotals launch_totals();   // resolves what has completed and returns the running totals (never waits)
  // device views of the re-projected clouds handed on to mapping
  const float4* d_last_corner(uint32_t s) const { return last_.p + h_last_off_[s]; }
  const float4* d_last_surf(uint32_t s) const { return last_.p + h_last_off_[n_streams() + s]; }

 private:
  int device_;
  hipStream_t st_ = nullptr;
  bool own_stream_ = false;
  std::vector<OdomStream*> streams_;
  // clouds of all streams, concatenated: [corner_0 .. corner_{ns-1} | surf_0 .. surf_{ns-1}], offsets 2*ns+1
  DevBuf<float4> cur_, last_;          // being written | handed on by the last call
  static constexpr int MAX_KEEP = 16;
  DevBuf<float4> older_[MAX_KEEP - 1];  // handed on by the calls before (still read by their consumers): older_[0] the most recent
  int keep_ = 2;                       // clouds handed on stay valid during the next keep_ calls (set_keep)
  std::vector<uint32_t> h_cur_off_, h_last_off_;
  SubMapIndexBatch index_;
  DevBuf<int> ind_;
  DevBuf<double> part_;
  PinBuf<OdomProblem> h_mirror_;
  PinBuf<uint32_t> h_err_;
  DevBuf<uint32_t> rf_;   // [2 * n_streams][OD_RF_N] ring-first tables of the clouds handed on by the last call (entries tagged with rf_epoch_)
  uint32_t rf_epoch_ = 0;
  uint32_t xchg_epoch_ = 0;   // process() calls so far, 1 .. 2^24 - 1 (OdomProblem::xchg_epoch)
  // what a call sends up before its first kernel — the problems, the re-projection parameters, the cloud offsets — is ONE block in
  // pinned memory and ONE copy (three copies were three ~7 us commands at the head of the odometry chain, the pipeline's longest)
  template <class T> struct View { T* p = nullptr; };
  DevBuf<char> up_dev_;
  PinBuf<char> up_host_;
  size_t up_bytes_ = 0;
  View<OdomProblem> prob_, h_prob_;
  View<ToEndParams> te_, h_te_;
  View<uint32_t> d_cur_off_, h_off_pin_;
  PinBuf<uint32_t> h_off_late_;      // the cloud offsets once more, uploaded when a late less-flat cloud has arrived
  std::function<void(const float4*&, uint32_t&)> late_less_flat;   // single-stream process() only; set and cleared by process_linked
  uint32_t late_bound = 0;           // upper bound of that cloud's size
  PinBuf<float4> h_stage_;
  PinBuf<float4> h_last_dl_;     // process_host(): the clouds get_last_clouds() hands out, copied behind the tail
  bool last_dl_valid_ = false;
  DevBuf<float4> up_[4], tmp_cloud_, link_full_;
  uint32_t n_link_full_ = 0;
  bool link_valid_ = false;
  hipEvent_t ev_link_ = nullptr;
  int pred_pairs_ = 5;               // launch pairs the slowest stream of the previous sweep needed (LOAMX_ODOM_PAIRS=exact)
  uint64_t pairs_enqueued_ = 0, pair_calls_ = 0;   // launch pairs enqueued / sweeps with iterations, since creation
  uint32_t lm_slots_[2] = {0, 0};   // workgroups of k_odom_lm<1> / <2> the device holds at once (occupancy x CUs)
  hipEvent_t ev_tail_ = nullptr, ev_pose_ = nullptr, ev_up_ = nullptr;
  bool tail_pending_ = false, up_pending_ = false;
  PinBuf<char> h_gather_;
  DevBuf<char> d_gather_;
  // launch timing: a small ring of per-call event sets
  static constexpr int LT_RING = 8, LT_MAXP = 8;   // calls in flight, launch pairs per call
  struct LtCall {
    hipEvent_t ev[3 * LT_MAXP] = {};   // pair k: before corr, between corr and lm, behind lm
    int pairs = 0;
    bool pending = false;
    int iters[LT_MAXP] = {};           // iterations of the slowest stream in pair k's k_odom_lm launch
    uint64_t bytes[LT_MAXP] = {}, feats[LT_MAXP] = {};
  };
  LtCall lt_[LT_RING];
  int lt_next_ = 0;
  std::atomic<bool> launch_timing_{false};
  std::mutex lt_mu_;   // (the totals are read by the calling thread while a chain's worker thread is inside process())
  LaunchTotals lt_tot_;
  void lt_resolve_();
};


}  // namespace loamx
