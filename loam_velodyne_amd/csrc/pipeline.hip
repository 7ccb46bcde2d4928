// Batched streaming pipeline: n independent streams, each advancing one sweep per step through
//   feature extraction (BasicScanRegistration) -> odometry (BasicLaserOdometry) -> registration against a frozen
//   sub-map (BasicLaserMapping::optimizeTransformTobeMapped, transformAssociateToMap, transformFullResToMap).
// This is the batched-sweep mode of BASELINE.json's north_star: the streams are the independent units that are sharded
// across GPUs; every stream keeps the reference's sequential semantics (its odometry state, its
// transformBefMapped / transformAftMapped), only the map is frozen for the epoch.  All device work of a step runs on one
// HIP stream with inputs resident in HBM; the host touches only offsets and 6-float poses between the stages.
#include "pinned_copy.hpp"
#include "features.hpp"
#include "odometry.hpp"
#include "registration.hpp"
#include "hostlink.hpp"
#include <atomic>
#include <deque>
#include <functional>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <memory>

namespace loamx {

// what odometry hands on to the registration of the same sweep
struct OdomPub {
  HTwist transform, transform_sum;
  OdomStats stats = {0, 0, 0, 0};
  int rc = LOAMX_SKIPPED;
  const float4* last_corner = nullptr; uint32_t n_last_corner = 0;
  const float4* last_surf = nullptr; uint32_t n_last_surf = 0;
  ToEndParams to_end;
};

struct PipeStreamState {
  HTwist bef, aft, tobe, incre;   // mapping-side transforms (transformSum comes from the odometry stream)
  OdomPub cur;                    // odometry results of the step being registered (the look-ahead's wait in Pipeline::ores)
  SweepStats map_stats = {0, 0, 0, 0, 0, 0, 0, 0};
  bool mapped = false;
};

class Pipeline {
 public:
  Pipeline(const loamx_scanreg_config& fc, const loamx_odom_config& oc, const loamx_map_config& mc, uint32_t n_streams)
      : reg(mc.device, n_streams), fcfg(fc), n_streams_(n_streams), st(n_streams), imu(n_streams), map_imu(n_streams), map_imu_seq(n_streams, 0) {
    // The odometry of the streams runs as G independent chains ("groups"), each with its own OdometryBatch, HIP stream and host thread:
    // a stream whose sweep needs 25 iterations (BasicLaserOdometry.cpp:246 runs to maxIterations when the stop test at :613-620 never
    // fires) delays only its own group, and the two steps of look-ahead absorb it — with ONE chain every step paid the launch pairs of
    // the slowest of all streams (profiles/r03: 5.2 pairs per step for a mean of 7.1 iterations).  Default 2: measured on MI355X
    // (profiles/r04_groups_ab.md) 1 -> 2 chains gains 8 %, but with 4 (six busy HIP streams in the process) EVERY chain slows down
    // (registration 0.55 -> 0.88 ms, k_gn_iter 64 -> 93 us) and the step is 35 % slower than with one.
    // Round 4 / 5, measured and removed: an engine that let EVERY stream run at its own pace inside shared launches (a per-stream state
    // machine on the device).  Bit-identical, but slower on the bench workload even with six steps of look-ahead and its thread off
    // the runtime's locks (16.3 k sweeps/s against the chains' 17.9 k, profiles/r05_ab.md): the code is in the history (odom_engine.inc).
    OdomParams op;
    op.scan_period = oc.scan_period;
    op.max_iterations = diag_env("LOAMX_ODOM_MAXIT") ? atoi(diag_env("LOAMX_ODOM_MAXIT")) : oc.max_iterations;   // (diagnostic override, LOAMX_DIAG builds only)
    if (op.max_iterations < 1 || op.max_iterations > 255) throw Error(LOAMX_E_INVALID, "odometry max_iterations must be in [1, 255]");   // (k_odom_lm tags its exchange records with iteration + 1 in 8 bits)
    op.delta_t_abort = oc.delta_t_abort;
    op.delta_r_abort = oc.delta_r_abort;
    {
      int g = (int)std::min<uint32_t>(n_streams, 2);
      if (const char* e = diag_env("LOAMX_ODOM_GROUPS")) g = atoi(e);
      n_groups = (uint32_t)std::max(1, std::min(g, (int)std::min<uint32_t>(n_streams, MAX_GROUPS)));
    }
    for (uint32_t g = 0; g < n_groups; g++) {
      chains.emplace_back(new OdomChain());
      OdomChain& c = *chains.back();
      c.s0 = (uint32_t)((uint64_t)g * n_streams / n_groups);
      c.s1 = (uint32_t)((uint64_t)(g + 1) * n_streams / n_groups);
      c.ob.reset(new OdometryBatch(mc.device, c.s1 - c.s0, nullptr));
    }
    for (auto& o : ores) o.resize(n_streams);
    for (auto& tr : imu) tr.history_size = std::max(200, fc.imu_history_size);
    reg.params.max_iterations = mc.max_iterations;
    reg.early_exit = true;   // step() blocks on M(t) anyway
    reg.double_buffer_full = true;   // two full-resolution staging buffers: the next step's re-projected clouds are written while this step's are registered (prestage_gather)
    reg.params.delta_t_abort = mc.delta_t_abort;
    reg.params.delta_r_abort = mc.delta_r_abort;
    reg.params.corner_leaf = mc.corner_filter_size;
    reg.params.surf_leaf = mc.surf_filter_size;
    for (auto& c : chains) { c->ob->params = op; c->ob->set_keep(OR - 2); }
    device = mc.device;
    if (getenv("LOAMX_NO_LOOKAHEAD")) prefetch = false;   // debugging / profiling: run the stages one after the other
  }
  Registrar reg;
  // Look-ahead rings: what a look-ahead step leaves behind for its registration (feature offsets, odometry results, events, timers)
  // lives in slot step % OR, so the odometry may run up to OR - 2 steps ahead of the registration (ahead_depth below)
  static constexpr int OR = 16;
  // one odometry chain: the streams [s0, s1), their batch object (own HIP stream), the host thread that drives it and its position
  struct OdomChain {
    uint32_t s0 = 0, s1 = 0;
    std::unique_ptr<OdometryBatch> ob;
    std::thread worker;
    std::atomic<int> done{-1};           // odometry of steps <= done is complete and published
    std::atomic<int> next{0};            // next step of this chain (its worker; the calling thread only while the workers are parked)
    std::atomic<bool> busy{false};
    hipEvent_t ev_tail[OR] = {};   // recorded behind O(k)'s tail (re-projection + index build) on the chain's stream
    hipEvent_t tm_a[OR] = {}, tm_b[OR] = {};   // chain timer of step k % OR (this thread's own)
    bool tm_pending[OR] = {};
    std::atomic<float> ms{0.f};          // length of the chain's most recent timed pass on its HIP stream
    double tr[4] = {0, 0, 0, 0};
    std::atomic<float> last_us{0.f};     // host time of the chain's most recent pass (LOAMX_PIPE_TRACE)
  };
  static constexpr uint32_t MAX_GROUPS = 16;
  uint32_t n_groups = 1;
  std::vector<std::unique_ptr<OdomChain>> chains;
  OdomChain& chain_of(uint32_t s) { uint32_t g = 0; while (g + 1 < n_groups && s >= chains[g]->s1) g++; return *chains[g]; }
  OdometryBatch& OB(uint32_t s) { return *chain_of(s).ob; }
  uint32_t LS(uint32_t s) { return s - chain_of(s).s0; }   // index of stream s inside its group
  loamx_scanreg_config fcfg;
  uint32_t n_streams_;
  int device;
  std::vector<PipeStreamState> st;
  std::vector<std::unique_ptr<FeatureExtractor>> fx;   // one staged batch per step (upload) or a ring of RING slots (stage_step)
  // Streaming input (loamx_pipeline_stage_step): step t lives in slot t % RING; steps t .. t + RING - 2 may be in flight while
  // t + RING - 1 is being staged.  The copies run on a stream of their own; launch_features() orders the extraction behind their event.
  static constexpr uint32_t RING = 8;
  bool streaming = false;
  // streaming: steps below staged_hi have been staged.  stage_step* may run on ONE other thread than step(): it publishes the slot
  // (release) after everything of it is enqueued, step() reads the count once (acquire); the two never touch the same slot — the
  // stager works on the slot of step t + RING - 1 at most while steps <= t + RING - 2 are in flight
  std::atomic<uint32_t> staged_hi{0};
  hipStream_t cstream = nullptr, dstream = nullptr;      // H2D staging / D2H of the registered clouds
  hipEvent_t ev_stage[RING] = {};
  hipEvent_t ev_reg_done = nullptr, ev_d2h[2] = {nullptr, nullptr};
  bool d2h_pending[2] = {false, false};
  uint64_t downloads_direct = 0, downloads_hip = 0;      // asynchronous downloads issued to the SDMA engine directly / through hipMemcpyAsync
  HostLinkDma hostlink;                                   // the same downloads on the SDMA engine directly (hostlink.hpp)
  HostLinkUp uplink;                                      // ... and the staging copies of stage_step (slot = t % RING); LOAMX_H2D_DIRECT=0: through HIP
  uint32_t up_runs[RING] = {};                            // block copies of the slot's step handed to uplink (0: none, the HIP copy stream carried them)
  std::atomic<long> last_step{-1};                        // the last step that has run (-1: none yet)
  std::vector<uint32_t> last_full_off;                   // offsets of the registered clouds of the last step (k-th mapped stream)
  // Raw input (loamx_pipeline_stage_step_raw): per slot the payloads, their binned clouds and what the binning leaves behind
  struct RawSlot {
    bool raw = false, finalized = true;
    std::vector<uint32_t> count, off;            // raw points per stream, their offsets in the slot's arrays
    DevBuf<char> d_bytes;
    DevBuf<float4> d_raw4, d_binned;
    DevBuf<uint32_t> d_ring_cnt;                 // [stream][n_rings]
    PinBuf<uint32_t> h_ring_cnt;
    DevBuf<ImuLast> d_last;
    PinBuf<ImuLast> h_last;
    DevBuf<double> d_imu_d;
    DevBuf<float> d_imu_f;
    PinBuf<double> h_imu_d;
    PinBuf<float> h_imu_f;
    std::vector<uint32_t> imu_H;
    std::vector<float> imu_trans;                // [stream][12], valid once finalized
    std::vector<double> scan_time;               // [stream] the sweeps' time stamps (laserOdometryTime of the mapping-side blend); empty: none given
    std::vector<uint64_t> map_imu_upto;          // [stream] mapping-side IMU messages that had arrived at staging time
    hipEvent_t ev_ingest = nullptr;
    uint32_t n_rings = 0;
  };
  RawSlot rawslot[RING];
  std::vector<ImuTracker> imu;                   // one IMU state machine per stream
  // Mapping-side IMU history (LaserMapping's own /imu/data subscription: IMUState2 {stamp, roll, pitch}, 200 deep —
  // BasicLaserMapping.cpp:602-605) per stream, for transformUpdate's roll / pitch blend (:171-200).  Messages are numbered: a sweep is
  // blended with the messages that had arrived when it was STAGED (stage_step_raw), however far the staging runs ahead of its step.
  struct MapImu { double stamp; float roll, pitch; uint64_t seq; };
  std::vector<std::deque<MapImu>> map_imu;
  std::vector<uint64_t> map_imu_seq;
  std::mutex map_imu_mu;                         // (update_imu may come from the staging thread)
  void map_imu_push(uint32_t s, double stamp, float roll, float pitch) {
    std::lock_guard<std::mutex> lk(map_imu_mu);
    if (map_imu[s].size() >= 200) map_imu[s].pop_front();
    map_imu[s].push_back(MapImu{stamp, roll, pitch, map_imu_seq[s]++});
  }
  // the IMU part of transformUpdate (:173-200) for one stream's pose; false: no message had arrived when the sweep was staged
  bool map_imu_blend(uint32_t s, uint64_t seq_limit, double odo_time, float scan_period, float* p6) {
    std::lock_guard<std::mutex> lk(map_imu_mu);
    const std::deque<MapImu>& Hq = map_imu[s];
    size_t n = 0;
    while (n < Hq.size() && Hq[n].seq < seq_limit) n++;
    if (!n) return false;
    size_t i = 0;
    while (i < n - 1 && (odo_time - Hq[i].stamp) + scan_period > 0) i++;
    float roll, pitch;
    if (i == 0 || (odo_time - Hq[i].stamp) + scan_period > 0) {
      roll = Hq[i].roll; pitch = Hq[i].pitch;   // scan time newer than the newest or older than the oldest IMU message
    } else {
      const float ratio = (float)(((Hq[i].stamp - odo_time) - scan_period) / (Hq[i].stamp - Hq[i - 1].stamp));
      const float inv = 1 - ratio;
      roll = Hq[i].roll * inv + Hq[i - 1].roll * ratio;
      pitch = Hq[i].pitch * inv + Hq[i - 1].pitch * ratio;
    }
    p6[0] = (float)(0.998 * p6[0] + 0.002 * pitch);
    p6[2] = (float)(0.998 * p6[2] + 0.002 * roll);
    return true;
  }
  RawBinner binner;
  FeatureExtractor& FX(uint32_t t) { return *fx[streaming ? t % RING : t]; }
  char& LA(uint32_t t) { return launched[streaming ? t % RING : t]; }
  uint32_t n_staged() const { return streaming ? staged_hi.load(std::memory_order_acquire) : (uint32_t)fx.size(); }
  // feature extraction of step t+1 is independent of odometry / registration of step t (in the reference they are
  // different ROS nodes): it runs on its own HIP stream, launched one step ahead, and overlaps with them
  hipStream_t fstream = nullptr;
  bool prefetch = true;
  std::vector<char> launched;
  PinBuf<uint32_t> h_off3[OR];
  hipEvent_t evF[OR][2] = {};
  // stage timers are read lazily (an elapsed time is taken once both events have completed) so that measuring never
  // makes the host wait for a stage
  struct LazyTimer {
    hipEvent_t a = nullptr, b = nullptr;
    bool pending = false;
    float ms = 0.f;
    void create() { if (!a) { LX_HIP(hipEventCreate(&a)); LX_HIP(hipEventCreate(&b)); } }
    void resolve() {
      if (pending && hipEventQuery(b) == hipSuccess) { LX_HIP(hipEventElapsedTime(&ms, a, b)); pending = false; }
    }
    void destroy() { if (a) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); a = b = nullptr; } }
  };
  LazyTimer tmM[2];   // registration: by step parity (the odometry chains keep their own timers, OdomChain)
  float feat_ms[OR] = {};
  int f_hi = -1;                       // features of steps <= f_hi have been launched (calling thread)
  // Steps the odometry chains may run ahead of the registration.  A stream that needs all 25 iterations needs them for five or six
  // sweeps in a row (a chain then takes ~570 us per step against the registration's ~430): on average the chains keep up, and a deeper
  // look-ahead lets them build the lead that such a run eats (depth 2: 15.6 k sweeps/s, 4: 16.6 k, 6: 16.9 k, 8 - 12: 16.4 - 16.6 k;
  // profiles/r04_ab.md).  The streaming ring holds RING = 8 slots: the same six steps.
  int ahead_depth = std::max(1, std::min(diag_env("LOAMX_ODOM_AHEAD") ? atoi(diag_env("LOAMX_ODOM_AHEAD")) : 6, OR - 2));
  int depth() const { return !prefetch ? 0 : (streaming ? std::min(ahead_depth, (int)RING - 2) : ahead_depth); }
  float last_ms[4] = {0, 0, 0, 0};
  std::atomic<bool> timing{false};

  // The odometry chains run on persistent host threads of their own and AHEAD of the registration: odometry O(k) of a stream only
  // depends on its O(k-1) and on the features F(k), never on a registration (separate ROS nodes in the reference) and never on another
  // stream, so a chain goes on to O(k+1) as soon as its O(k) is done, up to depth() steps ahead of the step being registered —
  // registration and odometry are serial chains across steps and the slowest sets the pace, not their sum.  The calling thread launches
  // the features (the only thread that does) and raises o_limit; every chain publishes its own `done`.  Results wait in a ring of
  // OR slots, the re-projected clouds in rotating buffers (OdometryBatch::set_keep), so step k's inputs stay valid while O(k+1) ...
  // O(k+depth) run.  Hand-overs happen every ~0.4 ms, so both sides spin briefly before they fall back to the condition variable (a
  // sleeping thread costs tens of microseconds to wake, on the critical path of every step).
  std::vector<OdomPub> ores[OR];       // [step % OR][stream]; a chain writes its own streams only
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<int> o_limit{-1};        // the chains may run steps <= o_limit (raised by the calling thread only)
  bool quit = false;
  std::exception_ptr job_err;
  int done_min() const { int m = INT32_MAX; for (auto& c : chains) m = std::min(m, c->done.load(std::memory_order_acquire)); return m; }
  int done_max() const { int m = -1; for (auto& c : chains) m = std::max(m, c->done.load(std::memory_order_acquire)); return m; }
  bool any_busy() const { for (auto& c : chains) if (c->busy.load(std::memory_order_acquire)) return true; return false; }
  static bool spin_until(const std::function<bool()>& ready, double max_us) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!ready()) {
      if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > max_us) return false;
      __builtin_ia32_pause();
    }
    return true;
  }
  void worker_main(OdomChain* cp) {
    OdomChain& c = *cp;
    (void)hipSetDevice(device);
    for (;;) {
      auto ready = [&] { return c.next.load(std::memory_order_acquire) <= o_limit.load(std::memory_order_acquire); };
      static const double spin_us = diag_env("LOAMX_SPIN_US") ? atof(diag_env("LOAMX_SPIN_US")) : 400.0;   // diagnostic
      if (!spin_until(ready, spin_us)) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ready() || quit; });
        if (quit) return;
      }
      {
        std::lock_guard<std::mutex> lk(mu);   // (busy and o_limit change under the mutex: park_odometry() relies on seeing them together)
        if (quit) return;
        if (!ready()) continue;
        c.busy.store(true, std::memory_order_release);
      }
      std::exception_ptr err;
      const int k = c.next.load(std::memory_order_acquire);
      try {
        const auto tc0 = std::chrono::steady_clock::now();
        c.tr[0] = tr_us(); run_odometry(c, (uint32_t)k); c.tr[3] = tr_us();
        c.last_us.store((float)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tc0).count(), std::memory_order_relaxed);
      } catch (...) { err = std::current_exception(); }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (err) { if (!job_err) job_err = err; o_limit.store(-1, std::memory_order_release); }   // every chain stops; the calling thread rethrows
        else { c.next.store(k + 1, std::memory_order_release); c.done.store(k, std::memory_order_release); }
        c.busy.store(false, std::memory_order_release);
      }
      cv.notify_all();
    }
  }
  // calling thread: allow the odometry chains to run up to step k
  void allow_odometry(int k) {
    if (k <= o_limit.load(std::memory_order_acquire)) return;
    for (auto& c : chains)
      if (!c->worker.joinable()) { OdomChain* cp = c.get(); c->worker = std::thread([this, cp] { worker_main(cp); }); }
    { std::lock_guard<std::mutex> lk(mu); o_limit.store(k, std::memory_order_release); }
    cv.notify_all();
  }
  // calling thread: block until O(t) of every chain is published (rethrows a failure of a worker)
  void wait_odometry(int t) {
    auto ready = [&] { return done_min() >= t || (!any_busy() && o_limit.load(std::memory_order_acquire) < t); };
    if (!spin_until(ready, 2000.0)) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, ready);
    }
    std::lock_guard<std::mutex> lk(mu);
    if (job_err) { std::exception_ptr e = job_err; job_err = nullptr; std::rethrow_exception(e); }
    LX_REQUIRE(done_min() >= t, "internal: an odometry chain stopped before the requested step");
  }
  // calling thread: block until the look-ahead has finished every step it has been allowed to run (its kernels are enqueued then:
  // a device synchronisation afterwards covers them).  Returns the last step whose odometry is published for every stream, -1 if none.
  int drain_lookahead() {
    const int lim = o_limit.load(std::memory_order_acquire);
    if (prefetch && lim >= 0 && chains[0]->worker.joinable()) wait_odometry(lim);
    return done_min();
  }
  // calling thread: stop the look-ahead and wait until every worker is idle.  restart >= 0: chains that have not reached that step
  // continue there (the caller jumped); restart < 0: every chain continues where it IS — the positions are read AFTER the workers
  // have gone idle, under the mutex (a position read before the wait is stale by the step a worker was inside: ADVICE.md round 3)
  void park_odometry(int restart) {
    if (restart >= 0) pre_t = -1;   // (a pre-staged re-projection belongs to the run that is abandoned)
    std::unique_lock<std::mutex> lk(mu);
    o_limit.store(-1, std::memory_order_release);
    cv.wait(lk, [&] { return !any_busy(); });
    if (restart >= 0)
      for (auto& c : chains)
        if (restart > c->done.load() && restart != c->next.load()) {
          c->next.store(restart, std::memory_order_release);
          c->done.store(restart - 1, std::memory_order_release);
        }
    job_err = nullptr;
  }
  // (upload / first use of the streaming ring: every chain starts over at step 0)
  void reset_odometry() {
    std::unique_lock<std::mutex> lk(mu);
    o_limit.store(-1, std::memory_order_release);
    cv.wait(lk, [&] { return !any_busy(); });
    for (auto& c : chains) { c->next.store(0, std::memory_order_release); c->done.store(-1, std::memory_order_release); }
    job_err = nullptr;
  }

  ~Pipeline() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; o_limit.store(-1, std::memory_order_release); }
    cv.notify_all();   // (a worker leaves its spin phase after 0.4 ms and then sees quit)
    for (auto& c : chains) if (c->worker.joinable()) c->worker.join();
    for (auto& c : chains) {
      for (auto& e : c->ev_tail) if (e) (void)hipEventDestroy(e);
      for (auto& e : c->tm_a) if (e) (void)hipEventDestroy(e);
      for (auto& e : c->tm_b) if (e) (void)hipEventDestroy(e);
    }
    fx.clear();
    for (auto& a : evF) for (auto& e : a) if (e) (void)hipEventDestroy(e);
    for (auto& tm : tmM) tm.destroy();
    for (auto& e : ev_stage) if (e) (void)hipEventDestroy(e);
    for (auto& r : rawslot) if (r.ev_ingest) (void)hipEventDestroy(r.ev_ingest);
    for (auto& e : ev_d2h) if (e) (void)hipEventDestroy(e);
    if (ev_reg_done) (void)hipEventDestroy(ev_reg_done);
    if (cstream) { (void)hipStreamSynchronize(cstream); (void)hipStreamDestroy(cstream); }
    if (dstream) { (void)hipStreamSynchronize(dstream); (void)hipStreamDestroy(dstream); }
    if (fstream) (void)hipStreamDestroy(fstream);
  }

  void launch_features(uint32_t t) {
    TraceRange trace_range("loamx:features");
    finalize_raw(t);
    FeatureExtractor& F = FX(t);
    const uint32_t ns = n_streams_, nring = F.total_rings();
    if (streaming) {
      LX_HIP(hipStreamWaitEvent(fstream, ev_stage[t % RING], 0));   // this slot's H2D copies (the tables; the sweeps too unless ROCr carried them)
      if (up_runs[t % RING]) { uplink.wait((int)(t % RING)); up_runs[t % RING] = 0; }   // (issued a step or more ago)
    }
    PinBuf<uint32_t>& hb = h_off3[t % OR];
    hb.reserve(3 * (ns + 1) + nring + 2);
    uint32_t* ho[3] = {hb.p, hb.p + (ns + 1), hb.p + 2 * (ns + 1)};
    uint32_t* hlf = hb.p + 3 * (ns + 1);
    for (auto& e : evF[t % OR]) if (!e) LX_HIP(hipEventCreate(&e));
    LX_HIP(hipEventRecord(evF[t % OR][0], fstream));
    F.run_async();
    // (the four offset tables lie back to back on the device in exactly this host layout: one copy)
    (void)ho; (void)hlf;
    LX_REQUIRE(F.n_offsets() == 3 * (ns + 1) + nring + 1, "internal: offset table layout");
    store_to_pinned_u32(hb.p, F.d_offsets(), F.n_offsets(), fstream);   // (by a kernel: a small device-to-host hipMemcpyAsync blocks its caller for milliseconds now and then — pinned_copy.hpp)
    LX_HIP(hipEventRecord(evF[t % OR][1], fstream));
    LA(t) = 1;
  }

  void upload(uint32_t n_steps, const loamx_cloud* clouds, const uint32_t* const* ring_size, const uint32_t* n_rings) {
    LX_REQUIRE(n_steps >= 1 && clouds && ring_size && n_rings, "invalid argument");
    LX_HIP(hipSetDevice(device));
    if (!fstream) fstream = create_stream(env_priority("LOAMX_PRIO_FEAT", -1), diag_env("LOAMX_FEAT_CU_STRIDE") ? atoi(diag_env("LOAMX_FEAT_CU_STRIDE")) : 0, /*part=*/1);
    LX_HIP(hipStreamSynchronize(fstream));
    fx.clear();
    streaming = false;
    staged_hi = 0;
    last_step = -1;
    launched.assign(n_steps, 0);
    reset_odometry();
    f_hi = -1;
    for (uint32_t t = 0; t < n_steps; t++) {
      auto f = std::make_unique<FeatureExtractor>(device, fstream);
      FeatParams& p = f->params;
      p.scan_period = fcfg.scan_period;
      p.n_regions = fcfg.n_feature_regions;
      p.curv_region = fcfg.curvature_region;
      p.max_sharp = fcfg.max_corner_sharp;
      p.max_less_sharp = fcfg.max_corner_less_sharp == 0 ? 10 * fcfg.max_corner_sharp : fcfg.max_corner_less_sharp;
      p.max_flat = fcfg.max_surface_flat;
      p.less_flat_leaf = fcfg.less_flat_filter_size;
      p.curv_thr = fcfg.surface_curvature_threshold;
      f->upload(n_streams_, clouds + (size_t)t * n_streams_, ring_size + (size_t)t * n_streams_, n_rings + (size_t)t * n_streams_);
      fx.push_back(std::move(f));
    }
  }

  void ensure_streaming_(uint32_t t) {
    if (!streaming) {   // first use: switch to the ring of slots
      LX_REQUIRE(t == 0, "streaming input starts at step 0");
      if (!fstream) fstream = create_stream(env_priority("LOAMX_PRIO_FEAT", -1), diag_env("LOAMX_FEAT_CU_STRIDE") ? atoi(diag_env("LOAMX_FEAT_CU_STRIDE")) : 0, /*part=*/1);
      LX_HIP(hipStreamSynchronize(fstream));
      fx.clear();
      for (uint32_t k = 0; k < RING; k++) {
        auto f = std::make_unique<FeatureExtractor>(device, fstream);
        FeatParams& p = f->params;
        p.scan_period = fcfg.scan_period;
        p.n_regions = fcfg.n_feature_regions;
        p.curv_region = fcfg.curvature_region;
        p.max_sharp = fcfg.max_corner_sharp;
        p.max_less_sharp = fcfg.max_corner_less_sharp == 0 ? 10 * fcfg.max_corner_sharp : fcfg.max_corner_less_sharp;
        p.max_flat = fcfg.max_surface_flat;
        p.less_flat_leaf = fcfg.less_flat_filter_size;
        p.curv_thr = fcfg.surface_curvature_threshold;
        fx.push_back(std::move(f));
      }
      launched.assign(RING, 0);
      if (!cstream) cstream = create_stream(0);
      for (auto& e : ev_stage) if (!e) LX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      streaming = true;
      staged_hi = 0;
      last_step = -1;
      reset_odometry();
      f_hi = -1;
    }
  }

  // Streaming input: stage ONE step (sweep s of the step = clouds[s]) without blocking; steps arrive in order.  Slot t % RING
  // is free once step t - RING has been registered, which the caller's own order of calls guarantees (stage(t) after step(t - RING)).
  void stage_step(uint32_t t, const loamx_cloud* clouds, const uint32_t* const* ring_size, const uint32_t* n_rings) {
    LX_REQUIRE(clouds && ring_size && n_rings, "invalid argument");
    LX_HIP(hipSetDevice(device));
    ensure_streaming_(t);
    LX_REQUIRE(t == staged_hi.load(), "steps must be staged in order");
    LX_REQUIRE(t < RING || last_step.load() + (long)RING >= (long)t, "stage_step(t) needs step(t - 8) to have run: only eight steps can be in flight");
    TraceRange trace_range("loamx:pipeline:stage_step");
    if (t > 0) finalize_raw(t - 1);
    rawslot[t % RING].raw = false;
    rawslot[t % RING].finalized = true;
    // The sweeps' block copies go to ROCr directly when source and destination are the runtime's own allocations (hostlink.hpp: a copy
    // stream of the HIP runtime would be a fifth busy HIP stream): the blocks are counted first (the group's size must be known when
    // it begins), then issued; launch_features() waits for the slot's signal on the host — a whole step later.
    static const bool direct = !(diag_env("LOAMX_H2D_DIRECT") && atoi(diag_env("LOAMX_H2D_DIRECT")) == 0);
    const int slot = (int)(t % RING);
    up_runs[slot] = 0;
    uint32_t n_blocks = 0;
    bool all_direct = direct;
    if (direct) {   // (the runs of clouds that lie back to back, as upload_async forms them)
      for (uint32_t s = 0; s < n_streams_ && all_direct;) {
        all_direct = clouds[s].stride == 16 && clouds[s].intensity_offset == 12;
        uint32_t e = s + 1;
        size_t cnt = clouds[s].count;
        while (all_direct && e < n_streams_ && (const char*)clouds[e].data == (const char*)clouds[s].data + sizeof(float4) * cnt) cnt += clouds[e++].count;
        if (cnt) { n_blocks++; all_direct = all_direct && uplink.host_ok(clouds[s].data, sizeof(float4) * cnt); }
        s = e;
      }
    }
    if (all_direct && n_blocks) {
      uplink.begin(slot, n_blocks);
      uint32_t issued = 0;
      try {
        fx[slot]->upload_async(n_streams_, clouds, ring_size, n_rings, cstream, ev_stage[slot], [&](void* dst, const void* src, size_t bytes) {
          if (!uplink.can_copy_h2d(dst, src, bytes)) throw Error(LOAMX_E_HIP, "internal: a staging block is not ROCr's memory after all");
          uplink.copy_h2d(slot, dst, src, bytes);
          issued++;
          return true;
        });
      } catch (...) {
        uplink.abandon(slot);
        throw;
      }
      up_runs[slot] = issued;
    } else {
      fx[slot]->upload_async(n_streams_, clouds, ring_size, n_rings, cstream, ev_stage[slot]);
    }
    LA(t) = 0;
    staged_hi.store(t + 1, std::memory_order_release);
  }

  // Raw input: step t as the sensor delivered it — per stream one revolution of (x, y, z) records in sensor axes and firing
  // order (the /velodyne_points payload, MultiScanRegistration.cpp:160-238).  The payloads cross PCIe as they are and are
  // re-strided, binned into rings and (with IMU data) de-skewed on the device, all on the copy stream; ring sizes and the IMU
  // state the projection loop leaves behind come back through pinned memory and are consumed by finalize_raw().
  void stage_step_raw(uint32_t t, const void* const* raw_xyz, const uint32_t* counts, uint32_t stride, const loamx_multiscan_mapper& mapper,
                      const double* scan_time) {
    LX_REQUIRE(raw_xyz && counts, "invalid argument");
    LX_REQUIRE(stride >= 12 && stride % 4 == 0, "raw stride must be a multiple of 4 and at least 12");
    LX_REQUIRE(mapper.n_scan_rings >= 1 && mapper.n_scan_rings <= RawBinner::MAX_RINGS, "n_scan_rings must be in [1, 256]");
    LX_REQUIRE(mapper.upper_bound_deg != mapper.lower_bound_deg, "vertical bounds must differ");
    LX_HIP(hipSetDevice(device));
    ensure_streaming_(t);
    LX_REQUIRE(t == staged_hi.load(), "steps must be staged in order");
    LX_REQUIRE(t < RING || last_step.load() + (long)RING >= (long)t, "stage_step_raw(t) needs step(t - 8) to have run: only eight steps can be in flight");
    TraceRange trace_range("loamx:pipeline:stage_step_raw");
    if (t > 0) finalize_raw(t - 1);   // the IMU state machine advances sweep by sweep: this step's table needs the previous reset
    const uint32_t ns = n_streams_, nr = mapper.n_scan_rings;
    RawSlot& R = rawslot[t % RING];
    R.raw = true;
    R.finalized = false;
    R.n_rings = nr;
    R.count.assign(counts, counts + ns);
    R.off.assign(ns + 1, 0);
    for (uint32_t s = 0; s < ns; s++) {
      LX_REQUIRE(raw_xyz[s] || counts[s] == 0, "NULL raw cloud");
      R.off[s + 1] = R.off[s] + counts[s];
    }
    const uint32_t ntot = R.off[ns];
    R.d_bytes.reserve((size_t)ntot * stride + 16);
    R.d_raw4.reserve((size_t)ntot + 1);
    R.d_binned.reserve((size_t)ntot + 1);
    R.d_ring_cnt.reserve((size_t)ns * nr + 1);
    R.h_ring_cnt.reserve((size_t)ns * nr + 1);
    R.d_last.reserve(ns);
    R.h_last.reserve(ns);
    R.imu_H.assign(ns, 0);
    R.imu_trans.assign((size_t)12 * ns, 0.f);
    if (scan_time) R.scan_time.assign(scan_time, scan_time + ns); else R.scan_time.clear();
    {
      std::lock_guard<std::mutex> lk(map_imu_mu);
      R.map_imu_upto = map_imu_seq;
    }
    if (!R.ev_ingest) LX_HIP(hipEventCreateWithFlags(&R.ev_ingest, hipEventDisableTiming));
    // IMU tables (one per stream with data), staged back to back
    size_t Htot = 0;
    for (uint32_t s = 0; s < ns; s++) { R.imu_H[s] = imu[s].size(); Htot += R.imu_H[s]; }
    if (Htot) {
      R.h_imu_d.reserve(2 * Htot); R.h_imu_f.reserve(9 * Htot); R.d_imu_d.reserve(2 * Htot); R.d_imu_f.reserve(9 * Htot);
    }
    MapperParams M;
    M.lower = mapper.lower_bound_deg; M.upper = mapper.upper_bound_deg; M.n_rings = nr;
    M.factor = (float)((int)nr - 1) / (mapper.upper_bound_deg - mapper.lower_bound_deg);   // MultiScanRegistration.cpp:41-50
    binner.init(cstream);
    LX_HIP(hipMemsetAsync(R.d_last.p, 0, sizeof(ImuLast) * ns, cstream));
    size_t hbase = 0;
    std::vector<ImuTable> tables(ns);
    for (uint32_t s = 0; s < ns; s++) {
      if (scan_time) imu[s].set_scan_time(scan_time[s]);
      const uint32_t H = R.imu_H[s];
      if (H) {
        imu[s].fill_table(R.h_imu_d.p + 2 * hbase, R.h_imu_f.p + 9 * hbase, tables[s]);
        tables[s].dt = R.d_imu_d.p + 2 * hbase;
        tables[s].dstamp = R.d_imu_d.p + 2 * hbase + H;
        tables[s].state = R.d_imu_f.p + 9 * hbase;
        hbase += H;
      }
    }
    if (Htot) {
      LX_HIP(hipMemcpyAsync(R.d_imu_d.p, R.h_imu_d.p, sizeof(double) * 2 * Htot, hipMemcpyHostToDevice, cstream));
      LX_HIP(hipMemcpyAsync(R.d_imu_f.p, R.h_imu_f.p, sizeof(float) * 9 * Htot, hipMemcpyHostToDevice, cstream));
    }
    for (uint32_t s = 0; s < ns; s++) {
      const uint32_t n = counts[s];
      if (n) LX_HIP(hipMemcpyAsync(R.d_bytes.p + (size_t)R.off[s] * stride, raw_xyz[s], (size_t)n * stride, hipMemcpyHostToDevice, cstream));
      raw_unpack(R.d_bytes.p + (size_t)R.off[s] * stride, stride, n, R.d_raw4.p + R.off[s], cstream);
      binner.run(R.d_raw4.p + R.off[s], n, M, fcfg.scan_period, R.d_binned.p + R.off[s], R.d_ring_cnt.p + (size_t)s * nr,
                 R.imu_H[s] ? &tables[s] : nullptr, R.imu_H[s] ? R.d_last.p + s : nullptr);
    }
    LX_HIP(hipMemcpyAsync(R.h_ring_cnt.p, R.d_ring_cnt.p, sizeof(uint32_t) * ns * nr, hipMemcpyDeviceToHost, cstream));
    LX_HIP(hipMemcpyAsync(R.h_last.p, R.d_last.p, sizeof(ImuLast) * ns, hipMemcpyDeviceToHost, cstream));
    LX_HIP(hipEventRecord(R.ev_ingest, cstream));
    LA(t) = 0;
    staged_hi.store(t + 1, std::memory_order_release);
  }

  // second half of a raw step's staging, once its binning has finished (long before it is needed in steady state): the IMU
  // state machines take over what the projection loops left behind and are reset for the next sweep (processScanlines:
  // reset(scanTime), updateIMUTransform), and the binned clouds are handed to the slot's feature extractor with their ring
  // sizes.  Idempotent; called in step order from stage_step*(t + 1) and launch_features(t), whichever comes first.
  std::mutex raw_mu;   // finalize_raw(t) is reached from the stager (stage_step*(t + 1)) and from the thread that launches the features of t
  void finalize_raw(uint32_t t) {
    if (!streaming) return;
    std::lock_guard<std::mutex> lk(raw_mu);
    RawSlot& R = rawslot[t % RING];
    if (!R.raw || R.finalized) return;
    LX_HIP(hipEventSynchronize(R.ev_ingest));
    const uint32_t ns = n_streams_, nr = R.n_rings;
    std::vector<const uint32_t*> rs(ns);
    std::vector<uint32_t> nrv(ns, nr);
    for (uint32_t s = 0; s < ns; s++) {
      if (R.imu_H[s] && R.h_last.p[s].valid) imu[s].apply_last(R.h_last.p[s]);
      imu[s].begin_sweep();
      memcpy(&R.imu_trans[12 * (size_t)s], imu[s].imu_trans(), sizeof(float) * 12);
      rs[s] = R.h_ring_cnt.p + (size_t)s * nr;
    }
    fx[t % RING]->upload_device(ns, R.d_binned.p, R.off.data(), rs.data(), nrv.data(), cstream, ev_stage[t % RING]);
    R.finalized = true;
  }

  // Registered full-resolution clouds of the step that just ran -> caller memory, asynchronously on a copy stream (the next
  // step's kernels do not wait for it: the registrar alternates between two full-resolution buffers).  Packed float4 records
  // only (a DMA cannot re-stride); out[k] receives the k-th stream that was registered.
  void download_step_async(loamx_cloud* out, uint32_t n_out) {
    LX_REQUIRE(out, "NULL argument");
    LX_REQUIRE(reg.double_buffer_full, "call loamx_pipeline_enable_async_downloads() before the first step");
    LX_HIP(hipSetDevice(device));
    const uint32_t nw = last_full_off.empty() ? 0u : (uint32_t)last_full_off.size() - 1;
    LX_REQUIRE(n_out >= nw, "fewer output descriptors than registered streams");
    const int par = (int)((run_count + 1) & 1);   // the buffer the last run used
    int rc_cap = LOAMX_OK;
    for (uint32_t k = 0; k < nw; k++) {
      check_cloud(&out[k], false);
      LX_REQUIRE(out[k].stride == 16 && out[k].intensity_offset == 12, "asynchronous downloads need packed float4 records (stride 16, intensity at 12)");
    }
    struct Run { char* dst; const char* src; size_t bytes; };
    std::vector<Run> runs;
    for (uint32_t k = 0; k < nw;) {   // destinations that lie back to back (and are filled exactly) share one copy, as in upload_async
      uint32_t e = k;
      size_t cnt = 0;
      char* base = (char*)out[k].data;
      do {
        const uint32_t n = last_full_off[e + 1] - last_full_off[e];
        const uint32_t m = std::min(n, out[e].count);
        if (n > out[e].count) rc_cap = LOAMX_E_CAPACITY;
        out[e].count = n;
        cnt += m;
        e++;
        if (m != n) break;   // a truncated cloud ends the run
      } while (e < nw && (char*)out[e].data == base + sizeof(float4) * cnt);
      if (cnt) runs.push_back(Run{base, (const char*)(reg.d_full_res() + last_full_off[k]), sizeof(float4) * cnt});
      k = e;
    }
    // The copies go to the SDMA engine directly (hostlink.hpp: the HIP runtime may pick its blit kernel for them, which stalls
    // every kernel that writes to host memory meanwhile) when source and destination are ROCr allocations; else through HIP.
    static const bool via_hip = diag_env("LOAMX_D2H_HIP") != nullptr;   // diagnostic: always hipMemcpyAsync
    bool direct = !via_hip && !runs.empty();
    for (const Run& r : runs) direct = direct && hostlink.can_copy(r.dst, r.src, r.bytes);
    wait_download(par);   // (a caller that downloads the same step twice)
    if (direct) {
      if (ev_reg_done) LX_HIP(hipEventSynchronize(ev_reg_done));   // step() has returned: the registration is complete and its clouds are visible
      try {
        hostlink.begin(par, (uint32_t)runs.size());
        for (const Run& r : runs) hostlink.copy_d2h(par, r.dst, r.src, r.bytes);
        downloads_direct++;
      } catch (const Error&) {   // ROCr refused a copy: the slot has been freed (HostLinkDma::abandon); the whole group goes through HIP
        direct = false;
      }
    }
    if (!direct && !runs.empty()) {
      if (!dstream) dstream = create_stream(0);
      if (!ev_d2h[par]) LX_HIP(hipEventCreateWithFlags(&ev_d2h[par], hipEventDisableTiming));
      LX_HIP(hipStreamWaitEvent(dstream, ev_reg_done, 0));
      for (const Run& r : runs) LX_HIP(hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyDeviceToHost, dstream));
      LX_HIP(hipEventRecord(ev_d2h[par], dstream));
      d2h_pending[par] = true;
      downloads_hip++;
    }
    if (rc_cap != LOAMX_OK) throw Error(LOAMX_E_CAPACITY, "an output cloud is smaller than the registered cloud (count fields hold the needed sizes)");
  }
  void wait_download(int par) {
    hostlink.wait(par);
    if (d2h_pending[par]) { LX_HIP(hipEventSynchronize(ev_d2h[par])); d2h_pending[par] = false; }
  }
  void wait_downloads() {
    for (int par = 0; par < 2; par++) wait_download(par);
  }
  uint64_t run_count = 0;   // registrations run so far (parity = which full-resolution buffer)

  // odometry of staged step t for the streams of one chain (needs the step's features, launched by the calling thread); results go to
  // ores[t % OR]
  void run_odometry(OdomChain& c, uint32_t t) {
    const uint32_t ns = n_streams_, ng = c.s1 - c.s0;
    FeatureExtractor& F = FX(t);
    LX_REQUIRE(LA(t), "internal: odometry of a step whose features were not launched");
    OdometryBatch& odom = *c.ob;
    uint32_t* hb = h_off3[t % OR].p;
    uint32_t* ho[3] = {hb, hb + (ns + 1), hb + 2 * (ns + 1)};
    uint32_t* hlf = hb + 3 * (ns + 1);
    wait_event(evF[t % OR][1]);
    F.check_finite_input();   // (LOAMX_E_INVALID out of step(): a staged sweep with NaN / Inf coordinates)
    c.tr[1] = tr_us();
    const bool timed = timing.load(std::memory_order_relaxed);   // latched: the caller flips the flag while this chain runs steps ahead
    if (timed && c.s0 == 0) LX_HIP(hipEventElapsedTime(&feat_ms[t % OR], evF[t % OR][0], evF[t % OR][1]));
    std::vector<OdomInput> in(ng);
    std::vector<int> rc(ng, 0);
    for (uint32_t s = c.s0; s < c.s1; s++) {
      const uint32_t la = hlf[F.ring_base(s)], lb = hlf[F.ring_base(s + 1)];
      in[s - c.s0] = OdomInput{F.d_feat(0) + ho[0][s], ho[0][s + 1] - ho[0][s], F.d_feat(1) + ho[1][s], ho[1][s + 1] - ho[1][s],
                               F.d_feat(2) + ho[2][s], ho[2][s + 1] - ho[2][s], F.d_less_flat() + la, lb - la};
    }
    const int slot = (int)(t % OR);
    if (timed) {   // this chain's own event pair for this step: both ends are recorded by this thread in this call
      if (!c.tm_a[slot]) { LX_HIP(hipEventCreate(&c.tm_a[slot])); LX_HIP(hipEventCreate(&c.tm_b[slot])); }
      c.tm_pending[slot] = false;
      LX_HIP(hipEventRecord(c.tm_a[slot], odom.stream()));
    }
    if (streaming && rawslot[t % RING].raw)   // imuTrans of this sweep (ScanRegistration publishes it with the clouds; LaserOdometry.cpp:239-248)
      for (uint32_t s = c.s0; s < c.s1; s++) odom.update_imu(s - c.s0, &rawslot[t % RING].imu_trans[12 * (size_t)s]);
    odom.process(in.data(), rc.data(), true);   // returns once the poses are known; clouds ready at odom.tail_event()
    c.tr[2] = tr_us();
    if (!c.ev_tail[slot]) LX_HIP(hipEventCreateWithFlags(&c.ev_tail[slot], hipEventDisableTiming));
    LX_HIP(hipEventRecord(c.ev_tail[slot], odom.stream()));   // behind the tail that process() enqueued
    if (timed) {   // the chain of step t = everything process() enqueued, tail included
      LX_HIP(hipEventRecord(c.tm_b[slot], odom.stream()));
      c.tm_pending[slot] = true;
    }
    for (int k = 0; k < OR; k++)   // (the elapsed time of an older step is taken once its events have completed: measuring never waits)
      if (c.tm_pending[k] && hipEventQuery(c.tm_b[k]) == hipSuccess) {
        float ms = 0.f;
        LX_HIP(hipEventElapsedTime(&ms, c.tm_a[k], c.tm_b[k]));
        c.tm_pending[k] = false;
        c.ms.store(ms, std::memory_order_relaxed);
      }
    for (uint32_t s = c.s0; s < c.s1; s++) {
      const uint32_t l = s - c.s0;
      OdomStream& O = odom.stream_state(l);
      OdomPub& N = ores[slot][s];
      N.transform = O.transform;
      N.transform_sum = O.transform_sum;
      N.stats = O.stats;
      N.rc = rc[l];
      N.last_corner = odom.d_last_corner(l); N.n_last_corner = O.n_last_corner;
      N.last_surf = odom.d_last_surf(l); N.n_last_surf = O.n_last_surf;
      N.to_end = odom.to_end_params(l, true);
    }
  }
  // Head of M(t+1) that does not depend on M(t): transformToEnd of the next step's full-resolution clouds (LaserOdometry.cpp:326) into
  // the registrar's OTHER staging buffer, enqueued behind M(t)'s first Gauss-Newton launches while the host waits for them — as soon
  // as the odometry of t+1 is known (it runs ahead).  step(t+1) then finds its clouds re-projected: one copy + one 12-25 us kernel less
  // between two steps' registrations.  (The host waits for M(t) at an event recorded in front of this work: Registrar::run_iterations.)
  int pre_t = -1;   // the step whose full-resolution clouds have been pre-staged
  void prestage_gather(int tn, int last_staged, hipStream_t s_) {
    static const bool off = !(diag_env("LOAMX_PRESTAGE") && atoi(diag_env("LOAMX_PRESTAGE")) != 0);   // measured: no gain (15.44 k with, 15.51 k without; profiles/r04_ab.md) — the step is not bound by this copy + kernel; opt-in
    if (off || !prefetch || pre_t == tn || tn > last_staged || tn < 1) return;
    const uint32_t ns = n_streams_;
    if (done_min() < tn) return;
    FeatureExtractor& F = FX((uint32_t)tn);
    std::vector<const float4*> fsrc;
    std::vector<uint32_t> nfr;
    std::vector<ToEndParams> tep;
    for (uint32_t s = 0; s < ns; s++) {
      const OdomPub& N = ores[tn % OR][s];
      if (N.rc != LOAMX_OK) continue;
      fsrc.push_back(F.d_cloud() + F.point_base(s));
      nfr.push_back(F.point_base(s + 1) - F.point_base(s));
      tep.push_back(N.to_end);
    }
    const uint32_t nw = (uint32_t)fsrc.size();
    if (!nw) return;
    const int par = (int)(run_count & 1);   // the buffer the NEXT run uses: its last download (two runs ago) must have finished
    hostlink.wait(par);
    if (d2h_pending[par]) { LX_HIP(hipStreamWaitEvent(s_, ev_d2h[par], 0)); d2h_pending[par] = false; }
    float4* dst = reg.stage_full_next(nw, nfr.data());
    if (!dst) return;
    std::vector<uint32_t> foff(nw + 1, 0);
    for (uint32_t k = 0; k < nw; k++) foff[k + 1] = foff[k] + nfr[k];
    chains[0]->ob->to_end_gather(dst, foff.data(), fsrc.data(), tep.data(), nw, s_);
    pre_t = tn;
  }

  // Software pipeline over consecutive steps (the stages are separate ROS nodes in the reference, so nothing in a later
  // stage of step t feeds an earlier stage of step t+1):
  //   registration M(t) on the registrar's stream  ||  odometry O(t+1) on the odometry stream  ||  features F(t+2)
  // step(t) returns when M(t) is complete; O(t+1) / F(t+2) are look-ahead whose results are kept for the next call.
  // host-side timeline of one step (LOAMX_PIPE_TRACE=1): microseconds since step() entry
  bool trace = getenv("LOAMX_PIPE_TRACE") != nullptr;
  std::chrono::steady_clock::time_point tr0, tr_exit = std::chrono::steady_clock::now();
  double tr_us() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); }

  int step(uint32_t t) {
    TraceRange trace_range("loamx:pipeline:step");
    LX_REQUIRE(t < n_staged(), "step index beyond the staged sweeps");
    LX_REQUIRE(!streaming || t + RING >= staged_hi.load(), "this step's slot has been re-staged already");   // (slot t % RING is rewritten by staging step t + RING)
    const auto t_entry = std::chrono::steady_clock::now();
    const double gap_us = std::chrono::duration<double, std::micro>(t_entry - tr_exit).count();   // the caller's time between two steps
    tr0 = t_entry;
    double trM[6] = {0, 0, 0, 0, 0, 0};
    LX_HIP(hipSetDevice(device));
    hipStream_t s_ = reg.stream();
    const uint32_t ns = n_streams_;
    // ---- this step's odometry: published by the look-ahead (the normal case), or run now
    const int ti = (int)t, last_staged = (int)n_staged() - 1;
    {
      bool jump = false;   // a chain for which this is neither a finished step nor its next one: the caller jumped, the chains restart here
      for (auto& c : chains) jump = jump || (ti > c->done.load(std::memory_order_acquire) && ti != c->next.load(std::memory_order_acquire));
      if (jump) {
        park_odometry(ti);
        f_hi = ti - 1;
      }
      LX_REQUIRE(ti + (OR - 2) >= done_max(), "this step's odometry results have been overwritten: steps run in order");
    }
    auto launch_upto = [&](int k) {   // features of the steps up to k (launched by this thread only, in step order)
      if (k > last_staged) k = last_staged;
      while (f_hi < k) { ++f_hi; if (!LA((uint32_t)f_hi)) launch_features((uint32_t)f_hi); }
    };
    if (ti > done_min()) {
      launch_upto(prefetch ? ti + 1 : ti);
      if (prefetch) {
        allow_odometry(std::min(ti + 1, last_staged));
        wait_odometry(ti);
      } else {
        if (o_limit.load(std::memory_order_acquire) >= 0) park_odometry(-1);   // the look-ahead was switched off: the chains go on here
        for (auto& c : chains)
          if (ti > c->done.load(std::memory_order_acquire)) {
            run_odometry(*c, t);
            c->next.store(ti + 1, std::memory_order_release);
            c->done.store(ti, std::memory_order_release);
          }
      }
    }
    LA(t) = 0;   // every chain has consumed the step's features (a restart at this step extracts them again: their offsets' slot is reused by step t + 3)
    for (uint32_t s = 0; s < ns; s++) st[s].cur = ores[t % OR][s];
    FeatureExtractor& F = FX(t);
    const float f_ms = feat_ms[t % OR];
    // the re-projected "last" clouds of THIS sweep are produced at the tails of the odometry chains
    for (auto& c : chains) LX_HIP(hipStreamWaitEvent(s_, c->ev_tail[t % OR], 0));
    // ---- look-ahead while M(t) runs: the odometry chain may go on to step t+1 now and — once M(t) is enqueued and the features
    // of step t+2 are launched (while this thread waits for M(t)'s first look at the flags) — to step t+2
    // (D steps of odometry look-ahead, FD >= 2 of features: in steady state the first two calls find nothing to do — the previous
    // step's launch_f2 went that far — and the one new step of features / odometry is released by launch_f2, off M(t)'s critical path)
    const int D = depth(), FD = std::max(D, 2);
    if (prefetch) {
      launch_upto(ti + FD - 1);
      allow_odometry(std::min(ti + std::min(D, FD - 1), last_staged));
    }
    bool f2_pending = prefetch && ti + 2 <= last_staged;
    auto launch_f2 = [&]() {
      if (f2_pending) { f2_pending = false; launch_upto(ti + FD); if (D >= 2) allow_odometry(std::min(ti + D, last_staged)); }
    };
    int ret = LOAMX_SKIPPED;
    try {
      trM[0] = tr_us();
      // ---- registration against the frozen sub-map: enqueue everything for M(t)
      LazyTimer& tm = tmM[t & 1];
      if (timing) {
        tm.create();
        tm.pending = false;
        LX_HIP(hipEventRecord(tm.a, s_));
      }
      std::vector<const float4*> cl(ns), sl(ns), fsrc(ns);
      std::vector<uint32_t> ncl(ns), nsl(ns), nfr(ns), who;
      std::vector<float> guess;
      std::vector<ToEndParams> tep;
      for (uint32_t s = 0; s < ns; s++) {
        PipeStreamState& P = st[s];
        if (P.cur.rc != LOAMX_OK) continue;   // a stream's first sweep only initialises the odometry
        transform_associate_to_map(P.cur.transform_sum, P.bef, P.aft, P.incre, P.tobe);
        float g[6];
        P.tobe.get(g);
        guess.insert(guess.end(), g, g + 6);
        const uint32_t k = (uint32_t)who.size();
        cl[k] = P.cur.last_corner; ncl[k] = P.cur.n_last_corner;
        sl[k] = P.cur.last_surf; nsl[k] = P.cur.n_last_surf;
        fsrc[k] = F.d_cloud() + F.point_base(s);
        nfr[k] = F.point_base(s + 1) - F.point_base(s);
        tep.push_back(P.cur.to_end);
        who.push_back(s);
      }
      const uint32_t nw = (uint32_t)who.size();
      if (nw) {
        // the full-resolution clouds are re-projected to the sweep end before they are registered (LaserOdometry.cpp:326):
        // one fused kernel writes them straight into the registrar's staging area
        std::vector<uint32_t> foff(nw + 1, 0);
        for (uint32_t k = 0; k < nw; k++) foff[k + 1] = foff[k] + nfr[k];
        const bool adopted = pre_t == ti && reg.adopt_full_next(nw, nfr.data());   // re-projected already, behind the previous step's first iterations
        pre_t = -1;
        if (!adopted) {
          if (reg.double_buffer_full) {   // this run reuses the buffer of the run before last: its download must have finished
            const int par = (int)(run_count & 1);
            const double tw0 = trace ? tr_us() : 0.0;
            hostlink.wait(par);   // (two steps old: landed long ago)
            if (trace) trM[4] = tr_us() - tw0;
            if (d2h_pending[par]) { LX_HIP(hipStreamWaitEvent(s_, ev_d2h[par], 0)); d2h_pending[par] = false; }
          }
          float4* full_dst = reg.stage_full(nw, nfr.data());
          chains[0]->ob->to_end_gather(full_dst, foff.data(), fsrc.data(), tep.data(), nw, s_);
        }
        last_full_off = foff;
        run_count++;
        reg.upload_device(nw, cl.data(), ncl.data(), sl.data(), nsl.data(), nullptr, nullptr, guess.data());
        // transformUpdate's IMU blend (BasicLaserMapping.cpp:171-200) changes transformTobeMapped BEFORE the full-resolution cloud is
        // registered (:235-240): a step with mapping-side IMU data for any of its streams registers its clouds after the blend
        bool blend = false;
        static const bool no_blend = diag_env("LOAMX_NO_MAP_IMU_BLEND") != nullptr;   // (tests: what the poses would be without the blend)
        if (!no_blend && streaming && rawslot[t % RING].raw && !rawslot[t % RING].scan_time.empty())
          for (uint32_t k = 0; k < nw; k++) blend = blend || rawslot[t % RING].map_imu_upto[who[k]] > 0;
        reg.defer_full = blend;
        reg.on_first_wait = [&]() { launch_f2(); prestage_gather(ti + 1, last_staged, s_); };
        reg.run_async();
        reg.on_first_wait = nullptr;
        if (blend) {
          std::vector<float> p6(6 * (size_t)nw);
          reg.download(p6.data(), nullptr);
          if (reg.submap_sufficient())   // (transformUpdate is only reached when the optimisation ran, :628-629)
            for (uint32_t k = 0; k < nw; k++)
              (void)map_imu_blend(who[k], rawslot[t % RING].map_imu_upto[who[k]], rawslot[t % RING].scan_time[who[k]], fcfg.scan_period, &p6[6 * (size_t)k]);
          reg.finish_with_poses(p6.data());
          reg.defer_full = false;
        }
        if (reg.double_buffer_full) {
          if (!ev_reg_done) LX_HIP(hipEventCreateWithFlags(&ev_reg_done, hipEventDisableTiming));
          LX_HIP(hipEventRecord(ev_reg_done, s_));
        }
      } else {
        last_full_off.clear();
      }
      launch_f2();
      if (nw) prestage_gather(ti + 1, last_staged, s_);   // (no-op when the first wait's callback did it)
      if (timing) {
        LX_HIP(hipEventRecord(tm.b, s_));
        tm.pending = true;
      }
      trM[1] = tr_us();
      // ---- finish M(t)
      if (nw) {
        std::vector<float> poses(6 * nw);
        std::vector<SweepStats> ss(nw);
        reg.download(poses.data(), nullptr);
        reg.download_stats(ss.data());
        for (uint32_t k = 0; k < nw; k++) {
          PipeStreamState& P = st[who[k]];
          P.map_stats = ss[k];
          P.mapped = true;
          if (reg.submap_sufficient()) {   // transformUpdate (BasicLaserMapping.cpp:171-203, :628-629)
            P.tobe.set(&poses[6 * k]);
            P.bef = P.cur.transform_sum;
            P.aft = P.tobe;
          }
        }
        ret = LOAMX_OK;
      }
      trM[2] = tr_us();
    } catch (...) {
      throw;
    }
    trM[3] = tr_us();
    last_step = (long)t;
    tr_exit = std::chrono::steady_clock::now();
    if (trace) {
      fprintf(stderr, "[pipe t=%u] caller gap %.0f | M-start %.0f  M-enqueued %.0f  M-downloaded %.0f  O-joined %.0f  (download-wait %.0f) |", t, gap_us, trM[0], trM[1], trM[2], trM[3], trM[4]);
      for (auto& c : chains)   // (each chain's most recent pass, whichever step that was: start, features ready, process() returned, end)
        fprintf(stderr, " O[%u-%u): %.0f %.0f %.0f %.0f done %d pass %.0f us |", c->s0, c->s1, c->tr[0], c->tr[1], c->tr[2], c->tr[3], c->done.load(), (double)c->last_us.load());
      fprintf(stderr, " f_hi %d o_limit %d staged %u", f_hi, o_limit.load(), n_staged());
      fprintf(stderr, "\n");
    }
    if (timing) {
      for (auto& x : tmM) x.resolve();
      last_ms[0] = f_ms;               // on the feature stream (overlapped)
      {   // the odometry chains (overlapped with the registrations): the longest chain's most recent timed pass
        float m = 0.f;
        for (auto& c : chains) m = std::max(m, c->ms.load(std::memory_order_relaxed));
        last_ms[1] = m;
      }
      last_ms[2] = tmM[t & 1].pending ? tmM[(t + 1) & 1].ms : tmM[t & 1].ms;   // the previous step's while this one is in flight
      last_ms[3] = last_ms[2];
    }
    return ret;
  }
};

}  // namespace loamx

using namespace loamx;

struct loamx_pipeline {
  Pipeline p;
  loamx_pipeline(const loamx_scanreg_config& f, const loamx_odom_config& o, const loamx_map_config& m, uint32_t n) : p(f, o, m, n) {}
};

extern "C" {

loamx_pipeline* loamx_pipeline_create(const loamx_scanreg_config* fcfg, const loamx_odom_config* ocfg, const loamx_map_config* mcfg,
                                      uint32_t n_streams) {
  loamx_pipeline* h = nullptr;
  guard([&]() {
    loamx_scanreg_config f;
    loamx_odom_config o;
    loamx_map_config m;
    if (fcfg) f = *fcfg; else loamx_scanreg_default_config(&f);
    if (ocfg) o = *ocfg; else loamx_odom_default_config(&o);
    if (mcfg) m = *mcfg; else loamx_map_default_config(&m);
    LX_REQUIRE(n_streams >= 1 && n_streams <= 1024, "n_streams must be in [1, 1024]");
    LX_REQUIRE(f.device == m.device && o.device == m.device, "all three configurations must name the same device");
    LX_REQUIRE(f.max_corner_less_sharp == 0 || f.max_corner_less_sharp >= f.max_corner_sharp, "max_corner_less_sharp must be >= max_corner_sharp");
    h = new loamx_pipeline(f, o, m, n_streams);
    return LOAMX_OK;
  });
  return h;
}
void loamx_pipeline_destroy(loamx_pipeline* h) { delete h; }

int loamx_pipeline_set_frozen(loamx_pipeline* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->p.reg.set_submap_host(corner_map, surf_map); return LOAMX_OK; });
}
int loamx_pipeline_set_frozen_device(loamx_pipeline* h, const void* d_corner, uint32_t nc, const void* d_surf, uint32_t ns) {
  return guard([&]() {
    LX_REQUIRE(h && (d_corner || !nc) && (d_surf || !ns), "NULL argument");
    h->p.reg.set_submap_device((const float4*)d_corner, nc, (const float4*)d_surf, ns);
    return LOAMX_OK;
  });
}
int loamx_pipeline_stage_frozen_device(loamx_pipeline* h, const void* d_corner, uint32_t nc, const void* d_surf, uint32_t ns, void* wait_event) {
  return guard([&]() {
    LX_REQUIRE(h && (d_corner || !nc) && (d_surf || !ns), "NULL argument");
    h->p.reg.stage_submap_device((const float4*)d_corner, nc, (const float4*)d_surf, ns, (hipEvent_t)wait_event);
    return LOAMX_OK;
  });
}
int loamx_pipeline_stage_frozen(loamx_pipeline* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->p.reg.stage_submap_host(corner_map, surf_map);
    return LOAMX_OK;
  });
}
int loamx_pipeline_swap_frozen(loamx_pipeline* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    if (!h->p.reg.submap_staged()) return (int)LOAMX_SKIPPED;
    h->p.reg.swap_submap();
    return (int)LOAMX_OK;
  });
}
int loamx_pipeline_set_state(loamx_pipeline* h, uint32_t stream, const float* transform, const float* transform_sum, const float* bef,
                             const float* aft) {
  return guard([&]() {
    LX_REQUIRE(h && stream < h->p.n_streams_, "invalid stream");
    h->p.park_odometry(-1);   // the odometry chains stop where they are; the sweeps they have not processed yet start from the new state
    if (transform) h->p.OB(stream).stream_state(h->p.LS(stream)).transform.set(transform);
    if (transform_sum) h->p.OB(stream).stream_state(h->p.LS(stream)).transform_sum.set(transform_sum);
    if (bef) h->p.st[stream].bef.set(bef);
    if (aft) h->p.st[stream].aft.set(aft);
    return LOAMX_OK;
  });
}
int loamx_pipeline_upload(loamx_pipeline* h, uint32_t n_steps, const loamx_cloud* clouds, const uint32_t* const* ring_size,
                          const uint32_t* n_rings) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->p.upload(n_steps, clouds, ring_size, n_rings); return LOAMX_OK; });
}
int loamx_pipeline_stage_step(loamx_pipeline* h, uint32_t step, const loamx_cloud* clouds, const uint32_t* const* ring_size, const uint32_t* n_rings) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->p.stage_step(step, clouds, ring_size, n_rings); return LOAMX_OK; });
}
int loamx_pipeline_stage_step_raw(loamx_pipeline* h, uint32_t step, const void* const* raw_xyz, const uint32_t* counts, uint32_t stride,
                                  const loamx_multiscan_mapper* mapper, const double* scan_time_sec) {
  return guard([&]() {
    LX_REQUIRE(h && mapper, "NULL argument");
    h->p.stage_step_raw(step, raw_xyz, counts, stride, *mapper, scan_time_sec);
    return LOAMX_OK;
  });
}
int loamx_pipeline_update_imu(loamx_pipeline* h, uint32_t stream, double stamp_sec, float roll, float pitch, float yaw, const float acc_xyz[3]) {
  return guard([&]() {
    LX_REQUIRE(h && acc_xyz && stream < h->p.n_streams_, "invalid argument");
    h->p.imu[stream].update(stamp_sec, roll, pitch, yaw, acc_xyz);
    h->p.map_imu_push(stream, stamp_sec, roll, pitch);   // (the same topic feeds LaserMapping's history in the reference's node graph)
    return LOAMX_OK;
  });
}
int loamx_pipeline_enable_async_downloads(loamx_pipeline* h) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->p.reg.double_buffer_full = true; return LOAMX_OK; });
}
int loamx_pipeline_download_step_async(loamx_pipeline* h, loamx_cloud* out, uint32_t n_out) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->p.download_step_async(out, n_out); return LOAMX_OK; });
}
int loamx_pipeline_download_counts(loamx_pipeline* h, uint64_t counts[2]) {
  return guard([&]() {
    LX_REQUIRE(h && counts, "NULL argument");
    counts[0] = h->p.downloads_direct;
    counts[1] = h->p.downloads_hip;
    return LOAMX_OK;
  });
}
int loamx_pipeline_wait_downloads(loamx_pipeline* h) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->p.wait_downloads(); return LOAMX_OK; });
}
int loamx_pipeline_step(loamx_pipeline* h, uint32_t step) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); return h->p.step(step); });
}
int loamx_pipeline_get(loamx_pipeline* h, uint32_t stream, float* transform, float* transform_sum, float* aft, int* stats8) {
  return guard([&]() {
    LX_REQUIRE(h && stream < h->p.n_streams_, "invalid stream");
    PipeStreamState& P = h->p.st[stream];
    const OdomPub& O = P.cur;   // the sweep that was registered last (odometry itself may already be one sweep ahead)
    if (transform) O.transform.get(transform);
    if (transform_sum) O.transform_sum.get(transform_sum);
    if (aft) P.aft.get(aft);
    if (stats8) {
      stats8[0] = O.stats.iterations; stats8[1] = O.stats.sel; stats8[2] = P.map_stats.iterations; stats8[3] = P.map_stats.sel;
      stats8[4] = P.map_stats.corner_q; stats8[5] = P.map_stats.surf_q; stats8[6] = P.map_stats.degenerate; stats8[7] = P.mapped ? 1 : 0;
    }
    return LOAMX_OK;
  });
}
int loamx_pipeline_download_full_res(loamx_pipeline* h, uint32_t slot, loamx_cloud* out) {
  return guard([&]() { LX_REQUIRE(h && out, "NULL argument"); return h->p.reg.download_full_res(slot, out); });
}
int loamx_pipeline_download_last_clouds(loamx_pipeline* h, uint32_t stream, loamx_cloud* last_corner, loamx_cloud* last_surf) {
  return guard([&]() {
    LX_REQUIRE(h && last_corner && last_surf, "NULL argument");
    LX_REQUIRE(stream < h->p.n_streams_, "stream index out of range");
    LX_REQUIRE(h->p.last_step.load() >= 0, "no step has run");
    check_cloud(last_corner, false);
    check_cloud(last_surf, false);
    LX_HIP(hipSetDevice(h->p.device));
    // (produced on the odometry chain's stream, behind the event the step's registration waited for: complete since step() returned;
    // their buffer is rewritten only once the chain has gone on for more steps than the look-ahead allows before the next step())
    const OdomPub& N = h->p.st[stream].cur;
    std::vector<float4> tmp(std::max(N.n_last_corner, N.n_last_surf));
    if (N.n_last_corner) LX_HIP(hipMemcpy(tmp.data(), N.last_corner, sizeof(float4) * N.n_last_corner, hipMemcpyDeviceToHost));
    const int rc = unpack_cloud(tmp.data(), N.n_last_corner, last_corner);
    if (N.n_last_surf) LX_HIP(hipMemcpy(tmp.data(), N.last_surf, sizeof(float4) * N.n_last_surf, hipMemcpyDeviceToHost));
    const int rs = unpack_cloud(tmp.data(), N.n_last_surf, last_surf);
    return rc != LOAMX_OK ? rc : rs;
  });
}
int loamx_pipeline_set_lookahead(loamx_pipeline* h, int on) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    if (!on && h->p.prefetch) h->p.park_odometry(-1);   // (every worker finishes the step it is in; the chains continue from there)
    h->p.prefetch = on != 0;
    return LOAMX_OK;
  });
}
int loamx_pipeline_drain_lookahead(loamx_pipeline* h, int* last_odometry_step) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    const int done = h->p.drain_lookahead();
    if (last_odometry_step) *last_odometry_step = done;
    return LOAMX_OK;
  });
}
int loamx_pipeline_set_timing(loamx_pipeline* h, int on) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->p.timing = on != 0;
    h->p.reg.set_timing(on != 0, on != 2);
    for (auto& c : h->p.chains) c->ob->set_launch_timing(on != 0 && on != 2);
    return LOAMX_OK;
  });
}
int loamx_pipeline_get_odom_launch_timing(loamx_pipeline* h, double ms4[4], uint64_t counts7[7]) {
  return guard([&]() {
    LX_REQUIRE(h && ms4 && counts7, "NULL argument");
    for (int k = 0; k < 4; k++) ms4[k] = 0.0;
    for (int k = 0; k < 7; k++) counts7[k] = 0;
    for (auto& c : h->p.chains) {
      const OdometryBatch::LaunchTotals t = c->ob->launch_totals();
      ms4[0] += t.lm_ms; ms4[1] += t.lm_noop_ms; ms4[2] += t.corr_ms; ms4[3] += t.corr_noop_ms;
      counts7[0] += t.lm_launches; counts7[1] += t.lm_noop_launches; counts7[2] += t.lm_iterations; counts7[3] += t.corr_launches;
      counts7[4] += t.corr_noop_launches; counts7[5] += t.lm_bytes; counts7[6] += t.corr_features;
    }
    return LOAMX_OK;
  });
}
int loamx_pipeline_get_timing(loamx_pipeline* h, float stage_ms[4], float reg_ms[4], uint64_t counts[4]) {
  return guard([&]() {
    LX_REQUIRE(h && stage_ms && reg_ms && counts, "NULL argument");
    for (int k = 0; k < 4; k++) stage_ms[k] = h->p.last_ms[k];
    h->p.reg.get_timing(reg_ms, counts);
    return LOAMX_OK;
  });
}
int loamx_pipeline_lookahead_depth(loamx_pipeline* h) { return h ? h->p.depth() : 0; }
void* loamx_pipeline_stream(loamx_pipeline* h) { return h ? (void*)h->p.reg.stream() : nullptr; }

}  // extern "C"
