// Sequential scan-to-map registration with a live map (BasicLaserMapping::process) for gfx950.
//
// Map layout in HBM (per feature type): ONE flat packed-float4 array + a parallel uint32 tag array holding the
// absolute 50 m-cube coordinates of each point.  The reference's 21x11x21 pointer grid (BasicLaserMapping.cpp:60-95,
// :311-441) becomes pure bookkeeping: shifting the window only changes (cenW, cenH, cenD); points whose cube leaves
// the window are dropped at the next partition.  Per sweep (one pass over the map at HBM speed each):
//   k_map_classify + scans + k_map_partition   sub-map (valid cubes, :503-509) | rest | dropped
//   Registrar (B = 1)                           stack round trip, voxel DS, grid index, <= 10 GN iterations (:511-533)
//   k_map_insert                                re-project DS features with the optimised pose, bucket into cubes (:536-577)
//   VoxelPipeline (segments = valid cubes)      per-cube pcl::VoxelGrid re-filtering (:580-593)
//   k_map_append / k_map_hist                   next map = rest ++ filtered; per-cube counts for the host directory
// Host: closed-form pose prediction (:103-167), cube window + field-of-view selection (:300-500), transformUpdate.
#include "registration.hpp"
#include "api_handles.h"
#include "pinned_copy.hpp"
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include "host_math.h"
#include "scan.hpp"

namespace loamx {

constexpr int MW = 21, MH = 11, MD = 21, MCUBES = MW * MH * MD;

__host__ __device__ inline uint32_t pack_tag(int ia, int ja, int ka) {
  return (uint32_t)(ia + 512) | ((uint32_t)(ja + 512) << 10) | ((uint32_t)(ka + 512) << 20);
}
__host__ __device__ inline void unpack_tag(uint32_t t, int& ia, int& ja, int& ka) {
  ia = (int)(t & 1023u) - 512;
  ja = (int)((t >> 10) & 1023u) - 512;
  ka = (int)((t >> 20) & 1023u) - 512;
}
// cube index of a map coordinate relative to the map origin (BasicLaserMapping.cpp:303-309 / :540-546 without the
// window centre): double arithmetic, truncation, negative fix-up
__host__ __device__ inline int cube_abs(float v) {
  const double CUBE_SIZE = 50.0, CUBE_HALF = CUBE_SIZE / 2;
  int c = (int)(((double)v + CUBE_HALF) / CUBE_SIZE);
  if ((double)v + CUBE_HALF < 0) c--;
  return c;
}

struct MapWindow {
  int cen[3];
};

// ---- the map's split of one sweep (:300-509 without the pointer grid): every map point is valid (its cube is in the field of view:
// it joins the sub-map), rest (inside the window, not visible) or dropped (its cube left the window).  ONE launch (rounds 1-4:
// classify, two device-wide scans, partition): a workgroup classifies its tile of 2048 points, scans the two flags inside the tile,
// finds the tile's two offsets by a decoupled look-back over the tiles before it (scan.hpp: chain_lookback) and moves its points —
// order preserved in both outputs.  valid -> (sub_pts, sub_seg, sub_valid = 1): where the per-cube re-filtering of the map update reads
// the sub-map (its first n_sub slots), folded into the bounds of the sub-map's grid index on the way (SubMapIndex::d_bounds);
// rest -> (new_pts, new_tags).  The last tile records the totals: counters[1] = valid, counters[3] = rest.
constexpr int MS_TILE = 2048;
__global__ __launch_bounds__(256) void k_map_split(const float4* __restrict__ pts, const uint32_t* __restrict__ tags, uint32_t n, MapWindow w,
                                                   const short* __restrict__ slot_lut, float4* __restrict__ sub_pts, uint32_t* __restrict__ sub_seg,
                                                   uint8_t* __restrict__ sub_valid, float4* __restrict__ new_pts, uint32_t* __restrict__ new_tags,
                                                   uint32_t* __restrict__ bounds, unsigned long long* __restrict__ chain_v,
                                                   unsigned long long* __restrict__ chain_r, unsigned long long epoch, uint32_t* __restrict__ counters,
                                                   uint32_t* __restrict__ err_word) {
  __shared__ uint32_t lds[17];
  __shared__ uint32_t s_base[2];
  __shared__ float red[4][6];
  const uint32_t b = blockIdx.x, i0 = b * MS_TILE + threadIdx.x * 8u;
  uint32_t cls[8], slot[8], tg[8];   // cls: 0 dropped, 1 rest, 2 valid
  uint32_t nv = 0, nr = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t i = i0 + k;
    cls[k] = 0; slot[k] = 0; tg[k] = 0;
    if (i < n) {
      tg[k] = tags[i];
      int ia, ja, ka;
      unpack_tag(tg[k], ia, ja, ka);
      const int I = ia + w.cen[0], J = ja + w.cen[1], K = ka + w.cen[2];
      if (I >= 0 && I < MW && J >= 0 && J < MH && K >= 0 && K < MD) {
        const int sl = slot_lut[I + MW * J + MW * MH * K];
        if (sl >= 0) { cls[k] = 2; slot[k] = (uint32_t)sl; nv++; } else { cls[k] = 1; nr++; }
      }
    }
  }
  uint32_t tv, tr;
  const uint32_t ov = block_excl_scan(nv, lds, tv);
  const uint32_t orr = block_excl_scan(nr, lds, tr);
  if (threadIdx.x < 64) {
    bool failed = false;
    const uint32_t ev = chain_lookback(chain_v, epoch, b, tv, failed);
    const uint32_t er = chain_lookback(chain_r, epoch, b, tr, failed);
    if (threadIdx.x == 0) {
      s_base[0] = ev; s_base[1] = er;
      if (b + 1 == gridDim.x) { counters[0] = n; counters[1] = ev + tv; counters[2] = n; counters[3] = er + tr; }
      if (failed && err_word) __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  uint32_t dv = s_base[0] + ov, dr = s_base[1] + orr;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (cls[k] == 2) {
      const float4 p = pts[i0 + k];
      sub_pts[dv] = p; sub_seg[dv] = slot[k]; sub_valid[dv] = 1;
      dv++;
      mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
      mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
      mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    } else if (cls[k] == 1) {
      new_pts[dr] = pts[i0 + k]; new_tags[dr] = tg[k];
      dr++;
    }
  }
  // bounds of the sub-map: per wave (shuffles), per workgroup (LDS), one atomic per workgroup and word
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { red[wid][a] = mn[a]; red[wid][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    float v = red[0][a];
    for (int q = 1; q < 4; q++) v = a < 3 ? fminf(v, red[q][a]) : fmaxf(v, red[q][a]);
    if (a < 3 ? v != FLT_MAX : v != -FLT_MAX) {
      if (a < 3) atomicMin(&bounds[a], enc_f32(v)); else atomicMax(&bounds[a], enc_f32(v));
    }
  }
}

// re-project the down-sampled features of one type with the final pose and bucket them (:536-577), ONE launch (rounds 1-4: insert,
// a device-wide scan, append).  Filter-input slot (n_old + j) receives feature j when its cube is valid; features in other in-window
// cubes go behind the rest points carried over by the split (*d_base = counters[3]), in feature order — their place by a scan inside
// the tile and a look-back over the tiles before it; the remainder (outside the window) is dropped as in the reference.
// The last tile records how many were appended (counters[5]).
__global__ __launch_bounds__(256) void k_map_insert(const float4* __restrict__ ds_pts, const uint32_t* __restrict__ ds_off, int type,
                                                    uint32_t n_slots, const Pose* __restrict__ pose, MapWindow w,
                                                    const short* __restrict__ slot_lut, uint32_t n_old, float4* __restrict__ fin_pts,
                                                    uint32_t* __restrict__ fin_seg, uint8_t* __restrict__ fin_valid,
                                                    const uint32_t* __restrict__ d_base, float4* __restrict__ new_pts, uint32_t* __restrict__ new_tags,
                                                    unsigned long long* __restrict__ chain, unsigned long long epoch, uint32_t* __restrict__ counters,
                                                    uint32_t* __restrict__ err_word) {
  __shared__ uint32_t lds[17];
  __shared__ uint32_t s_base;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t a = ds_off[type], b = ds_off[type + 1];
  uint8_t valid = 0;
  uint32_t rest = 0, tag = 0;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < n_slots && j < b - a) {
    const Pose T = *pose;
    p = ds_pts[a + j];
    to_map(T, p.x, p.y, p.z);
    const int ia = cube_abs(p.x), ja = cube_abs(p.y), ka = cube_abs(p.z);
    const int I = ia + w.cen[0], J = ja + w.cen[1], K = ka + w.cen[2];
    if (I >= 0 && I < MW && J >= 0 && J < MH && K >= 0 && K < MD) {
      const int slot = slot_lut[I + MW * J + MW * MH * K];
      if (slot >= 0) {
        valid = 1;
        fin_pts[n_old + j] = p;
        fin_seg[n_old + j] = (uint32_t)slot;
      } else {
        rest = 1;
        tag = pack_tag(ia, ja, ka);
      }
    }
  }
  if (j < n_slots) fin_valid[n_old + j] = valid;
  uint32_t total;
  const uint32_t off = block_excl_scan(rest, lds, total);
  if (threadIdx.x < 64) {
    bool failed = false;
    const uint32_t ex = chain_lookback(chain, epoch, blockIdx.x, total, failed);
    if (threadIdx.x == 0) {
      s_base = ex;
      if (blockIdx.x + 1 == gridDim.x) { counters[4] = n_slots; counters[5] = ex + total; }
      if (failed && err_word) __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (rest) {
    const uint32_t d = *d_base + s_base + off;
    new_pts[d] = p;
    new_tags[d] = tag;
  }
}

// filtered voxels (segment-ordered) go behind rest + inserted; base = d_base0 + d_base1
__global__ __launch_bounds__(256) void k_map_append_filtered(const float4* __restrict__ filt, const uint32_t* __restrict__ out_off,
                                                             uint32_t nslots, const uint32_t* __restrict__ slot_tag, uint32_t max_n,
                                                             const uint32_t* __restrict__ d_base0, const uint32_t* __restrict__ d_base1,
                                                             float4* __restrict__ new_pts, uint32_t* __restrict__ new_tags,
                                                             uint32_t* __restrict__ d_total, uint32_t* __restrict__ hist_to_clear) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t c = v; c < (uint32_t)MCUBES; c += gridDim.x * blockDim.x) hist_to_clear[c] = 0u;   // (k_map_hist runs next on this stream)
  const uint32_t nf = out_off[nslots];
  const uint32_t base = *d_base0 + *d_base1;
  if (v == 0) *d_total = base + nf;
  if (v >= nf || v >= max_n) return;
  uint32_t lo = 0, hi = nslots;   // out_off[lo] <= v < out_off[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (out_off[mid] <= v) lo = mid; else hi = mid;
  }
  new_pts[base + v] = filt[v];
  new_tags[base + v] = slot_tag[lo];
}

// per-cube point counts of the new map.  The map's points sit in a few dozen cubes: 200 k global atomics on those few counters
// serialise (measured: 0.6 ms per launch); every workgroup counts in LDS first and adds each cube it touched once.
__global__ __launch_bounds__(256) void k_map_hist(const uint32_t* __restrict__ tags, const uint32_t* __restrict__ d_n, uint32_t max_n,
                                                  MapWindow w, uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_hist[MCUBES];
  for (int c = (int)threadIdx.x; c < MCUBES; c += 256) s_hist[c] = 0u;
  __syncthreads();
  const uint32_t n = min(*d_n, max_n);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    int ia, ja, ka;
    unpack_tag(tags[i], ia, ja, ka);
    const int I = ia + w.cen[0], J = ja + w.cen[1], K = ka + w.cen[2];
    if (I >= 0 && I < MW && J >= 0 && J < MH && K >= 0 && K < MD) atomicAdd(&s_hist[I + MW * J + MW * MH * K], 1u);
  }
  __syncthreads();
  for (int c = (int)threadIdx.x; c < MCUBES; c += 256)
    if (s_hist[c]) atomicAdd(&hist[c], s_hist[c]);
}

// gather flags for the surround cloud (createDownsizedMap :251-257)
__global__ __launch_bounds__(256) void k_map_surround_flags(const uint32_t* __restrict__ tags, const uint32_t* __restrict__ d_n,
                                                            uint32_t max_n, MapWindow w, const uint8_t* __restrict__ sur_lut,
                                                            uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_n) return;
  uint32_t f = 0;
  if (i < *d_n) {
    int ia, ja, ka;
    unpack_tag(tags[i], ia, ja, ka);
    const int I = ia + w.cen[0], J = ja + w.cen[1], K = ka + w.cen[2];
    if (I >= 0 && I < MW && J >= 0 && J < MH && K >= 0 && K < MD) f = sur_lut[I + MW * J + MW * MH * K];
  }
  flag[i] = f;
}
__global__ __launch_bounds__(256) void k_map_compact(const float4* __restrict__ pts, const uint32_t* __restrict__ flag,
                                                     const uint32_t* __restrict__ scan, uint32_t n, uint32_t dst_base,
                                                     const uint32_t* __restrict__ d_dst_base, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  const uint32_t b = dst_base + (d_dst_base ? *d_dst_base : 0u);
  out[b + scan[i]] = pts[i];
}
__global__ void k_map_surround_valid(uint8_t* v, uint32_t max_n, const uint32_t* __restrict__ d_nc, const uint32_t* __restrict__ d_ns) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < max_n) v[i] = i < (*d_nc + *d_ns) ? 1 : 0;
}

// ----------------------------------------------------------------------------------------------------------------
struct TypeMap {   // one feature type (corner / surf)
  DevBuf<float4> pts[2];      // double-buffered map
  DevBuf<uint32_t> tags[2];
  int cur = 0;
  uint32_t n = 0;             // host copy of the point count (exact after every process())
  uint32_t room = 0;          // points every buffer of this type has room for (ensure())
  std::vector<uint32_t> cube_cnt = std::vector<uint32_t>(MCUBES, 0);   // host directory, window coordinates
  // per-sweep work buffers
  DevBuf<uint32_t> fin_seg, out_off, hist;
  DevBuf<float4> fin, filt;   // fin: the per-cube re-filtering's input — the sub-map (written by the partition) followed by the sweep's new features
  DevBuf<uint8_t> fin_valid;
  struct { uint32_t* p = nullptr; } counters;   // view into hist (behind the MCUBES bins): [0..1] valid scan n/total, [2..3] rest scan n/total, [4..5] insert-rest scan, [6] new total
  VoxelPipeline vox;
  PinBuf<uint32_t> h_hist;     // the histogram and the counters behind it
  // look-back words of the fused kernels' chains (k_map_split: valid, rest; k_map_insert: rest), tagged with a per-launch epoch
  DevBuf<unsigned long long> chain;
  size_t chain_stride = 0;
  uint32_t chain_epoch = 0;
};

class Mapper {
 public:
  explicit Mapper(const loamx_map_config& cfg);
  loamx_map_config cfg;
  Registrar reg;
  HTwist sum, incre, tobe, bef, aft;
  int cen[3] = {10, 5, 10};
  long frame_count = 0, map_frame_count = 4;
  bool fresh_map = false;
  SweepStats last_stats = {0, 0, 0, 0, 0, 0, 0, 0};
  bool last_optimized = false;
  // IMUState2 history (BasicLaserMapping.h:47-75, capacity 200 :56), stamps / laserOdometryTime in seconds
  struct ImuState2 { double stamp; float roll, pitch; };
  std::deque<ImuState2> imu_history;
  double laser_odometry_time = 0.0;
  void update_imu(double stamp, float roll, float pitch) {   // updateIMU :602-605
    if (imu_history.size() >= 200) imu_history.pop_front();
    imu_history.push_back({stamp, roll, pitch});
  }
  // the IMU part of transformUpdate (:173-200) applied to (rot_x, rot_z) of a pose
  void imu_blend_pose(float* p6) const {
    size_t i = 0;
    while (i < imu_history.size() - 1 && (laser_odometry_time - imu_history[i].stamp) + cfg.scan_period > 0) i++;
    float roll, pitch;
    if (i == 0 || (laser_odometry_time - imu_history[i].stamp) + cfg.scan_period > 0) {
      roll = imu_history[i].roll; pitch = imu_history[i].pitch;
    } else {
      const float ratio = (float)(((imu_history[i].stamp - laser_odometry_time) - cfg.scan_period) / (imu_history[i].stamp - imu_history[i - 1].stamp));
      const float inv = 1 - ratio;
      roll = imu_history[i].roll * inv + imu_history[i - 1].roll * ratio;
      pitch = imu_history[i].pitch * inv + imu_history[i - 1].pitch * ratio;
    }
    p6[0] = (float)(0.998 * p6[0] + 0.002 * pitch);
    p6[2] = (float)(0.998 * p6[2] + 0.002 * roll);
  }
  uint32_t last_sub[2] = {0, 0};
  TypeMap tm[2];
  // slot look-up table (short[MCUBES]) | surround flags (uint8[MCUBES]) | cube tags of the valid slots — two sets: the one the current
  // sweep's kernels read and the one a speculative partition (below) was given
  DevBuf<char> d_tables[2];
  PinBuf<char> h_tables[2];
  int tab_cur = 0;
  // What a sweep's pose decides about the map before anything is registered: the cube window, the valid cubes of the 5x5x5
  // neighbourhood in the reference's order (:443-500), their slots, the sub-map's size.
  struct Plan {
    int cen[3] = {0, 0, 0};
    int nvalid = 0;
    std::vector<short> lut;
    std::vector<uint8_t> slut;
    std::vector<uint32_t> stag;
    uint32_t n_sub[2] = {0, 0};
    bool same_as(const Plan& o) const {
      return cen[0] == o.cen[0] && cen[1] == o.cen[1] && cen[2] == o.cen[2] && nvalid == o.nvalid && n_sub[0] == o.n_sub[0] &&
             n_sub[1] == o.n_sub[1] && lut == o.lut && slut == o.slut && stag == o.stag;
    }
  };
  // may_shift = false: give up (return false) when the cube window would have to move first
  bool make_plan(const HTwist& pose, bool may_shift, Plan& out);
  // tables up (set `set`), partition of both feature types, their grid indices: everything of a sweep that depends on the plan alone
  void enqueue_partition(const Plan& plan, int set, const uint32_t n_in_room[2]);
  // SPECULATION: the partition and the index build of the NEXT sweep depend only on the map as this sweep's update leaves it and on the
  // valid-cube list of the next pose — discrete outputs of a pose that moves by centimetres per sweep.  The helper thread predicts the
  // next transformTobeMapped (constant-velocity odometry through transformAssociateToMap), plans for it and enqueues partition + index
  // behind the update, while the caller is busy with the next sweep's extraction and odometry; the next process() plans for the TRUE pose
  // and adopts the prepared work when — and only when — the two plans are equal in every entry (the prepared buffers are then exactly
  // what it would have produced); otherwise it partitions as before.  ~60 us of host enqueue + ~40 us of device work off the sweep's path.
  Plan spec_plan;
  bool spec_valid = false;
  bool spec_enabled = getenv("LOAMX_MAP_NO_SPECULATION") == nullptr;   // (read when the handle is made; result-neutral: a prepared partition is only ever adopted when it is the one the sweep would build)
  uint64_t spec_hits = 0, spec_misses = 0;
  float sum_prev6[6] = {0, 0, 0, 0, 0, 0};   // transformSum of the previous processed sweep (the prediction's velocity)
  bool have_sum_prev = false;
  struct SpecInputs { float sum6[6], sum_prev6[6]; HTwist bef, aft; bool ok = false, ready = false; uint32_t n_in_room[2] = {0, 0}; } spec_in;   // (guarded by helper.mu)
  uint32_t spec_room[2] = {0, 0};   // n_in_room of the partition that was prepared (valid with spec_valid; written by the helper before it posts)
  const short* slot_lut_v = nullptr;
  const uint8_t* sur_lut_v = nullptr;
  const uint32_t* slot_tag_v = nullptr;
  // surround cloud
  DevBuf<float4> sur_in, sur_out;
  DevBuf<uint32_t> sur_flag, sur_scan, sur_off, sur_cnt, sur_tiles;
  DevBuf<uint8_t> sur_valid;
  VoxelPipeline sur_vox;
  uint32_t n_surround = 0;
  PinBuf<uint32_t> h_sur;
  PinBuf<uint32_t> h_err;                            // raised by a fused kernel whose look-back gave up (checked behind process()'s synchronisation)
  hipStream_t st2 = nullptr, st3 = nullptr;         // the corner / surf map's update (the sub-map partition: st2 and the registration's stream)
  // The map update of a sweep (insertion, per-cube re-filtering, the new cube directory) is enqueued behind the registration but NOT waited
  // for: process() returns with the pose and the registered cloud while the update runs on st2 / st3; whoever needs the map next — the
  // next process(), the getters, a snapshot — finishes it first (finish_update: a wait that is normally over long before).
  // Its ~14 launches and copies cost the host ~60 us: a helper thread of this object enqueues them (behind an event of the
  // registration's stream) while the calling thread fetches the results.
  bool upd_pending = false;
  uint32_t upd_n_sub[2] = {0, 0};
  void finish_update();     // the calling thread: the helper has run its job; the update's result is in the host's cube directory
  // the same for the next process() call only: the helper's job is through the update and the prepared partition — on every fifth frame it
  // is still cutting the surround cloud then (createDownsizedMap: ~0.2 ms on a stream of its own, reading the updated map, which the next
  // sweep's partition only reads as well); everything else that touches the handle waits for the whole job (finish_update)
  void finish_update_front();
  void complete_update();   // the wait + the bookkeeping themselves (either thread, behind the helper's enqueue)
  void compute_surround(const MapWindow& w, const uint8_t* sur_lut);
  struct Helper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, quit = false;
    std::atomic<uint32_t> posted{0};   // bumped by post(): the helper spins on it for a while before it sleeps on cv
    std::atomic<bool> running{false};  // mirror of `busy` for wait()'s spin phase
    std::atomic<bool> front_done{true};   // the running job has left its front part (update + prepared partition) behind: finish_update_front()
    std::atomic<bool> front_failed{false};   // ... by an exception (wait() rethrows it)
    std::exception_ptr err;
    void post(std::function<void()> f);
    void wait();          // returns when the posted job has run; rethrows what it threw
    ~Helper();
  } helper;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  ~Mapper() {
    try { helper.wait(); } catch (...) {}
    if (st2) { (void)hipStreamSynchronize(st2); (void)hipStreamDestroy(st2); }
    if (st3) { (void)hipStreamSynchronize(st3); (void)hipStreamDestroy(st3); }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
  }

  // the sweep's clouds already in HBM (the linked entry point): packed float4 arrays of this device, read behind `ready` (may be NULL)
  struct DeviceInput {
    const float4* corner; uint32_t n_corner;
    const float4* surf; uint32_t n_surf;
    const float4* full; uint32_t n_full;
    hipEvent_t ready;
  };
  // dev given: corner_last / surf_last are ignored (may be NULL), full_res (may be NULL) only receives the registered cloud
  int process(const loamx_cloud* corner_last, const loamx_cloud* surf_last, loamx_cloud* full_res, const DeviceInput* dev = nullptr);
  // The map side of process() alone, for a sweep that has been registered elsewhere (the batched pipeline against a frozen map): stack,
  // down-size, insert into the cubes with the GIVEN pose and re-filter the touched cubes (BasicLaserMapping.cpp:512-593) — the
  // optimisation (:628-923) does not run, the pose is taken as transformTobeMapped.  The merge step of a map epoch.
  int insert(const loamx_cloud* corner_last, const loamx_cloud* surf_last, const float pose6[6]);
  const float* forced_pose = nullptr;   // (insert(): process() takes this as transformTobeMapped instead of transformAssociateToMap's)
  void load_cubes(const loamx_cloud* corner, const loamx_cloud* surf);
  int get_cubes(int which, loamx_cloud* out);
  int get_surround(loamx_cloud* out);
  void save_snapshot(const char* path);
  void load_snapshot(const char* path);

 private:
  void shift_counts(int axis, int dir);
  void ensure(TypeMap& t, uint32_t n_map_max, uint32_t n_in);
};

Mapper::Mapper(const loamx_map_config& c) : cfg(c), reg(c.device, 1) {
  reg.params.max_iterations = c.max_iterations;
  reg.params.delta_t_abort = c.delta_t_abort;
  reg.params.delta_r_abort = c.delta_r_abort;
  reg.params.corner_leaf = c.corner_filter_size;
  reg.params.surf_leaf = c.surf_filter_size;
  // the two feature types' map updates are independent: the corner map's runs on a stream of its own next to the surf map's
  // (process(): forked behind the registration, joined before the results are read)
  h_err.reserve(16);
  *h_err.p = 0u;
  st2 = create_stream(0);
  st3 = create_stream(0);
  LX_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  LX_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  for (int t = 0; t < 2; t++) {
    tm[t].vox.init(t == 0 ? st2 : st3);
    tm[t].hist.reserve(MCUBES + 16);
    tm[t].counters.p = tm[t].hist.p + MCUBES;   // (a view: the counters live behind the histogram so that both come down in one copy)
    tm[t].h_hist.reserve(MCUBES + 16);
    tm[t].out_off.reserve(130);
  }
  sur_vox.init(st3);
  sur_off.reserve(4);
  sur_cnt.reserve(16);
  sur_tiles.reserve(SCAN_SCRATCH_WORDS);
  LX_HIP(hipMemsetAsync(sur_tiles.p, 0, sizeof(uint32_t) * sur_tiles.cap, reg.stream()));
}

// the reference's pointer-swap loops (:311-441) applied to the host count directory: contents move by one cube along
// `axis` (dir=+1 towards higher indices) and the vacated layer is cleared
void Mapper::shift_counts(int axis, int dir) {
  const int n[3] = {MW, MH, MD};
  for (int t = 0; t < 2; t++) {
    std::vector<uint32_t> nc(MCUBES, 0);
    for (int k = 0; k < MD; k++)
      for (int j = 0; j < MH; j++)
        for (int i = 0; i < MW; i++) {
          int ijk[3] = {i, j, k};
          ijk[axis] -= dir;   // source cube
          if (ijk[axis] < 0 || ijk[axis] >= n[axis]) continue;
          nc[i + MW * j + MW * MH * k] = tm[t].cube_cnt[ijk[0] + MW * ijk[1] + MW * MH * ijk[2]];
        }
    tm[t].cube_cnt.swap(nc);
  }
}

void Mapper::ensure(TypeMap& t, uint32_t n_map_max, uint32_t n_in) {
  hipStream_t st = reg.stream();
  // A rolling map grows with every sweep that sees something new, and growing a device buffer is a hipFree + hipMalloc: 1.5-2 ms with the
  // device drained, in front of a 0.3 ms call (measured: four such stalls in 110 sweeps of the live bench with 25 % head room).  Room
  // is taken in large steps instead — twice what the map needs when a buffer has to move (at least 128 k points), so a map that grows
  // steadily moves its buffers O(log) times — the sub-map's index and the surround cloud's buffers included.
  if (n_map_max + 1 > t.room) t.room = std::max<uint32_t>(2u * (n_map_max + 1), 1u << 17);   // (a small handle stays small: ~13 MB per feature type at the floor)
  const uint32_t room = t.room;
  for (int b = 0; b < 2; b++) {
    t.pts[b].reserve(room, st, b == t.cur);
    t.tags[b].reserve(room, st, b == t.cur);
  }
  {   // chains of the fused split / insert kernels: one word per tile, cleared when the buffer (re)appears
    const size_t need = std::max<size_t>(room / MS_TILE, n_in / 256) + 4;
    if (need > t.chain_stride) {
      t.chain_stride = need + need / 2;
      t.chain.reserve(3 * t.chain_stride);
      LX_HIP(hipMemsetAsync(t.chain.p, 0, sizeof(unsigned long long) * t.chain.cap, st));
      t.chain_epoch = 0;
    }
  }
  t.fin.reserve(room); t.fin_seg.reserve(room); t.fin_valid.reserve(room);
  t.filt.reserve(room);
  t.vox.reserve(room, 126);
  (&t == &tm[0] ? reg.corner_index : reg.surf_index).reserve_points(room);
}

namespace {
// LOAMX_MAP_TRACE: host-side stamps of process() (us since entry), averaged over 50 calls and printed — where the host's share of a sweep goes
struct MapTrace {
  bool on;
  int n = 0, calls = 0;
  double t0 = 0;
  const char* name[16];
  double at[16], sum[16] = {}, worst = 0;
  static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  MapTrace() : on(getenv("LOAMX_MAP_TRACE") != nullptr) {}
  void begin() { if (on) { n = 0; t0 = now(); } }
  void mark(const char* what) { if (on && n < 16) { name[n] = what; at[n++] = now() - t0; } }
  void end() {
    if (!on) return;
    for (int k = 0; k < n; k++) sum[k] += at[k];
    if (n) worst = std::max(worst, at[n - 1]);
    if (n && at[n - 1] > 700.0) {
      fprintf(stderr, "[map trace, slow call]");
      for (int k = 0; k < n; k++) fprintf(stderr, " %s %.1f", name[k], at[k]);
      fprintf(stderr, "\n");
    }
    if (++calls < 50) return;
    fprintf(stderr, "[map trace, mean of %d calls]", calls);
    for (int k = 0; k < n; k++) { fprintf(stderr, " %s %.1f", name[k], sum[k] / calls); sum[k] = 0; }
    fprintf(stderr, " (slowest call %.1f)\n", worst);
    calls = 0; worst = 0;
  }
};
}  // namespace

int Mapper::process(const loamx_cloud* corner_last, const loamx_cloud* surf_last, loamx_cloud* full_res, const DeviceInput* dev) {
  TraceRange trace_range("loamx:mapping:process");
  static thread_local MapTrace tr;
  tr.begin();
  if (!dev) {
    check_cloud(corner_last, false);
    check_cloud(surf_last, false);
  }
  if (full_res) check_cloud(full_res, false);
  LX_HIP(hipSetDevice(cfg.device));
  finish_update_front();
  tr.mark("finished_update");
  frame_count++;
  if (frame_count < 1) return LOAMX_SKIPPED;   // _stackFrameNum = 1 (:269-274)
  frame_count = 0;

  if (forced_pose) tobe.set(forced_pose);
  else transform_associate_to_map(sum, bef, aft, incre, tobe);
  Plan plan;
  make_plan(tobe, /*may_shift=*/true, plan);
  const int nvalid = plan.nvalid;
  uint32_t n_sub[2] = {plan.n_sub[0], plan.n_sub[1]};
  last_sub[0] = n_sub[0];
  last_sub[1] = n_sub[1];
  tr.mark("planned");
  MapWindow w;
  for (int a = 0; a < 3; a++) w.cen[a] = cen[a];
  const uint32_t n_in[2] = {dev ? dev->n_corner : corner_last->count, dev ? dev->n_surf : surf_last->count};
  const bool want_full = dev ? dev->n_full != 0 : (full_res && full_res->count);

  // a partition prepared for the predicted pose is this sweep's own if the plans agree entry by entry (and the sweep fits the room the
  // buffers were given: ensure() must not move them)
  // ... nor grow a look-back chain: the prepared side sized them for sweeps of at most spec_room points
  const bool adopt = spec_valid && plan.same_as(spec_plan) && tm[0].n + n_in[0] + 65 <= tm[0].room && tm[1].n + n_in[1] + 65 <= tm[1].room &&
                     n_in[0] <= spec_room[0] && n_in[1] <= spec_room[1];
  if (spec_valid) (adopt ? spec_hits : spec_misses)++;
  spec_valid = false;
  if (adopt) tab_cur ^= 1;   // (the prepared tables are the current ones now; slot_lut_v / sur_lut_v / slot_tag_v point into them already)

  // the sweep's clouds and the guess go up first: the copies run while this thread is still enqueuing the partition (the chain
  // of short launches below is bound by the host's launch rate, not by the device)
  {
    float g6[6];
    tobe.get(g6);
    if (dev) {
      if (dev->ready) LX_HIP(hipStreamWaitEvent(reg.stream(), dev->ready, 0));
      const float4* c[1] = {dev->corner};
      const float4* sf[1] = {dev->surf};
      const float4* fl[1] = {dev->full};
      reg.upload_device(1, c, &dev->n_corner, sf, &dev->n_surf, dev->n_full ? fl : nullptr, &dev->n_full, g6);
    } else {
      reg.upload(1, corner_last, surf_last, full_res, g6, false);
    }
  }
  tr.mark("uploaded");
  if (!adopt) {
    helper.wait();   // (a surround cloud still being cut reads the look-up tables the partition is about to rewrite; ensure() may move the map)
    enqueue_partition(plan, tab_cur, n_in);
  }
  tr.mark(adopt ? "partition_adopted" : "partition+index");

  // ---- registration against the sub-map (guard + iterations inside Registrar::run_async)
  reg.early_exit = true;   // process() is blocking
  // (insert(): the caller's pose goes into the map verbatim — a pipeline pose has had its IMU blend already)
  const bool imu_blend = !imu_history.empty() && !forced_pose;
  reg.defer_full = imu_blend;
  reg.run_async();
  last_optimized = reg.submap_sufficient();
  if (imu_blend) {
    // transformUpdate with IMU data (:171-200) changes transformTobeMapped before the new features are inserted into the map
    // and before the full-resolution cloud is registered: blend on the host, hand the pose back to the device
    float p6[6];
    reg.download(p6, nullptr);
    if (last_optimized) imu_blend_pose(p6);
    reg.finish_with_poses(p6);
  }

  tr.mark("registered");
  // ---- map insertion + per-cube re-filtering: corners on st2, surfs on st3, side by side behind the registration — and NOT in front of
  // its results (see upd_pending)
  LX_HIP(hipEventRecord(ev_fork, reg.stream()));
  if (want_full && full_res) reg.download_full_res_async(0, full_res);   // (lands while the map is updated)
  for (int t = 0; t < 2; t++) upd_n_sub[t] = n_sub[t];
  const uint32_t n_sub0 = n_sub[0], n_sub1 = n_sub[1], n_in0 = n_in[0], n_in1 = n_in[1];
  // createDownsizedMap (:242-264) is due on every 5th processed frame: the helper goes on to finish the update and cut the surround cloud
  if (!forced_pose) map_frame_count++;   // (insert() is not a processed frame: no surround cloud is due)
  fresh_map = false;
  const bool surround_due = map_frame_count >= 5;
  if (surround_due) { map_frame_count = 0; fresh_map = true; }
  upd_pending = true;
  const bool speculate = !forced_pose && spec_enabled;
  {
    std::lock_guard<std::mutex> lk(helper.mu);
    spec_in.ready = false;
    spec_in.ok = false;
  }
  struct SpecRelease {   // whatever happens below, the helper must not wait for this sweep's results for ever
    Mapper* m;
    ~SpecRelease() {
      std::lock_guard<std::mutex> lk(m->helper.mu);
      m->spec_in.ready = true;
      m->helper.cv.notify_all();
    }
  } spec_release{this};
  helper.post([this, n_sub0, n_sub1, n_in0, n_in1, nvalid, w, surround_due, speculate]() {
  struct FrontDone {   // (also when the front part throws: the caller's next wait must end — it goes on to Helper::wait(), which rethrows)
    Helper& h;
    ~FrontDone() {
      if (std::uncaught_exceptions() > 0) h.front_failed.store(true, std::memory_order_release);
      h.front_done.store(true, std::memory_order_release);
    }
  };
  std::unique_ptr<FrontDone> front(new FrontDone{helper});
  static const bool htrace = getenv("LOAMX_MAP_TRACE") != nullptr;   // (helper-side stamps: a job slower than 1 ms says where it spent its time)
  double hs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hs[0] = htrace ? MapTrace::now() : 0.0;
  LX_HIP(hipSetDevice(cfg.device));
  const uint32_t n_sub[2] = {n_sub0, n_sub1}, n_in[2] = {n_in0, n_in1};
  // (trace only: which call of the update's enqueue took longest — a job that needs milliseconds to ENQUEUE is a call that blocked)
  double ht_prev = hs[0], ht_worst = 0;
  const char* ht_what = "";
  auto HT = [&](const char* what) { if (htrace) { const double n = MapTrace::now(); if (n - ht_prev > ht_worst) { ht_worst = n - ht_prev; ht_what = what; } ht_prev = n; } };
  LX_HIP(hipStreamWaitEvent(st2, ev_fork, 0));
  LX_HIP(hipStreamWaitEvent(st3, ev_fork, 0));
  HT("stream waits");
  for (int t = 1; t >= 0; t--) {   // (the surf map first: its chain is the longer one, and the host needs ~80 us to enqueue either)
    TypeMap& T = tm[t];
    hipStream_t st = t == 0 ? st2 : st3;
    const int nxt = 1 - T.cur;
    const uint32_t n_old = n_sub[t], n_slots = n_in[t], n_fin = n_old + n_slots;
    // (the sub-map's points, cube slots and valid flags are in fin / fin_seg / fin_valid already: the partition wrote them there)
    if (n_slots) {
      const uint32_t nb = (n_slots + 255) / 256;
      if (++T.chain_epoch >= (1u << 30) - 2u) {
        LX_HIP(hipMemsetAsync(T.chain.p, 0, sizeof(unsigned long long) * T.chain.cap, st));
        T.chain_epoch = 1;
      }
      hipLaunchKernelGGL(k_map_insert, dim3(nb), dim3(256), 0, st, reg.d_ds_points(), reg.d_ds_offsets(), t, n_slots, reg.d_poses(), w,
                         slot_lut_v, n_old, T.fin.p, T.fin_seg.p, T.fin_valid.p, T.counters.p + 3, T.pts[nxt].p, T.tags[nxt].p,
                         T.chain.p + 2 * T.chain_stride, (unsigned long long)T.chain_epoch, T.counters.p, h_err.p);
    } else {
      LX_HIP(hipMemsetAsync(T.counters.p + 4, 0, sizeof(uint32_t) * 2, st));
    }
    HT("k_map_insert");
    const uint32_t nslots = (uint32_t)std::max(nvalid, 1);
    const float inv = 1.0f / (t == 0 ? cfg.corner_filter_size : cfg.surf_filter_size);
    T.vox.compute_ijk(T.fin.p, T.fin_valid.p, n_fin, nullptr, nslots, inv, inv, T.fin_seg.p);
    HT("compute_ijk");
    T.vox.sort_reduce(T.fin.p, T.fin_valid.p, n_fin, nullptr, nslots, T.filt.p, T.out_off.p, T.fin_seg.p);
    HT("sort_reduce");
    const uint32_t max_f = n_fin ? n_fin : 1;
    hipLaunchKernelGGL(k_map_append_filtered, dim3((max_f + 255) / 256), dim3(256), 0, st, T.filt.p, T.out_off.p, nslots, slot_tag_v,
                       max_f, T.counters.p + 3, T.counters.p + 5, T.pts[nxt].p, T.tags[nxt].p, T.counters.p + 6, T.hist.p);
    const uint32_t max_new = T.n + n_slots + 1;
    hipLaunchKernelGGL(k_map_hist, dim3(std::min<uint32_t>((max_new + 2047) / 2048, 256u)), dim3(256), 0, st, T.tags[nxt].p, T.counters.p + 6, max_new, w, T.hist.p);
    HT("append + hist");
    store_to_pinned_u32(T.h_hist.p, T.hist.p, MCUBES + 16, st);   // (histogram + the counters behind it, by a kernel: pinned_copy.hpp — the copy call blocked for milliseconds now and then)
    HT("histogram copy");
  }
  static const bool trace = getenv("LOAMX_MAP_TRACE") != nullptr;
  const double t0 = trace ? MapTrace::now() : 0.0;
  hs[1] = t0;   // update enqueued
  if (surround_due) complete_update();
  const double t1 = trace ? MapTrace::now() : 0.0;
  // (the surround cloud is cut LAST, behind the prepared partition: the next sweep waits for the partition, not for the cloud — round 6; the
  // cloud is cut with THIS sweep's look-up table, the preparation below points sur_lut_v at the next one's)
  const uint8_t* sur_lut_now = sur_lut_v;
  if (speculate) {   // the next sweep's partition + index for the predicted pose (see spec_plan)
    complete_update();
    hs[2] = htrace ? MapTrace::now() : 0.0;   // update complete
    SpecInputs in;
    {
      std::unique_lock<std::mutex> lk(helper.mu);
      helper.cv.wait(lk, [this]() { return spec_in.ready; });
      in = spec_in;
    }
    hs[3] = htrace ? MapTrace::now() : 0.0;   // the caller's results are in
    if (in.ok) {
      float p6[6];
      for (int k = 0; k < 6; k++) p6[k] = in.sum6[k] + (in.sum6[k] - in.sum_prev6[k]);
      HTwist sum_pred, incre_tmp, tobe_pred;
      sum_pred.set(p6);
      transform_associate_to_map(sum_pred, in.bef, in.aft, incre_tmp, tobe_pred);
      Plan sp;
      if (make_plan(tobe_pred, /*may_shift=*/false, sp)) {
        enqueue_partition(sp, tab_cur ^ 1, in.n_in_room);
        spec_room[0] = in.n_in_room[0];   // (the look-back chains of k_map_insert were sized for sweeps of at most this many points)
        spec_room[1] = in.n_in_room[1];
        spec_plan = std::move(sp);
        spec_valid = true;
      }
    }
  }
  hs[4] = htrace ? MapTrace::now() : 0.0;     // partition prepared (enqueued)
  if (htrace && hs[4] - hs[0] > 1000.0)
    fprintf(stderr, "[map trace, slow helper job] update enqueued %.0f us (longest call: %s, %.0f us), update complete %.0f, caller's results in %.0f, partition prepared %.0f\n",
            hs[1] - hs[0], ht_what, ht_worst, hs[2] ? hs[2] - hs[0] : -1.0, hs[3] ? hs[3] - hs[0] : -1.0, hs[4] - hs[0]);
  front.reset();   // the next process() call may start
  if (surround_due) {
    const double t2 = trace ? MapTrace::now() : 0.0;
    compute_surround(w, sur_lut_now);
    if (trace) fprintf(stderr, "[map trace, helper] update completed %.1f us after its enqueue, surround cloud %.1f us\n", t1 - t0, MapTrace::now() - t2);
  }
  });
  tr.mark("update_posted");

  // ---- results (the registration's stream: poses, statistics, the registered cloud's copy — not the map update)
  float pose6[6];
  int stats4[4];
  reg.sync();
  tr.mark("synced");
  // the partition this registration read was split by a fused kernel with a look-back: had a tile given up, the sub-map would be
  // incomplete — fail THIS sweep, not the next call (the word is host-visible; the split precedes the registration on its stream)
  if (*(volatile uint32_t*)h_err.p) { *h_err.p = 0u; throw Error(LOAMX_E_HIP, "map partition: a tile's look-back gave up waiting for the tiles before it"); }
  reg.download(pose6, stats4);
  LX_HIP(hipGetLastError());
  SweepStats ss;
  reg.download_stats(&ss);
  last_stats = ss;
  int rc = LOAMX_OK;
  if (last_optimized) {
    // transformUpdate (:171-203; the IMU blend, if any, is already in pose6): only reached when the optimisation ran (:628-629)
    tobe.set(pose6);
    bef = sum;
    aft = tobe;
  }
  if (speculate) {   // what the prediction needs of this sweep (the helper waits for it behind the update)
    std::lock_guard<std::mutex> lk(helper.mu);
    sum.get(spec_in.sum6);
    for (int k = 0; k < 6; k++) spec_in.sum_prev6[k] = have_sum_prev ? sum_prev6[k] : spec_in.sum6[k];
    spec_in.bef = bef;
    spec_in.aft = aft;
    spec_in.n_in_room[0] = 2 * n_in[0] + 1024;
    spec_in.n_in_room[1] = 2 * n_in[1] + 1024;
    spec_in.ok = true;
  }
  if (!forced_pose) { sum.get(sum_prev6); have_sum_prev = true; }
  if (want_full && full_res) {
    int r = reg.download_full_res(0, full_res);
    if (r != LOAMX_OK) rc = r;
  }

  tr.mark("results");
  tr.end();
  return rc;
}

bool Mapper::make_plan(const HTwist& pose_tw, bool may_shift, Plan& out) {
  const Pose guess = pose_tw.pose();
  // pointOnYAxis (:294-298)
  float py[3] = {0.f, 10.f, 0.f};
  to_map(guess, py[0], py[1], py[2]);

  int cc[3] = {cube_abs(pose_tw.pos.x) + cen[0], cube_abs(pose_tw.pos.y) + cen[1], cube_abs(pose_tw.pos.z) + cen[2]};
  const int dims[3] = {MW, MH, MD};
  for (int a = 0; a < 3; a++) {
    if (!may_shift && (cc[a] < 3 || cc[a] >= dims[a] - 3)) return false;
    while (cc[a] < 3) { shift_counts(a, +1); cc[a]++; cen[a]++; }
    while (cc[a] >= dims[a] - 3) { shift_counts(a, -1); cc[a]--; cen[a]--; }
  }
  for (int a = 0; a < 3; a++) out.cen[a] = cen[a];

  // 5x5x5 neighbourhood, FOV test on the cube corners (:443-500); valid list in the reference's i->j->k order
  out.lut.assign(MCUBES, -1);
  out.slut.assign(MCUBES, 0);
  out.stag.clear();
  int nvalid = 0;
  for (int i = cc[0] - 2; i <= cc[0] + 2; i++)
    for (int j = cc[1] - 2; j <= cc[1] + 2; j++)
      for (int k = cc[2] - 2; k <= cc[2] + 2; k++) {
        if (i < 0 || i >= MW || j < 0 || j >= MH || k < 0 || k >= MD) continue;
        const float centerX = 50.0f * (i - cen[0]), centerY = 50.0f * (j - cen[1]), centerZ = 50.0f * (k - cen[2]);
        bool in_fov = false;
        for (int ii = -1; ii <= 1; ii += 2)
          for (int jj = -1; jj <= 1; jj += 2)
            for (int kk = -1; kk <= 1; kk += 2) {
              const float cx = centerX + 25.0f * ii, cy = centerY + 25.0f * jj, cz = centerZ + 25.0f * kk;
              const float ax = pose_tw.pos.x - cx, ay = pose_tw.pos.y - cy, az = pose_tw.pos.z - cz;
              const float s1 = ax * ax + ay * ay + az * az;
              const float bx = py[0] - cx, by = py[1] - cy, bz = py[2] - cz;
              const float s2 = bx * bx + by * by + bz * bz;
              const float check1 = 100.0f + s1 - s2 - 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
              const float check2 = 100.0f + s1 - s2 + 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
              if (check1 < 0 && check2 > 0) in_fov = true;
            }
        const int idx = i + MW * j + MW * MH * k;
        if (in_fov) {
          out.lut[idx] = (short)nvalid++;
          out.stag.push_back(pack_tag(i - cen[0], j - cen[1], k - cen[2]));
        }
        out.slut[idx] = 1;
      }
  if (out.stag.empty()) out.stag.push_back(0);
  LX_REQUIRE(out.stag.size() <= 128, "internal: more than 125 valid cubes");
  out.nvalid = nvalid;
  for (int t = 0; t < 2; t++) {
    out.n_sub[t] = 0;
    for (int idx = 0; idx < MCUBES; idx++)
      if (out.lut[idx] >= 0) out.n_sub[t] += tm[t].cube_cnt[idx];
  }
  return true;
}

void Mapper::enqueue_partition(const Plan& plan, int set, const uint32_t n_in_room[2]) {
  hipStream_t st = reg.stream();
  // the three tables travel as one block through pinned memory owned by this object (its previous copy from this set has completed:
  // a whole sweep with its waits lies in between) — no host wait here
  {
    const size_t o_sur = sizeof(short) * MCUBES, o_tag = (o_sur + MCUBES + 3) & ~(size_t)3, bytes = o_tag + sizeof(uint32_t) * 128;
    h_tables[set].reserve(bytes);
    d_tables[set].reserve(bytes);
    memcpy(h_tables[set].p, plan.lut.data(), sizeof(short) * MCUBES);
    memcpy(h_tables[set].p + o_sur, plan.slut.data(), MCUBES);
    memcpy(h_tables[set].p + o_tag, plan.stag.data(), sizeof(uint32_t) * plan.stag.size());
    LX_HIP(hipMemcpyAsync(d_tables[set].p, h_tables[set].p, bytes, hipMemcpyHostToDevice, st));
    slot_lut_v = (const short*)d_tables[set].p;
    sur_lut_v = (const uint8_t*)(d_tables[set].p + o_sur);
    slot_tag_v = (const uint32_t*)(d_tables[set].p + o_tag);
  }
  MapWindow w;
  for (int a = 0; a < 3; a++) w.cen[a] = plan.cen[a];
  // ---- partition the map: sub-map | rest | dropped (the two types side by side, as in the update)
  for (int t = 0; t < 2; t++) ensure(tm[t], tm[t].n + n_in_room[t] + 64, n_in_room[t]);
  LX_HIP(hipEventRecord(ev_fork, reg.stream()));   // (behind the look-up tables' copy)
  LX_HIP(hipStreamWaitEvent(st2, ev_fork, 0));
  for (int t = 0; t < 2; t++) {
    TypeMap& T = tm[t];
    hipStream_t stt = t == 0 ? st2 : reg.stream();
    const int cur = T.cur, nxt = 1 - cur;
    if (T.n) {
      if (++T.chain_epoch >= (1u << 30) - 2u) {   // (the chains' words carry a 30-bit epoch: cleared on the stream before it comes round)
        LX_HIP(hipMemsetAsync(T.chain.p, 0, sizeof(unsigned long long) * T.chain.cap, stt));
        T.chain_epoch = 1;
      }
      hipLaunchKernelGGL(k_map_split, dim3((T.n + MS_TILE - 1) / MS_TILE), dim3(256), 0, stt, T.pts[cur].p, T.tags[cur].p, T.n, w, slot_lut_v, T.fin.p,
                         T.fin_seg.p, T.fin_valid.p, T.pts[nxt].p, T.tags[nxt].p, (t == 0 ? reg.corner_index : reg.surf_index).d_bounds(),
                         T.chain.p, T.chain.p + T.chain_stride, (unsigned long long)T.chain_epoch, T.counters.p, h_err.p);
    } else {
      LX_HIP(hipMemsetAsync(T.counters.p, 0, sizeof(uint32_t) * 4, stt));
    }
  }
  // ... and their grid indices (the corner sub-map's behind its partition on st2)
  reg.set_submap_device_split(tm[0].fin.p, plan.n_sub[0], st2, tm[1].fin.p, plan.n_sub[1], /*bounds_done=*/true);
  LX_HIP(hipEventRecord(ev_join, st2));
  LX_HIP(hipStreamWaitEvent(reg.stream(), ev_join, 0));
}

// createDownsizedMap (:242-264): the surround cloud, cut from the UPDATED map.  Runs on the helper thread behind the update (st3; the
// two host waits in here are why: they would sit in front of the caller's next sweep otherwise)
void Mapper::compute_surround(const MapWindow& w, const uint8_t* sur_lut) {
  hipStream_t st = st3;
  const uint32_t nc = tm[0].n, nsf = tm[1].n, ntot = nc + nsf;
  const uint32_t room2 = std::max(tm[0].room + tm[1].room, ntot + 2), room1 = std::max(std::max(tm[0].room, tm[1].room), std::max(nc, nsf) + 2);   // (ensure(): room in large steps)
  sur_in.reserve(room2);
  sur_out.reserve(room2);
  sur_flag.reserve(room1);
  sur_scan.reserve(room1);
  sur_valid.reserve(room2);
  sur_vox.reserve(room2, 2);
  static const bool trace = getenv("LOAMX_MAP_TRACE") != nullptr;
  const double tq0 = trace ? MapTrace::now() : 0.0;
  double tq1 = 0;
  for (int t = 0; t < 2; t++) {
    TypeMap& T = tm[t];
    const uint32_t n = T.n;
    uint32_t* cnt = sur_cnt.p + 4 * t;   // [n, total]
    if (n) {
      const uint32_t nb = (n + 255) / 256;
      hipLaunchKernelGGL(k_map_surround_flags, dim3(nb), dim3(256), 0, st, T.tags[T.cur].p, T.counters.p + 6, n, w, sur_lut, sur_flag.p);
      exclusive_scan_u32_n(sur_flag.p, sur_scan.p, sur_tiles.p, cnt, n, st);
      hipLaunchKernelGGL(k_map_compact, dim3(nb), dim3(256), 0, st, T.pts[T.cur].p, sur_flag.p, sur_scan.p, n, 0u,
                         t == 0 ? (const uint32_t*)nullptr : (const uint32_t*)(sur_cnt.p + 1), sur_in.p);
    } else {
      LX_HIP(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * 2, st));
    }
  }
  if (ntot) {
    hipLaunchKernelGGL(k_map_surround_valid, dim3((ntot + 255) / 256), dim3(256), 0, st, sur_valid.p, ntot, sur_cnt.p + 1, sur_cnt.p + 5);
    const float inv = 1.0f / cfg.corner_filter_size;   // the corner filter, not the map filter (:261)
    // (offsets up and the result's size down through pinned words of this object: true asynchronous copies, one wait)
    h_sur.reserve(8);
    h_sur.p[0] = 0u; h_sur.p[1] = ntot;
    LX_HIP(hipMemcpyAsync(sur_off.p, h_sur.p, sizeof(uint32_t) * 2, hipMemcpyHostToDevice, st));
    sur_vox.compute_ijk(sur_in.p, sur_valid.p, ntot, sur_off.p, 1, inv, inv);
    sur_vox.sort_reduce(sur_in.p, sur_valid.p, ntot, sur_off.p, 1, sur_out.p, sur_off.p + 2);
    store_to_pinned_u32(h_sur.p + 2, sur_off.p + 2, 2, st);
    if (trace) tq1 = MapTrace::now();
    LX_HIP(hipStreamSynchronize(st));
    if (trace && MapTrace::now() - tq0 > 600.0) fprintf(stderr, "[map trace, surround] enqueue %.1f us, wait %.1f us (%u points)\n", tq1 - tq0, MapTrace::now() - tq1, ntot);
    sur_vox.check();
    n_surround = h_sur.p[3];
  } else {
    n_surround = 0;
  }
}

void Mapper::Helper::post(std::function<void()> f) {
  std::unique_lock<std::mutex> lk(mu);
  if (!th.joinable())
    th = std::thread([this]() {
      std::unique_lock<std::mutex> l(mu);
      for (;;) {
        // a job arrives every ~0.7 ms while sweeps flow: waking from a condition variable costs tens of microseconds as a rule and ~10 ms
        // when the thread has lost its time slice (measured: one such call per ~100 sweeps on some hosts, profiles/r06_ab.md section 8),
        // so the helper spins for up to 2 ms on the post counter before it goes to sleep — an idle handle costs nothing after that
        if (!(quit || (busy && job))) {
          const uint32_t seen = posted.load(std::memory_order_acquire);
          l.unlock();
          const auto t_in = std::chrono::steady_clock::now();
          for (unsigned spins = 0; posted.load(std::memory_order_acquire) == seen;) {
            if ((++spins & 255u) == 0u && std::chrono::steady_clock::now() - t_in > std::chrono::milliseconds(2)) break;
            __builtin_ia32_pause();
          }
          l.lock();
        }
        cv.wait(l, [this]() { return quit || (busy && job); });
        if (quit) return;
        std::function<void()> j = std::move(job);
        job = nullptr;
        l.unlock();
        std::exception_ptr e;
        try { j(); } catch (...) { e = std::current_exception(); }
        l.lock();
        err = e;
        busy = false;
        running.store(false, std::memory_order_release);
        cv.notify_all();
      }
    });
  cv.wait(lk, [this]() { return !busy; });
  job = std::move(f);
  busy = true;
  front_failed.store(false, std::memory_order_release);
  front_done.store(false, std::memory_order_release);
  running.store(true, std::memory_order_release);
  posted.fetch_add(1, std::memory_order_release);
  cv.notify_all();
}
void Mapper::Helper::wait() {
  {   // (the job is a few hundred microseconds of enqueueing as a rule: spin for it before sleeping on the condition variable)
    const auto t_in = std::chrono::steady_clock::now();
    for (unsigned spins = 0; running.load(std::memory_order_acquire);) {
      if ((++spins & 255u) == 0u && std::chrono::steady_clock::now() - t_in > std::chrono::milliseconds(2)) break;
      __builtin_ia32_pause();
    }
  }
  std::unique_lock<std::mutex> lk(mu);
  cv.wait(lk, [this]() { return !busy; });
  if (err) { std::exception_ptr e = err; err = nullptr; std::rethrow_exception(e); }
}
Mapper::Helper::~Helper() {
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this]() { return !busy; });
    quit = true;
    posted.fetch_add(1, std::memory_order_release);
    cv.notify_all();
  }
  if (th.joinable()) th.join();
}

void Mapper::finish_update_front() {
  const auto t_in = std::chrono::steady_clock::now();
  for (unsigned spins = 0; !helper.front_done.load(std::memory_order_acquire);) {
    if ((++spins & 255u) == 0u && std::chrono::steady_clock::now() - t_in > std::chrono::milliseconds(2)) { helper.wait(); break; }
    __builtin_ia32_pause();
  }
  // (a job that has ended or whose front part threw: take what it threw — no wait in the first case, the job's unwinding in the second)
  if (!helper.running.load(std::memory_order_acquire) || helper.front_failed.load(std::memory_order_acquire)) helper.wait();
  complete_update();   // (a job that neither cut a surround cloud nor prepared a partition has only enqueued the update)
}

void Mapper::finish_update() {
  helper.wait();   // (the launches are enqueued by the helper thread; on a surround frame it has completed the update as well)
  complete_update();
}

void Mapper::complete_update() {
  if (!upd_pending) return;
  upd_pending = false;
  LX_HIP(hipSetDevice(cfg.device));
  spin_sync(st2);   // (polling, not the runtime's blocking wait: common.h)
  spin_sync(st3);   // (the histogram / counter copies are the last thing on either stream)
  for (int t = 0; t < 2; t++) tm[t].vox.check();   // (a timed-out wait inside the per-cube voxel kernel must not corrupt the map silently)
  if (*(volatile uint32_t*)h_err.p) { *h_err.p = 0u; throw Error(LOAMX_E_HIP, "map update: a tile's look-back gave up waiting for the tiles before it"); }
  for (int t = 0; t < 2; t++) {
    TypeMap& T = tm[t];
    LX_REQUIRE(T.n == 0 || T.h_hist.p[MCUBES + 1] == upd_n_sub[t], "internal: sub-map size differs from the host cube directory");
    T.cur = 1 - T.cur;
    T.n = T.h_hist.p[MCUBES + 6];
    for (int idx = 0; idx < MCUBES; idx++) T.cube_cnt[idx] = T.h_hist.p[idx];
  }
}

int Mapper::insert(const loamx_cloud* corner_last, const loamx_cloud* surf_last, const float pose6[6]) {
  // process() with zero Gauss-Newton launches and a forced pose: the map side alone (:512-593).  Everything of the handle that is NOT
  // the map is put back when the call ends, also when it throws: the iteration limit, the frame counter (an insertion is not a
  // processed frame) and the transforms — the handle's transformTobeMapped / BefMapped / AftMapped belong to its own process() calls
  struct Restore {
    Mapper* m; int it; long fc; HTwist tobe, bef, aft; bool opt;
    ~Restore() {
      m->reg.params.max_iterations = it; m->forced_pose = nullptr; m->frame_count = fc;
      m->tobe = tobe; m->bef = bef; m->aft = aft; m->last_optimized = opt;
    }
  } restore{this, reg.params.max_iterations, frame_count, tobe, bef, aft, last_optimized};
  reg.params.max_iterations = 0;   // no Gauss-Newton launch: the registrar stacks and down-sizes with the pose it is given
  forced_pose = pose6;
  frame_count = 0;                 // (every call inserts: _stackFrameNum counts sweeps that go through process())
  return process(corner_last, surf_last, nullptr);
}

void Mapper::load_cubes(const loamx_cloud* corner, const loamx_cloud* surf) {
  finish_update();
  spec_valid = false;   // (a partition prepared for the old map is of no use)
  LX_HIP(hipSetDevice(cfg.device));
  hipStream_t st = reg.stream();
  const loamx_cloud* cl[2] = {corner, surf};
  for (int t = 0; t < 2; t++) {
    if (!cl[t] || !cl[t]->count) continue;
    check_cloud(cl[t], false);
    TypeMap& T = tm[t];
    std::vector<float4> p(cl[t]->count), keep;
    std::vector<uint32_t> tags;
    pack_cloud(cl[t], p.data());
    for (const float4& q : p) {
      const int ia = cube_abs(q.x), ja = cube_abs(q.y), ka = cube_abs(q.z);
      const int I = ia + cen[0], J = ja + cen[1], K = ka + cen[2];
      if (I < 0 || I >= MW || J < 0 || J >= MH || K < 0 || K >= MD) continue;
      keep.push_back(q);
      tags.push_back(pack_tag(ia, ja, ka));
      T.cube_cnt[I + MW * J + MW * MH * K]++;
    }
    const uint32_t add = (uint32_t)keep.size();
    ensure(T, T.n + add + 64, 0);
    if (add) {
      LX_HIP(hipMemcpyAsync(T.pts[T.cur].p + T.n, keep.data(), sizeof(float4) * add, hipMemcpyHostToDevice, st));
      LX_HIP(hipMemcpyAsync(T.tags[T.cur].p + T.n, tags.data(), sizeof(uint32_t) * add, hipMemcpyHostToDevice, st));
      LX_HIP(hipStreamSynchronize(st));
    }
    T.n += add;
  }
}

int Mapper::get_cubes(int which, loamx_cloud* out) {
  finish_update();
  LX_REQUIRE(which == 0 || which == 1, "which must be 0 (corner) or 1 (surf)");
  check_cloud(out, false);
  LX_HIP(hipSetDevice(cfg.device));
  TypeMap& T = tm[which];
  std::vector<float4> tmp(T.n);
  if (T.n) LX_HIP(hipMemcpy(tmp.data(), T.pts[T.cur].p, sizeof(float4) * T.n, hipMemcpyDeviceToHost));
  return unpack_cloud(tmp.data(), T.n, out);
}

// ---- map snapshot on disk: the rolling map (both feature types, in storage order — the order that
// fixes the sub-map order of BasicLaserMapping.cpp:503-509), the cube window, the frame counters (:269-274, :245-249) and the
// five transforms.  A handle restored from a snapshot continues bit for bit like the one that wrote it.
// File: "LOAMXMAP" | u32 version = 1 | i32 cen[3] | i64 frame_count, map_frame_count | f32 leaf corner, surf |
//       f32[6] x 5 (sum, incre, tobe, bef, aft) | u32 n_corner, n_surf | float4 corner[n_corner] | float4 surf[n_surf]
namespace {
struct SnapshotHeader {
  char magic[8];
  uint32_t version;
  int32_t cen[3];
  int64_t frame_count, map_frame_count;
  float corner_leaf, surf_leaf;
  float transforms[5][6];
  uint32_t n[2];
};
}  // namespace

void Mapper::save_snapshot(const char* path) {
  finish_update();
  LX_REQUIRE(path && *path, "NULL path");
  LX_HIP(hipSetDevice(cfg.device));
  LX_HIP(hipStreamSynchronize(reg.stream()));
  SnapshotHeader h;
  memcpy(h.magic, "LOAMXMAP", 8);
  h.version = 1;
  for (int k = 0; k < 3; k++) h.cen[k] = cen[k];
  h.frame_count = frame_count;
  h.map_frame_count = map_frame_count;
  h.corner_leaf = cfg.corner_filter_size;
  h.surf_leaf = cfg.surf_filter_size;
  const HTwist* tw[5] = {&sum, &incre, &tobe, &bef, &aft};
  for (int k = 0; k < 5; k++) tw[k]->get(h.transforms[k]);
  for (int t = 0; t < 2; t++) h.n[t] = tm[t].n;
  FILE* f = fopen(path, "wb");
  if (!f) throw Error(LOAMX_E_INVALID, std::string("cannot open ") + path + " for writing");
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
  for (int t = 0; t < 2 && ok; t++) {
    std::vector<float4> tmp(tm[t].n);
    if (tm[t].n) LX_HIP(hipMemcpy(tmp.data(), tm[t].pts[tm[t].cur].p, sizeof(float4) * tm[t].n, hipMemcpyDeviceToHost));
    ok = tm[t].n == 0 || fwrite(tmp.data(), sizeof(float4), tm[t].n, f) == tm[t].n;
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) throw Error(LOAMX_E_INVALID, std::string("short write to ") + path);
}

void Mapper::load_snapshot(const char* path) {
  finish_update();
  spec_valid = false;   // (a partition prepared for the old map is of no use)
  LX_REQUIRE(path && *path, "NULL path");
  FILE* f = fopen(path, "rb");
  if (!f) throw Error(LOAMX_E_INVALID, std::string("cannot open ") + path);
  SnapshotHeader h;
  std::vector<float4> pts[2];
  bool ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, "LOAMXMAP", 8) == 0 && h.version == 1;
  for (int t = 0; t < 2 && ok; t++) {
    ok = h.n[t] < (1u << 30);
    if (!ok) break;
    pts[t].resize(h.n[t]);
    ok = h.n[t] == 0 || fread(pts[t].data(), sizeof(float4), h.n[t], f) == h.n[t];
  }
  fclose(f);
  if (!ok) throw Error(LOAMX_E_INVALID, std::string(path) + " is not a loamx map snapshot (version 1) or is truncated");
  LX_REQUIRE(h.corner_leaf == cfg.corner_filter_size && h.surf_leaf == cfg.surf_filter_size,
             "the snapshot was written with other map filter sizes than this handle's");
  // every point must lie inside the cube window stored with it — checked before anything of the handle changes
  for (int t = 0; t < 2; t++)
    for (const float4& q : pts[t]) {
      const int I = cube_abs(q.x) + h.cen[0], J = cube_abs(q.y) + h.cen[1], K = cube_abs(q.z) + h.cen[2];
      LX_REQUIRE(I >= 0 && I < MW && J >= 0 && J < MH && K >= 0 && K < MD, "snapshot points fall outside the cube window stored with them");
    }
  LX_HIP(hipSetDevice(cfg.device));
  LX_HIP(hipStreamSynchronize(reg.stream()));
  for (int k = 0; k < 3; k++) cen[k] = h.cen[k];
  frame_count = (long)h.frame_count;
  map_frame_count = (long)h.map_frame_count;
  HTwist* tw[5] = {&sum, &incre, &tobe, &bef, &aft};
  for (int k = 0; k < 5; k++) tw[k]->set(h.transforms[k]);
  fresh_map = false;
  n_surround = 0;
  for (int t = 0; t < 2; t++) {
    tm[t].n = 0;
    std::fill(tm[t].cube_cnt.begin(), tm[t].cube_cnt.end(), 0u);
  }
  loamx_cloud c[2];
  for (int t = 0; t < 2; t++) c[t] = loamx_cloud{pts[t].data(), (uint32_t)pts[t].size(), 16, 12, 0};
  load_cubes(&c[0], &c[1]);   // by coordinate, against the restored window; storage order kept
  LX_REQUIRE(tm[0].n == h.n[0] && tm[1].n == h.n[1], "snapshot points fall outside the cube window stored with them");
}

int Mapper::get_surround(loamx_cloud* out) {
  finish_update();
  check_cloud(out, false);
  LX_HIP(hipSetDevice(cfg.device));
  std::vector<float4> tmp(n_surround);
  if (n_surround) LX_HIP(hipMemcpy(tmp.data(), sur_out.p, sizeof(float4) * n_surround, hipMemcpyDeviceToHost));
  return unpack_cloud(tmp.data(), n_surround, out);
}

}  // namespace loamx

// ----------------------------------------------------------------------------------------------------------------
// C-ABI
// ----------------------------------------------------------------------------------------------------------------
using namespace loamx;

struct loamx_map {
  Mapper m;
  explicit loamx_map(const loamx_map_config& c) : m(c) {}
};

extern "C" {

loamx_map* loamx_map_create(const loamx_map_config* cfg) {
  loamx_map* h = nullptr;
  guard([&]() {
    loamx_map_config c;
    if (cfg) c = *cfg; else loamx_map_default_config(&c);
    // same validation as LaserMapping::setup (LaserMapping.cpp:56-152)
    LX_REQUIRE(c.scan_period > 0.f, "scan_period must be positive");
    LX_REQUIRE(c.max_iterations >= 1 && c.max_iterations <= 64, "max_iterations must be in [1, 64]");
    LX_REQUIRE(c.delta_t_abort > 0.f && c.delta_r_abort > 0.f, "abort thresholds must be positive");
    LX_REQUIRE(c.corner_filter_size >= 0.001f && c.surf_filter_size >= 0.001f, "filter sizes must be >= 0.001");
    h = new loamx_map(c);
    return LOAMX_OK;
  });
  return h;
}
void loamx_map_destroy(loamx_map* h) { delete h; }

int loamx_map_update_odometry(loamx_map* h, const float t[6]) {
  return guard([&]() { LX_REQUIRE(h && t, "NULL argument"); h->m.sum.set(t); return LOAMX_OK; });
}
int loamx_map_insert(loamx_map* h, const loamx_cloud* corner_last, const loamx_cloud* surf_last, const float pose6[6]) {
  return guard([&]() {
    LX_REQUIRE(h && corner_last && surf_last && pose6, "NULL argument");
    return h->m.insert(corner_last, surf_last, pose6);
  });
}
int loamx_map_process(loamx_map* h, const loamx_cloud* corner_last, const loamx_cloud* surf_last, loamx_cloud* full_res) {
  return guard([&]() {
    LX_REQUIRE(h && corner_last && surf_last, "NULL argument");
    return h->m.process(corner_last, surf_last, full_res);
  });
}
int loamx_map_process_linked(loamx_map* h, loamx_odom* od, loamx_cloud* full_res_registered) {
  return guard([&]() {
    LX_REQUIRE(h && od, "NULL argument");
    OdometryBatch& ob = od->od;
    LX_REQUIRE(ob.device() == h->m.cfg.device, "linked handles must live on one device");
    LX_REQUIRE(ob.link_valid(), "loamx_odom_process_linked has not handed a sweep on");
    float sum6[6];
    ob.stream_state(0).transform_sum.get(sum6);
    h->m.sum.set(sum6);   // updateOdometry (:607-611)
    Mapper::DeviceInput in{ob.d_last_corner(0), ob.stream_state(0).n_last_corner, ob.d_last_surf(0), ob.stream_state(0).n_last_surf,
                           ob.d_link_full(), ob.n_link_full(), ob.link_ready()};
    return h->m.process(nullptr, nullptr, full_res_registered, &in);
  });
}
int loamx_map_get_speculation(loamx_map* h, uint64_t counts[2]) {
  return guard([&]() {
    LX_REQUIRE(h && counts, "NULL argument");
    h->m.finish_update();
    counts[0] = h->m.spec_hits;
    counts[1] = h->m.spec_misses;
    return LOAMX_OK;
  });
}
int loamx_map_get_transform(loamx_map* h, int which, float t[6]) {
  return guard([&]() {
    LX_REQUIRE(h && t && which >= 0 && which < 4, "invalid argument");
    const HTwist* tw[4] = {&h->m.aft, &h->m.bef, &h->m.tobe, &h->m.sum};
    tw[which]->get(t);
    return LOAMX_OK;
  });
}
int loamx_map_set_transform(loamx_map* h, int which, const float t[6]) {
  return guard([&]() {
    LX_REQUIRE(h && t && which >= 0 && which < 4, "invalid argument");
    HTwist* tw[4] = {&h->m.aft, &h->m.bef, &h->m.tobe, &h->m.sum};
    tw[which]->set(t);
    return LOAMX_OK;
  });
}
int loamx_map_update_imu(loamx_map* h, double stamp_sec, float roll, float pitch) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->m.update_imu(stamp_sec, roll, pitch);
    return LOAMX_OK;
  });
}
int loamx_map_set_time(loamx_map* h, double laser_odometry_time_sec) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->m.laser_odometry_time = laser_odometry_time_sec;
    return LOAMX_OK;
  });
}
int loamx_map_has_fresh_map(loamx_map* h) { return (h && h->m.fresh_map) ? 1 : 0; }
int loamx_map_get_surround(loamx_map* h, loamx_cloud* out) {
  return guard([&]() { LX_REQUIRE(h && out, "NULL argument"); return h->m.get_surround(out); });
}
int loamx_map_load_cubes(loamx_map* h, const loamx_cloud* corner, const loamx_cloud* surf) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->m.load_cubes(corner, surf); return LOAMX_OK; });
}
int loamx_map_get_cubes(loamx_map* h, int which, loamx_cloud* out) {
  return guard([&]() { LX_REQUIRE(h && out, "NULL argument"); return h->m.get_cubes(which, out); });
}
int loamx_map_save_snapshot(loamx_map* h, const char* path) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->m.save_snapshot(path); return LOAMX_OK; });
}
int loamx_map_load_snapshot(loamx_map* h, const char* path) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->m.load_snapshot(path); return LOAMX_OK; });
}
int loamx_map_set_timing(loamx_map* h, int on) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); h->m.reg.set_timing(on != 0); return LOAMX_OK; });
}
int loamx_map_get_timing(loamx_map* h, float ms[4], uint64_t counts[4]) {
  return guard([&]() { LX_REQUIRE(h && ms && counts, "NULL argument"); h->m.reg.get_timing(ms, counts); return LOAMX_OK; });
}
int loamx_map_get_stats(loamx_map* h, int s[8]) {
  return guard([&]() {
    LX_REQUIRE(h && s, "NULL argument");
    const SweepStats& st = h->m.last_stats;
    s[0] = st.iterations; s[1] = st.sel; s[2] = st.corner_q; s[3] = st.surf_q;
    s[4] = (int)h->m.last_sub[0]; s[5] = (int)h->m.last_sub[1]; s[6] = st.degenerate; s[7] = h->m.last_optimized ? 1 : 0;
    return LOAMX_OK;
  });
}

}  // extern "C"
