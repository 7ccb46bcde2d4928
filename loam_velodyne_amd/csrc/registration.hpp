// Scan-to-map registration engine shared by the batched mode (loamx_batch_*) and the sequential
// BasicLaserMapping replacement (loamx_map_*).  Host classes; kernels live in registration.hip.
#pragma once
#include <functional>
#include "common.h"
#include "dev_math.hpp"
#include "voxel.hpp"
#include "voxbucket.hpp"

namespace loamx {

// uniform grid over a sub-map: cell edge >= 1.05 m so the 3x3x3 neighbourhood of a query's cell contains every
// point within the 1 m gate of BasicLaserMapping.cpp:671/:760.
struct GridDesc {
  float ox, oy, oz, inv_h;
  int nx, ny, nz;
  uint32_t ncell;
};

constexpr uint32_t LX_MAX_CELLS = 16u * 1024 * 1024 - 2048;   // scan limit (scan.hpp)
constexpr int LX_RES_THREADS = 256;
constexpr int LX_NSUM = 28;   // 21 upper-triangular AtA + 6 AtB + row count

class SubMapIndex {
 public:
  void init(hipStream_t st);
  // (re)build over n device points (packed float4; .w ignored).  Asynchronous on the stream.
  // bounds_done: the kernel that produced d_pts has folded every point into d_bounds() as it wrote it (enc_f32 atomicMin / atomicMax on
  // words 0-2 / 3-5; the accumulators are left reset by every build) — the bounding-box launch is skipped
  void build(const float4* d_pts, uint32_t n, bool bounds_done = false);
  // room for sub-maps of up to n points without another allocation (a live map grows: Mapper::ensure)
  void reserve_points(uint32_t n) { sorted_.reserve(n); cell_of_.reserve(n); rank_of_.reserve(n); }
  uint32_t* d_bounds() const { return scratch_.p; }
  // exchange contents with another index (buffers, sizes and the streams they are bound to stay with the contents' owner)
  void swap(SubMapIndex& o);
  void bind(hipStream_t st) { st_ = st; }
  uint32_t size() const { return n_; }
  const float4* sorted() const { return sorted_.p; }          // .w = original index (bit pattern)
  const uint32_t* cell_start() const { return cell_start_.p; }
  const GridDesc* desc() const { return d_desc_.p; }

 private:
  hipStream_t st_ = nullptr;
  uint32_t n_ = 0;
  DevBuf<float4> sorted_;
  DevBuf<uint32_t> cell_of_, rank_of_, cell_start_, cursor_, tile_sums_, scratch_;   // scratch_: bbox enc[6], ncell+1, total; cursor_: the cell counters (empty between builds)
  DevBuf<GridDesc> d_desc_;
};

// The same index over K clouds at once (one set of launches): clouds are concatenated, cloud c = [off[c], off[c+1]).
// All cell tables live in one array (cloud c's table starts at desc[c].cell_base) and one global scan yields start
// offsets that index the concatenated sorted array directly; .w of a sorted point = its index inside its own cloud.
struct GridDescB {
  GridDesc g;
  uint32_t cell_base;   // first entry of this cloud's cell table
  uint32_t pt_base;     // off[c]
};
// order-preserving float <-> uint32 (bounds accumulated with integer atomicMin / atomicMax)
__device__ inline uint32_t enc_f32(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float dec_f32(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}
// Bounds accumulators: six words per cloud, each in a cache line of its own (BB_STRIDE words apart) — atomics on one line serialise
// (~12 ns each, measured), and a producer kernel sends a few hundred per word.
constexpr uint32_t BB_STRIDE = 32;
__device__ __host__ inline uint32_t bb_word(uint32_t c, uint32_t a) { return (6u * c + a) * BB_STRIDE; }
// A producer kernel folds its output points into the bounds of cloud c (SubMapIndexBatch::d_bounds) as it writes them: reduced per wave
// (shuffles) and per workgroup (LDS) when their points belong to one cloud — all but the few waves that straddle a boundary — so a
// cloud's words see one atomic per workgroup.  EVERY thread of the workgroup must call (inactive ones with active = false);
// workgroups of at most 1024 threads.
__device__ inline void cloud_bounds_update(uint32_t* __restrict__ enc, bool active, uint32_t c, float x, float y, float z) {
  constexpr uint32_t NONE = 0xffffffffu, MIXED = 0xfffffffeu;
  __shared__ float s_red[16][6];
  __shared__ uint32_t s_cloud[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (int)((blockDim.x + 63) >> 6);
  const unsigned long long m = __ballot(active);
  uint32_t wc = NONE;
  if (m) {
    const uint32_t c0 = (uint32_t)__shfl((int)c, __ffsll((long long)m) - 1, 64);
    wc = __ballot(active && c != c0) == 0ull ? c0 : MIXED;
  }
  if (wc < MIXED) {
    float mn[3] = {active ? x : FLT_MAX, active ? y : FLT_MAX, active ? z : FLT_MAX};
    float mx[3] = {active ? x : -FLT_MAX, active ? y : -FLT_MAX, active ? z : -FLT_MAX};
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
        mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; a++) { s_red[wid][a] = mn[a]; s_red[wid][3 + a] = mx[a]; }
    }
  } else if (wc == MIXED && active) {   // a wave across a cloud boundary: every lane for itself
    atomicMin(&enc[bb_word(c, 0)], enc_f32(x)); atomicMin(&enc[bb_word(c, 1)], enc_f32(y)); atomicMin(&enc[bb_word(c, 2)], enc_f32(z));
    atomicMax(&enc[bb_word(c, 3)], enc_f32(x)); atomicMax(&enc[bb_word(c, 4)], enc_f32(y)); atomicMax(&enc[bb_word(c, 5)], enc_f32(z));
  }
  if (lane == 0) s_cloud[wid] = wc;
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = (int)threadIdx.x;
    for (int w = 0; w < nw; w++) {
      const uint32_t cw = s_cloud[w];
      if (cw >= MIXED) continue;
      bool first = true;
      for (int v = 0; v < w; v++) first = first && s_cloud[v] != cw;
      if (!first) continue;   // (combined with an earlier wave of the same cloud)
      float r = s_red[w][a];
      for (int u = w + 1; u < nw; u++)
        if (s_cloud[u] == cw) r = a < 3 ? fminf(r, s_red[u][a]) : fmaxf(r, s_red[u][a]);
      if (a < 3) atomicMin(&enc[bb_word(cw, a)], enc_f32(r)); else atomicMax(&enc[bb_word(cw, a)], enc_f32(r));
    }
  }
}
__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ inline void cell_coords(const GridDesc& g, float x, float y, float z, int& cx, int& cy, int& cz) {
  cx = clampi((int)floorf((x - g.ox) * g.inv_h), 0, g.nx - 1);
  cy = clampi((int)floorf((y - g.oy) * g.inv_h), 0, g.ny - 1);
  cz = clampi((int)floorf((z - g.oz) * g.inv_h), 0, g.nz - 1);
}

// Points arrive in scan order, so neighbouring lanes mostly fall into the same cell: one atomic per RUN of equal cells
// in a wave instead of one per point (64 lanes hammering two or three counters serialise in L2).
// run_head_len: for the calling lane, the lane that starts its run and, if it is that lane, the run's length.
__device__ inline void wave_runs(uint32_t key, bool active, int& head_lane, int& run_len) {
  const int lane = (int)__lane_id();
  const uint32_t prev = __shfl_up(key, 1, 64);
  const unsigned long long act = __ballot(active);
  const unsigned long long heads = __ballot(active && (lane == 0 || prev != key || !((act >> (lane > 0 ? lane - 1 : 0)) & 1ull)));
  const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  const unsigned long long below = heads & upto;
  head_lane = below ? 63 - __builtin_clzll(below) : lane;
  const unsigned long long stops = (heads | ~act) & ~upto;   // next head, or the first inactive lane
  const int end = stops ? __builtin_ctzll(stops) : 64;
  run_len = end - lane;   // meaningful on the head lane
}

class SubMapIndexBatch {
 public:
  void init(hipStream_t st);
  // d_pts: concatenated points; h_off: K+1 host offsets.  Asynchronous on the stream.
  // d_off_ready: the K+1 offsets already on the device (skips the upload); prepared: prepare(K) was enqueued earlier
  // bounds_done: the kernel that produced d_pts has folded them into d_bounds() already (cloud_bounds_update; prepare(K) came first)
  void build(const float4* d_pts, const uint32_t* h_off, uint32_t K, const uint32_t* d_off_ready = nullptr, bool prepared = false,
             bool bounds_done = false);
  void prepare(uint32_t K);
  uint32_t* d_bounds() const { return enc_.p; }
  float cell_size = 1.05f;   // initial cell edge (grown by 1.25x while the cell table would exceed its budget)
  bool pack_ring = false;    // .w of a sorted point = (ring << 24) | index inside its cloud, ring = (int) of the input's .w (255: does not fit)
  const float4* sorted() const { return sorted_.p; }
  const uint32_t* cell_start(uint32_t c) const { return cell_start_.p; }   // tables are addressed through desc(c)->cell_base
  const GridDescB* desc(uint32_t c) const { return d_desc_.p + c; }
  const uint32_t* cell_table() const { return cell_start_.p; }

 private:
  hipStream_t st_ = nullptr;
  DevBuf<float4> sorted_;
  void reset_bounds_(uint32_t K);
  uint32_t enc_ready_ = 0;   // clouds whose bounds accumulators are known to be reset
  DevBuf<uint32_t> cell_of_, rank_of_, cell_start_, cursor_, tile_sums_, scratch_, d_off_, enc_;   // cursor_: the cell counters (empty between builds)
  DevBuf<GridDescB> d_desc_;
  PinBuf<uint32_t> h_off_pin_;
};

struct SweepStats {
  int iterations, sel, corner_q, surf_q, degenerate, done, pad0, pad1;
};

// check word of the pinned (pose, statistics) mirror of one sweep: the host polls `done` while the kernel may still be
// storing, and takes the pair only when the word it finds matches the words it read (SweepStats::pad1 of the mirror)
__host__ __device__ inline int mirror_check_word(const SweepStats& st, const Pose& T) {
  const unsigned* w = reinterpret_cast<const unsigned*>(&T);
  unsigned x = 0x5bd1e995u ^ (unsigned)st.iterations ^ ((unsigned)st.sel << 8) ^ ((unsigned)st.degenerate << 30);
  for (int i = 0; i < (int)(sizeof(Pose) / 4); i++) x = (x << 5 | x >> 27) ^ w[i];
  return (int)x;
}

struct RegParams {
  int max_iterations = 10;
  float delta_t_abort = 0.05f, delta_r_abort = 0.05f;
  float corner_leaf = 0.2f, surf_leaf = 0.4f;
};

// Registers up to max_sweeps sweeps' (corner_last, surf_last) against the two indexed sub-maps.
class Registrar {
 public:
  Registrar(int device, uint32_t max_sweeps);
  ~Registrar();
  RegParams params;
  SubMapIndex corner_index, surf_index;
  // Double-buffered sub-map (BASELINE configs[4]): the NEXT epoch's map is indexed on a stream of its own
  // while sweeps are still registered against the current one; swap_submap() makes it current at a batch boundary.
  // The caller's buffers are only read until the staged build has finished (the index keeps a cell-sorted copy).
  void stage_submap_device(const float4* d_corner, uint32_t nc, const float4* d_surf, uint32_t ns, hipEvent_t wait_for = nullptr);
  void stage_submap_host(const loamx_cloud* corner, const loamx_cloud* surf);
  bool submap_staged() const { return next_staged_; }
  void swap_submap();
  bool double_buffer_full = false;   // stage_full() alternates between two buffers (the pipeline's asynchronous downloads)
  bool defer_full = false;    // run_async() leaves the full-resolution clouds unregistered; finish_with_poses() does it
  void finish_with_poses(const float* poses6);
  std::function<void()> on_first_wait;   // early_exit: called once, right before run_async() first blocks on the flags
  bool early_exit = false;    // run_async() may block on the done flags to skip the launches after convergence
  hipStream_t stream() const { return st_; }

  // frozen sub-map (host records or device float4)
  void set_submap_host(const loamx_cloud* corner, const loamx_cloud* surf);
  void set_submap_device(const float4* d_corner, uint32_t nc, const float4* d_surf, uint32_t ns, bool sync = true);

  // stage inputs (H2D, async on the stream)
  void upload(uint32_t n_sweeps, const loamx_cloud* corner_last, const loamx_cloud* surf_last, const loamx_cloud* full_res,
              const float* guess6, bool wait = true);
  // same with device-resident packed float4 inputs (copied device-to-device; async on the stream)
  void upload_device(uint32_t n_sweeps, const float4* const* corner_last, const uint32_t* n_corner, const float4* const* surf_last,
                     const uint32_t* n_surf, const float4* const* full_res, const uint32_t* n_full, const float* guess6);
  // reserve the full-resolution staging area for the next upload_device() and return it: the caller fills
  // [full_offset(s), full_offset(s+1)) itself (e.g. with a fused re-projection kernel) and passes full_res = NULL
  float4* stage_full(uint32_t n_sweeps, const uint32_t* n_full);
  float4* stage_full_next(uint32_t n_sweeps, const uint32_t* n_full);   // NULL: not possible now
  bool adopt_full_next(uint32_t n_sweeps, const uint32_t* n_full);      // false: nothing (matching) was pre-staged
  // device-only: stack round trip + voxel DS + LM iterations (+ full-res registration)
  void run_async();
  void sync();
  void download(float* poses6, int* stats4);
  void download_stats(SweepStats* out);
  int download_full_res(uint32_t sweep, loamx_cloud* out);
  void download_full_res_async(uint32_t sweep, const loamx_cloud* into = nullptr);   // into: the cloud download_full_res() will be given (a pinned one takes the copy directly)
  void set_submap_device_split(const float4* d_corner, uint32_t nc, hipStream_t corner_stream, const float4* d_surf, uint32_t ns, bool bounds_done = false);
  // down-sampled query clouds of a sweep (device pointers valid until the next run); counts need a sync'd download
  void download_ds(uint32_t sweep, std::vector<float4>& corner_ds, std::vector<float4>& surf_ds);
  // registered (final-pose) DS clouds, for map insertion: device array + offsets
  const float4* d_ds_points() const { return ds_pts_.p; }
  const uint32_t* d_ds_offsets() const { return ds_off_.p; }
  const Pose* d_poses() const { return poses_.p; }
  const float4* d_full_res() const { return full_.p; }
  uint32_t full_offset(uint32_t s) const { return h_full_off_[s]; }

  // parity hook (loamx_batch_knn_probe): exact 5-NN of n map-frame points in the corner (0) / surf (1) sub-map index
  void knn_probe(int which, const float* xyz, uint32_t n, uint32_t* idx5, float* d2_5);
  void qr6_probe(const float* ata, const float* atb, uint32_t n, float* x_coop, float* x_scalar);
  void xrec_stress(uint32_t pairs, uint32_t rounds, unsigned long long out4[4]);
  void set_timing(bool on, bool per_launch = true) { timing_ = on; launch_timing_ = per_launch; }   // per_launch: event pairs around every Gauss-Newton launch
  void get_timing(float ms[4], uint64_t counts[4]);
  uint32_t n_sweeps() const { return n_sweeps_; }
  bool submap_sufficient() const { return corner_index.size() > 10 && surf_index.size() > 100; }

 private:
  int device_;
  uint32_t max_sweeps_, n_sweeps_ = 0;
  hipStream_t st_ = nullptr;
  bool timing_ = false, timed_run_ = false, launch_timing_ = true;

  // owned copies of a host-provided sub-map
  DevBuf<float4> own_corner_, own_surf_;

  // inputs: segments 2s (corner), 2s+1 (surf)
  PinBuf<float4> h_in_, h_full_;
  std::vector<uint32_t> h_seg_off_, h_full_off_;
  PinBuf<float> h_guess_;
  DevBuf<float4> in_, stack_, ds_pts_, full_, full_alt_;
  DevBuf<uint32_t> seg_off_, full_off_, ds_off_;
  DevBuf<float> guess_;
  DevBuf<const float4*> src_ptrs_;
  bool full_staged_ = false, full_next_staged_ = false, results_final_ = false;
  std::vector<uint32_t> next_full_off_;
  hipEvent_t ev_look_ = nullptr;
  PinBuf<float4> h_full_dl_;      // download_full_res(): pinned landing area of one sweep's registered cloud
  void* full_dl_direct_ = nullptr;   // download_full_res_async() copied straight into this caller memory
  int full_dl_sweep_ = -1;       // the sweep whose copy download_full_res_async() has enqueued since the clouds were registered (-1: none)
  VoxelPipeline vox_;
  // the stack clouds' voxel grid normally takes the bucketed path (voxbucket.hpp); a run that gives up is repeated through the
  // general kernel as soon as the host has synchronised with it (LOAMX_VOX_LEGACY=1 forces the general kernel)
  VoxBucket vb_;
  bool vb_disabled_ = false;   // LOAMX_VOX_LEGACY
  bool jacobi_eig_ = false;    // LOAMX_EIG_JACOBI: the edge fit's eigen-decomposition by the oracle's Jacobi iteration instead of the closed form
  bool vb_unchecked_ = false;  // the last run used the bucketed path and its fail word has not been looked at yet
  void enqueue_front(bool legacy);   // k_pose_init + stack round trip + voxel grid
  void enqueue_full(int mode);       // transformFullResToMap
  bool full_enqueued_ = false;
  void redo_if_bucket_path_failed(); // after a stream sync
  void note_bucket_give_up();
  uint64_t vb_give_ups_ = 0;         // runs repeated through the general kernel so far
  uint32_t n_in_ = 0, n_full_ = 0, max_q_per_sweep_ = 0;
  bool mirrors_written_ = false;      // h_poses_ / h_stats_ hold (after a stream sync) this run's final values
  bool run_iterations(bool trace, double& th2, double& th3);   // true: the bucketed voxel path gave up, run again
  bool wait_for_mirrors();
  void fetch_results();
  int pred_iters_ = 4;        // early_exit: iterations to enqueue before the first look at the done flags

  DevBuf<Pose> poses_;
  DevBuf<SweepStats> stats_;
  DevBuf<float> matP_;        // 36 per sweep
  DevBuf<double> partials_;   // per sweep x blocks x LX_NSUM
  // views of the per-run parameters (separate buffers after upload(), one block after upload_device())
  const float* d_guess_ = nullptr;
  const uint32_t* d_seg_off_ = nullptr;
  const uint32_t* d_full_off_ = nullptr;
  const float4* const* d_src_ = nullptr;
  DevBuf<char> blob_;
  PinBuf<char> h_blob_;
  SubMapIndex corner_next_, surf_next_;
  DevBuf<float4> next_corner_, next_surf_;
  PinBuf<float4> h_next_corner_, h_next_surf_;
  hipStream_t st_build_ = nullptr;
  hipEvent_t ev_build_ = nullptr, ev_swap_ = nullptr;
  bool next_staged_ = false, swapped_once_ = false;
  DevBuf<uint32_t> arrive_;   // per sweep: k_gn_iter workgroups that have delivered their tile sums
  DevBuf<uint32_t> full_done_;   // per sweep: tag of the k_transform_full launch that registered its full-resolution cloud
  uint32_t full_tag_ = 0;
  uint32_t nblk_ = 0;

  std::vector<hipEvent_t> ev_;   // timing events: [0]=run start, [1]=run end, then pairs per Gauss-Newton launch
  int n_res_launch_ = 0;
  PinBuf<SweepStats> h_stats_;
  PinBuf<Pose> h_poses_;
};

}  // namespace loamx
