// Bucketed voxel-grid down-sampling of the registration's stack clouds (pcl::VoxelGrid semantics, BasicLaserMapping.cpp:512-527)
// in a handful of short launches — no phase barriers, no multi-pass global sort:
//   plan          (k_vb_bounds, k_vb_hist, k_vb_scan: three small launches over the UNtransformed feature points of every segment
//                 = sweep x {corner, surf}) a box padded by one voxel, a 32 Ki-bin histogram of PCL's linear voxel index over that
//                 box, and from its prefix sums a partition of the index range into buckets of ~VB_T points (contiguous index
//                 ranges, so the buckets of a segment in order are its voxels in order).  The map -> sensor round trip that follows
//                 moves a point by a few 1e-5 m, i.e. almost never across a voxel face, so the partition predicted here fits the
//                 transformed points up to a handful of strays.
//   k_vb_stack    one thread per point: the round trip itself (:512-516), the exact voxel, its bucket (one table look-up) and a
//                 slot in the bucket's fixed-capacity array (one atomic per run of equal buckets in a wave).
//   k_vb_reduce   one workgroup per bucket: LSD radix sort of (voxel index, input position) entirely in LDS, run heads, output
//                 offset from a look-back over the earlier buckets' head counts, voxel means summed in input order.
// Exactness does not rest on the prediction: a point outside the padded box, a bucket over capacity, a bin too large for a
// bucket, a box with more than INT_MAX voxels (PCL's pass-through case) each raise the run's fail word; the three kernels then
// leave an empty result and the host re-runs the sweep(s) through the general kernel (k_vox_ds_seg).  Output is bit-identical
// to that kernel's (same voxel order, same summation order).
#pragma once
#include "common.h"
#include "dev_math.cuh"

namespace loamx {

#ifndef VB_CAP_LOG2
#define VB_CAP_LOG2 12
#endif
constexpr int VB_CAP = 1 << VB_CAP_LOG2;        // slots per bucket (4096: a 512-thread workgroup with 41 KB of LDS)
constexpr int VB_T = VB_CAP / 2;                // target points per bucket
constexpr int VB_MAXBIN = VB_CAP - VB_T - 128;  // largest bin a bucket can take: T + MAXBIN + strays <= CAP
constexpr int VB_BIN_BITS = 15;
constexpr int VB_BINS = 1 << VB_BIN_BITS;       // histogram bins per segment
constexpr int VB_MAXBUCK = 1024;                // buckets per segment (larger segments take the general kernel)
constexpr int VB_MAXSEG = 4096;

struct VbSeg {        // plan of one segment
  int mn[3];          // padded box, voxel units
  uint32_t dim[3];
  uint32_t shift;     // bin = linear index >> shift
  uint32_t bucket0;   // first bucket (index into the run's bucket arrays)
  uint32_t nbuckets;
  uint32_t pos_bits;  // bits of the largest input position inside the segment
};
struct VbBucket {
  uint32_t seg;
  uint32_t key_lo;    // smallest linear index the bucket can hold
  uint32_t key_bits;  // bits of (largest - smallest) index of the bucket's non-empty bins
  uint32_t pad;
};

class VoxBucket {
 public:
  void init(hipStream_t st) { st_ = st; }
  // segments = contiguous ranges [h_seg_off[k], h_seg_off[k+1]) of the n input points (read from in, or from src[k] when given);
  // segment k belongs to sweep k / 2 (poses) and uses inv_even / inv_odd = 1 / leaf by parity.  stack receives the round-trip
  // points, out / d_out_off the voxel means and the nseg + 1 output offsets.  Asynchronous on the stream.
  void run(const float4* in, const float4* const* d_src, uint32_t n, const uint32_t* d_seg_off, const uint32_t* h_seg_off, uint32_t nseg,
           const Pose* d_poses, float inv_even, float inv_odd, float4* stack, float4* out, uint32_t* d_out_off);
  static bool fits(uint32_t n, uint32_t nseg) { return nseg >= 1 && nseg <= (uint32_t)VB_MAXSEG && n >= 1 && n < (1u << 24); }
  // after the stream has been synchronised: did the last run() give up (the caller must redo it with the general kernel)?
  bool failed() const { return h_fail_.p && *(volatile uint32_t*)h_fail_.p == epoch_; }
  // device word / value a later kernel of the same stream can test to skip work that would use the empty result
  const uint32_t* d_fail_word() const { return ctl_.p; }
  uint32_t epoch() const { return epoch_; }
  uint32_t why() const;   // after failed(): bit mask of the give-up reasons (voxbucket.hip, vb_fail)
  void check();   // throws when a look-back wait timed out
  uint32_t last_buckets() const { return nb_; }

 private:
  hipStream_t st_ = nullptr;
  DevBuf<VbSeg> segs_;
  DevBuf<VbBucket> buckets_;
  DevBuf<uint16_t> bin2bucket_;            // [nseg][VB_BINS]
  DevBuf<int> mm_;                         // [nseg][6] voxel bounds of the untransformed points
  DevBuf<uint32_t> hist_;                  // [nseg][VB_BINS]
  DevBuf<uint32_t> cnt_, heads_, ctl_;     // per bucket: points, run heads + 1 once published; ctl: [0] fail epoch, [1] claim counter
  DevBuf<unsigned long long> elems_;       // [buckets][VB_CAP]
  PinBuf<uint32_t> h_fail_;                // [0] fail epoch (host-visible copy), [1] timeout, [2..7] epoch of the last run that met reason r
  uint32_t epoch_ = 0, claim_base_ = 0, nb_ = 0;
  bool ctl_ready_ = false;
};

}  // namespace loamx
