// Bucketed voxel-grid down-sampling of the registration's stack clouds (pcl::VoxelGrid semantics, BasicLaserMapping.cpp:512-527)
// in three short launches — no phase barriers, no multi-pass global sort, no histogram:
//   k_vb_plan     one workgroup per segment (sweep x {corner, surf}): 512 evenly spaced points of the segment are ranked by their voxel
//                 (iz, iy, ix) and every (512 / buckets)-th one becomes a splitter — the segment's voxel order is cut into
//                 ceil(points / VB_T) contiguous ranges ("buckets") of about VB_T points each.
//   k_vb_stack    one thread per point: the map -> sensor round trip itself (:512-516), the exact voxel, its bucket (binary search
//                 over the segment's splitters) and a slot in the bucket's fixed-capacity array (counters aggregated per workgroup).
//   k_vb_reduce   one workgroup per bucket: voxel indices linearised over the bucket's own bounding box (few bits), LSD radix sort
//                 of (voxel index, input position) entirely in LDS, run heads, output offset from a look-back over the earlier
//                 buckets' head counts, voxel means summed in input order.
// Buckets are ranges of the voxel order, so the buckets of a segment one after the other ARE pcl::VoxelGrid's output order, whatever
// the splitters: the sample only balances the load.  A bucket over capacity (a sample that missed a dense spot, more than VB_CAP
// points in one voxel), a coordinate beyond +-2^20 voxels, a segment box with more than INT_MAX voxels (PCL's pass-through case)
// raise the run's fail word; the kernels then leave an empty result and the host re-runs the sweep(s) through the general kernel
// (k_vox_ds_seg).  Output is bit-identical to that kernel's (same voxel order, same summation order).
#pragma once
#include "common.h"
#include "dev_math.hpp"

namespace loamx {

#ifndef VB_CAP_LOG2
#define VB_CAP_LOG2 12
#endif
constexpr int VB_CAP = 1 << VB_CAP_LOG2;        // slots per bucket (4096: a 512-thread workgroup with 41 KB of LDS)
constexpr int VB_T = VB_CAP / 2;                // target points per bucket
constexpr int VB_SAMPLE = 512;                  // sample size per segment (= threads of k_vb_plan); also the most buckets a segment may have
constexpr int VB_MAXSEG = 4096;

struct VbSeg {        // plan of one segment
  uint32_t bucket0;   // first bucket (index into the run's bucket arrays)
  uint32_t nbuckets;
  uint32_t pos_bits;  // bits of the largest input position inside the segment
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
  DevBuf<unsigned long long> lo_;          // per bucket: the smallest voxel key it takes (the first bucket of a segment: 0)
  DevBuf<uint32_t> bseg_;                  // per bucket: its segment
  DevBuf<int> box_;                        // [nseg][6] voxel bounds of the stack points (k_vb_reduce; PCL's pass-through test)
  DevBuf<uint32_t> cnt_, heads_, ctl_;     // per bucket: points, run heads + 1 once published; ctl: [0] fail epoch
  DevBuf<uint32_t> elems_;                 // [buckets][VB_CAP] input positions inside the segment, in arrival order
  PinBuf<uint32_t> h_fail_;                // [0] fail epoch (host-visible copy), [1] timeout, [2..] epoch of the last run that met reason r
  uint32_t epoch_ = 0, nb_ = 0;
  bool ctl_ready_ = false;
};

}  // namespace loamx
