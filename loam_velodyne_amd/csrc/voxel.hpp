// Segmented voxel-grid down-sampling (pcl::VoxelGrid semantics) on device: keys -> stable radix sort -> run heads ->
// their scan -> per-voxel float means in input order, all inside one persistent kernel (k_vox_ds, voxel.hip).  Used for the mapping stack clouds
// (reference BasicLaserMapping.cpp:519-527), the per-ring less-flat clouds (BasicScanRegistration.cpp:246-252) and the
// per-cube map re-filtering (BasicLaserMapping.cpp:580-593).
//
// Voxel membership = floor(p * (1/leaf)) per axis (PCL multiplies by the reciprocal leaf); output order inside a
// segment = ascending (iz, iy, ix) = PCL's ascending voxel index; the mean covers x, y, z and intensity.  PCL's
// unstable std::sort leaves the summation order inside a voxel unspecified; here it is input order (stable radix sort).
#pragma once
#include <algorithm>
#include "common.h"
#include "scan.hpp"

namespace loamx {


__device__ inline uint32_t vox_find_seg(const uint32_t* __restrict__ off, uint32_t nseg, uint32_t i) {
  uint32_t lo = 0, hi = nseg;   // off[lo] <= i < off[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// per-segment integer bounds.  Must be called by ALL threads of the block (no early returns before it); `active`
// marks the threads that carry a point.  Three levels keep the global atomics rare: lanes sharing the segment of the
// wave's first active lane are reduced with shuffles; waves whose segment is the block's lead segment (the segment of
// the block's first point — almost always, segments are thousands of points long) combine in LDS; one thread per bound
// then issues the block's global atomics.  Stragglers (a wave or lane straddling a segment border) use their own.
__device__ inline void seg_minmax_update(int* __restrict__ seg_minmax, bool active, uint32_t seg, int ix, int iy, int iz) {
  __shared__ int blk_mm[6];
  __shared__ uint32_t blk_seg;
  __shared__ int blk_used;
  if (threadIdx.x == 0) { blk_seg = seg; blk_used = 0; }
  if (threadIdx.x < 6) blk_mm[threadIdx.x] = threadIdx.x < 3 ? 2147483647 : (-2147483647 - 1);
  __syncthreads();
  const uint32_t lead = blk_seg;   // (thread 0 may be inactive: then nobody matches a stale value only by accident, which is harmless)
  const unsigned long long mact = __ballot(active);
  if (mact != 0) {
    const int leader = __builtin_ctzll(mact);
    const uint32_t first = __shfl(seg, leader, 64);
    const bool same = active && seg == first;
    int v[6] = {same ? ix : 2147483647, same ? iy : 2147483647, same ? iz : 2147483647,
                same ? ix : (-2147483647 - 1), same ? iy : (-2147483647 - 1), same ? iz : (-2147483647 - 1)};
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int a_ = __shfl_xor(v[k], d, 64), b_ = __shfl_xor(v[3 + k], d, 64);
        v[k] = a_ < v[k] ? a_ : v[k];
        v[3 + k] = b_ > v[3 + k] ? b_ : v[3 + k];
      }
    }
    if ((int)__lane_id() == leader) {
      if (first == lead) {
        // LDS combine for the lead segment
        atomicMin(&blk_mm[0], v[0]); atomicMin(&blk_mm[1], v[1]); atomicMin(&blk_mm[2], v[2]);
        atomicMax(&blk_mm[3], v[3]); atomicMax(&blk_mm[4], v[4]); atomicMax(&blk_mm[5], v[5]);
        blk_used = 1;
      } else {
        int* mm = seg_minmax + 6 * first;
        atomicMin(&mm[0], v[0]); atomicMin(&mm[1], v[1]); atomicMin(&mm[2], v[2]);
        atomicMax(&mm[3], v[3]); atomicMax(&mm[4], v[4]); atomicMax(&mm[5], v[5]);
      }
    }
    if (active && !same) {
      int* mm = seg_minmax + 6 * seg;
      atomicMin(&mm[0], ix); atomicMin(&mm[1], iy); atomicMin(&mm[2], iz);
      atomicMax(&mm[3], ix); atomicMax(&mm[4], iy); atomicMax(&mm[5], iz);
    }
  }
  __syncthreads();
  if (threadIdx.x < 6 && blk_used) {
    int* mm = seg_minmax + 6 * lead;
    if (threadIdx.x < 3) atomicMin(&mm[threadIdx.x], blk_mm[threadIdx.x]);
    else atomicMax(&mm[threadIdx.x], blk_mm[threadIdx.x]);
  }
}

class VoxelPipeline {
 public:
  void init(hipStream_t st);
  void reserve(uint32_t n_slots, uint32_t nseg);
  int* ijk() { return ijk_.p; }
  int* seg_minmax() { return seg_minmax_.p; }
  void reset_minmax(uint32_t nseg);
  // optional helper: ijk + per-segment bounds from points (valid may be NULL = all valid);
  // segment s uses inv_even when s is even, inv_odd otherwise
  // d_seg_ids (optional): explicit segment id per slot instead of the contiguous ranges of d_seg_off
  void compute_ijk(const float4* pts, const uint8_t* valid, uint32_t n, const uint32_t* d_seg_off, uint32_t nseg, float inv_even,
                   float inv_odd, const uint32_t* d_seg_ids = nullptr);
  // after ijk()/seg_minmax() are filled for the n slots: sort + reduce.  out gets the voxel means, d_out_off[nseg+1]
  // the per-segment output offsets.  Slots with valid[i]==0 are ignored.
  void sort_reduce(const float4* pts, const uint8_t* valid, uint32_t n, const uint32_t* d_seg_off, uint32_t nseg, float4* out,
                   uint32_t* d_out_off, const uint32_t* d_seg_ids = nullptr);
  // after the stream has been synchronised: throws if a wait inside the last launches timed out
  void check();

 private:
  hipStream_t st_ = nullptr;
  DevBuf<int> ijk_, seg_minmax_;
  DevBuf<unsigned long long> keys_[2];
  DevBuf<uint32_t> vals_[2], zero_;   // zero_ (cleared before every launch): TileSync counters | digit histograms | tile status | tile head counts
  size_t zero_words_ = 0, status_words_ = 0;
  PinBuf<uint32_t> h_err_;
  uint32_t slots_ = 0, slots_seg_ = 0;
  bool zero_ready_ = false;   // compute_ijk has cleared zero_ for the general kernel of the same (n, nseg)
  uint32_t zero_ready_n_ = 0, zero_ready_nseg_ = 0;
};

}  // namespace loamx
