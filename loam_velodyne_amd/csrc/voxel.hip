// Segmented voxel-grid down-sampling kernels (see voxel.cuh).
#include "voxel.cuh"
#include <rocprim/rocprim.hpp>

namespace loamx {

__global__ void k_vox_init_minmax(int* mm, uint32_t nseg) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * nseg) mm[i] = (i % 6) < 3 ? 2147483647 : (-2147483647 - 1);
}

__global__ __launch_bounds__(256) void k_vox_ijk(const float4* __restrict__ pts, const uint8_t* __restrict__ valid, uint32_t n,
                                                 const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_ids,
                                                 uint32_t nseg, float inv_even, float inv_odd, int* __restrict__ ijk,
                                                 int* __restrict__ seg_minmax) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n && !(valid && !valid[i]);
  uint32_t seg = 0;
  int ix = 0, iy = 0, iz = 0;
  if (active) {
    seg = seg_ids ? seg_ids[i] : vox_find_seg(seg_off, nseg, i);
    const float inv = (seg & 1) ? inv_odd : inv_even;
    const float4 p = pts[i];
    ix = (int)floorf(p.x * inv); iy = (int)floorf(p.y * inv); iz = (int)floorf(p.z * inv);
    ijk[3 * i] = ix; ijk[3 * i + 1] = iy; ijk[3 * i + 2] = iz;
  }
  seg_minmax_update(seg_minmax, active, seg, ix, iy, iz);
}

// key = seg << 36 | dz << 24 | dy << 12 | dx  (order == PCL's ix + iy*divx + iz*divx*divy inside a segment).
// A segment whose box would overflow PCL's int32 voxel index (or the 12-bit fields) is passed through unfiltered, as
// PCL does ("leaf size is too small for the input dataset"): every point keeps its own key.  Ignored slots get the
// pseudo-segment nseg so they sort behind everything.
__global__ __launch_bounds__(256) void k_vox_keys(uint32_t n, const uint8_t* __restrict__ valid, const uint32_t* __restrict__ seg_off,
                                                  const uint32_t* __restrict__ seg_ids, uint32_t nseg, const int* __restrict__ ijk,
                                                  const int* __restrict__ seg_minmax,
                                                  unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  vals[i] = i;
  if (valid && !valid[i]) {
    keys[i] = (unsigned long long)nseg << VOX_SEG_SHIFT;
    return;
  }
  const uint32_t seg = seg_ids ? seg_ids[i] : vox_find_seg(seg_off, nseg, i);
  const int* mm = seg_minmax + 6 * seg;
  const long long dx = (long long)mm[3] - mm[0] + 1, dy = (long long)mm[4] - mm[1] + 1, dz = (long long)mm[5] - mm[2] + 1;
  unsigned long long k;
  if (dx * dy * dz > 2147483647LL || dx > 4096 || dy > 4096 || dz > 4096) {
    k = (unsigned long long)(seg_ids ? i : i - seg_off[seg]);
  } else {
    k = ((unsigned long long)(ijk[3 * i + 2] - mm[2]) << 24) | ((unsigned long long)(ijk[3 * i + 1] - mm[1]) << 12) |
        (unsigned long long)(ijk[3 * i] - mm[0]);
  }
  keys[i] = ((unsigned long long)seg << VOX_SEG_SHIFT) | k;
}

// head flags of the voxel runs; the points are gathered into sorted order on the way (coalesced stores), so that the
// reduction reads them contiguously instead of chasing vals[] -> pts[] one dependent load after the other
__global__ __launch_bounds__(256) void k_vox_heads(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                   const float4* __restrict__ pts, uint32_t n, uint32_t nseg,
                                                   uint32_t* __restrict__ head, float4* __restrict__ gathered, uint32_t* __restrict__ d_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *d_n = n;   // element count for the device-n scan that follows
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const bool ignored = (k >> VOX_SEG_SHIFT) >= nseg;
  head[i] = (!ignored && (i == 0 || k != keys[i - 1])) ? 1u : 0u;
  gathered[i] = pts[vals[i]];
}

// one thread per voxel head: float mean of x,y,z,intensity over the run, accumulated in input order (the sort is stable).
// The workgroup's 256 sorted points and keys are staged in LDS; a run that continues into the next workgroup's range is
// finished from memory.
__global__ __launch_bounds__(256) void k_vox_reduce(const unsigned long long* __restrict__ keys, const float4* __restrict__ gathered,
                                                    const uint32_t* __restrict__ head, const uint32_t* __restrict__ head_scan,
                                                    uint32_t n, float4* __restrict__ out) {
  __shared__ float4 sp[256];
  __shared__ unsigned long long sk[256];
  const uint32_t base = blockIdx.x * blockDim.x, i = base + threadIdx.x;
  if (i < n) { sp[threadIdx.x] = gathered[i]; sk[threadIdx.x] = keys[i]; }
  __syncthreads();
  if (i >= n || !head[i]) return;
  const unsigned long long k = sk[threadIdx.x];
  float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
  uint32_t j = i;
  const uint32_t lim = base + 256 < n ? base + 256 : n;
  do {
    const float4 p = sp[j - base];
    sx += p.x; sy += p.y; sz += p.z; si += p.w;
    j++;
  } while (j < lim && sk[j - base] == k);
  if (j == lim) {
    while (j < n && keys[j] == k) {
      const float4 p = gathered[j];
      sx += p.x; sy += p.y; sz += p.z; si += p.w;
      j++;
    }
  }
  const float cnt = (float)(j - i);
  out[head_scan[i]] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
}

// out_off[s] = number of voxels emitted before segment s (first sorted slot whose segment >= s); out_off[nseg] = total
__global__ void k_vox_offsets(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ head_scan, uint32_t n,
                              uint32_t nseg, uint32_t* __restrict__ out_off) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nseg) return;
  uint32_t lo = 0, hi = n;   // first position with seg(key) >= s
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if ((keys[mid] >> VOX_SEG_SHIFT) >= s) hi = mid; else lo = mid + 1;
  }
  out_off[s] = head_scan[lo];   // head_scan[n] = total
}

void VoxelPipeline::init(hipStream_t st) {
  st_ = st;
  tile_sums_.reserve(8192);
  scratch_.reserve(16);
}

void VoxelPipeline::reserve(uint32_t n, uint32_t nseg) {
  LX_REQUIRE(n < SCAN_MAX_N, "too many points for one voxel pass");
  LX_REQUIRE(nseg < (1u << 20), "too many voxel segments");
  ijk_.reserve((size_t)3 * n + 3);
  seg_minmax_.reserve((size_t)6 * nseg + 6);
  keys_.reserve(n + 1);
  keys_sorted_.reserve(n + 1);
  vals_.reserve(n + 1);
  vals_sorted_.reserve(n + 1);
  head_.reserve(n + 2);
  head_scan_.reserve(n + 2);
  gathered_.reserve(n + 1);
  size_t need = 0;
  LX_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_.p, keys_sorted_.p, vals_.p, vals_sorted_.p, (size_t)(n ? n : 1), 0, 48, st_));
  if (need > sort_tmp_bytes_) {
    sort_tmp_.reserve(need);
    sort_tmp_bytes_ = need;
  }
}

void VoxelPipeline::reset_minmax(uint32_t nseg) {
  hipLaunchKernelGGL(k_vox_init_minmax, dim3((6 * nseg + 255) / 256), dim3(256), 0, st_, seg_minmax_.p, nseg);
}

void VoxelPipeline::compute_ijk(const float4* pts, const uint8_t* valid, uint32_t n, const uint32_t* d_seg_off, uint32_t nseg,
                                float inv_even, float inv_odd, const uint32_t* d_seg_ids) {
  reset_minmax(nseg);
  if (n == 0) return;
  hipLaunchKernelGGL(k_vox_ijk, dim3((n + 255) / 256), dim3(256), 0, st_, pts, valid, n, d_seg_off, d_seg_ids, nseg, inv_even, inv_odd, ijk_.p,
                     seg_minmax_.p);
}

void VoxelPipeline::sort_reduce(const float4* pts, const uint8_t* valid, uint32_t n, const uint32_t* d_seg_off, uint32_t nseg,
                                float4* out, uint32_t* d_out_off, const uint32_t* d_seg_ids) {
  if (n == 0) {
    LX_HIP(hipMemsetAsync(d_out_off, 0, sizeof(uint32_t) * (nseg + 1), st_));
    return;
  }
  const uint32_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_vox_keys, dim3(nb), dim3(256), 0, st_, n, valid, d_seg_off, d_seg_ids, nseg, ijk_.p, seg_minmax_.p, keys_.p, vals_.p);
  size_t tmp = sort_tmp_bytes_;
  int seg_bits = 1;
  while ((1u << seg_bits) <= nseg) seg_bits++;
  LX_HIP(rocprim::radix_sort_pairs(sort_tmp_.p, tmp, keys_.p, keys_sorted_.p, vals_.p, vals_sorted_.p, (size_t)n, 0,
                                   VOX_SEG_SHIFT + seg_bits, st_));
  hipLaunchKernelGGL(k_vox_heads, dim3(nb), dim3(256), 0, st_, keys_sorted_.p, vals_sorted_.p, pts, n, nseg, head_.p, gathered_.p,
                     scratch_.p);
  exclusive_scan_u32(head_.p, head_scan_.p, tile_sums_.p, scratch_.p, scratch_.p + 1, n, st_);
  hipLaunchKernelGGL(k_vox_reduce, dim3(nb), dim3(256), 0, st_, keys_sorted_.p, gathered_.p, head_.p, head_scan_.p, n, out);
  hipLaunchKernelGGL(k_vox_offsets, dim3((nseg + 64) / 64), dim3(64), 0, st_, keys_sorted_.p, head_scan_.p, n, nseg, d_out_off);
  LX_HIP(hipGetLastError());
}

}  // namespace loamx
