// Device-wide exclusive scan of uint32 with the element count read from device memory (no host sync).
// Three launches: tile scan (2048 elements / 256-thread block), scan of tile sums (one 1024-thread block, up to
// 8192 tiles = 16 Mi elements), uniform add.  wave64 shuffles for the intra-wave step.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace loamx {

constexpr int SCAN_TILE = 2048;
constexpr uint32_t SCAN_MAX_N = 8192u * SCAN_TILE;

__device__ inline uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// block-wide exclusive scan of one value per thread (blockDim.x multiple of 64, <= 1024); returns exclusive prefix,
// total in `total` (valid for all threads)
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t* lds /* >= 17 */, uint32_t& total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t inc = wave_incl_scan(v, lane);
  if (lane == 63) lds[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    uint32_t s = lane < nw ? lds[lane] : 0u;
    uint32_t si = wave_incl_scan(s, lane);
    if (lane < nw) lds[lane] = si - s;
    if (lane == nw - 1) lds[16] = si;
  }
  __syncthreads();
  uint32_t r = inc - v + lds[wid];
  total = lds[16];
  __syncthreads();
  return r;
}

// in may alias out
__global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                    uint32_t* __restrict__ tile_sums, const uint32_t* __restrict__ d_n) {
  __shared__ uint32_t lds[17];
  const uint32_t n = *d_n;
  const uint32_t base = blockIdx.x * SCAN_TILE;
  if (base >= n) return;
  uint32_t v[8];
  const uint32_t i0 = base + threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { uint32_t t = v[k]; v[k] = s; s += t; }
  uint32_t total;
  uint32_t off = block_excl_scan(s, lds, total);
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (i0 + k < n) out[i0 + k] = v[k] + off;
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// one block of 1024 threads; writes exclusive tile offsets in place and the grand total to *d_total
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t* __restrict__ tile_sums, const uint32_t* __restrict__ d_n,
                                                    uint32_t* __restrict__ d_total) {
  __shared__ uint32_t lds[17];
  const uint32_t n = *d_n;
  const uint32_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t v[8];
  const uint32_t i0 = threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = (i0 + k < ntiles) ? tile_sums[i0 + k] : 0u;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { uint32_t t = v[k]; v[k] = s; s += t; }
  uint32_t total;
  uint32_t off = block_excl_scan(s, lds, total);
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (i0 + k < ntiles) tile_sums[i0 + k] = v[k] + off;
  if (threadIdx.x == 0 && d_total) *d_total = total;
}

// out[i] += tile offset; also writes out[n] = total (so out is a proper "starts" array of n+1 entries)
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_sums,
                                                  const uint32_t* __restrict__ d_n, const uint32_t* __restrict__ d_total) {
  const uint32_t n = *d_n;
  const uint32_t base = blockIdx.x * SCAN_TILE;
  if (base >= n) return;
  const uint32_t off = tile_sums[blockIdx.x];
  const uint32_t i0 = base + threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (i0 + k < n) out[i0 + k] += off;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *d_total;
}

// host launcher.  max_n bounds the launch; the real count is *d_n (<= max_n).  out needs max_n+1 entries.
inline void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* tile_sums /* >= 8192 */, const uint32_t* d_n,
                               uint32_t* d_total, uint32_t max_n, hipStream_t st) {
  const uint32_t ntiles = (max_n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, st, in, out, tile_sums, d_n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, tile_sums, d_n, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(256), 0, st, out, tile_sums, d_n, d_total);
}

}  // namespace loamx
