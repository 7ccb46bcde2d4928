// Device-wide exclusive scan kernels (see scan.cuh).
#include "scan.cuh"

namespace loamx {

// in may alias out
__global__ __launch_bounds__(256) void k_scan_tiles(const uint32_t* in, uint32_t* __restrict__ out,
                                                    uint32_t* __restrict__ tile_sums, const uint32_t* __restrict__ d_n, uint32_t* zero_in) {
  __shared__ uint32_t lds[17];
  const uint32_t n = *d_n;
  const uint32_t base = blockIdx.x * SCAN_TILE;
  if (base >= n) return;
  uint32_t v[8];
  const uint32_t i0 = base + threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
  if (zero_in) {   // the input is a histogram that its producer wants back empty (only entries that hold something are written)
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (i0 + k < n && v[k]) zero_in[i0 + k] = 0u;
  }
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
// out2 (optional): a second copy of the result (counting sorts keep one as the table and consume the other as cursors)
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_sums,
                                                  const uint32_t* __restrict__ d_n, const uint32_t* __restrict__ d_total,
                                                  uint32_t* __restrict__ out2) {
  const uint32_t n = *d_n;
  const uint32_t base = blockIdx.x * SCAN_TILE;
  if (base >= n) return;
  const uint32_t off = tile_sums[blockIdx.x];
  const uint32_t i0 = base + threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (i0 + k < n) {
      const uint32_t v = out[i0 + k] + off;
      out[i0 + k] = v;
      if (out2) out2[i0 + k] = v;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[n] = *d_total;
    if (out2) out2[n] = *d_total;
  }
}

void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* tile_sums /* >= 8192 */, const uint32_t* d_n,
                               uint32_t* d_total, uint32_t max_n, hipStream_t st, uint32_t* out2, uint32_t* zero_in) {
  const uint32_t ntiles = (max_n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, st, in, out, tile_sums, d_n, zero_in);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, tile_sums, d_n, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(256), 0, st, out, tile_sums, d_n, d_total, out2);
}

__global__ void k_scan_set_n(uint32_t* p, uint32_t v) { *p = v; }

void exclusive_scan_u32_n(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, uint32_t* scratch2, uint32_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_scan_set_n, dim3(1), dim3(1), 0, st, scratch2, n);
  exclusive_scan_u32(in, out, tile_sums, scratch2, scratch2 + 1, n, st);
}

}  // namespace loamx
