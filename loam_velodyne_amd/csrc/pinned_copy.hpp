// Host -> device transfers of the single-sweep entry points by a KERNEL that reads the pinned host block over PCIe itself.
// Measured (profiles/r05_ab.md section 6): on these latency-bound chains a lone hipMemcpyAsync of a few hundred KB goes to an SDMA
// engine, and the hand-overs between that queue and the compute queue cost far more than the copy — ~180 us per VLP-16 sweep in front
// of the feature extraction; a kernel on the same stream costs ~10 us and needs no hand-over.  Blocks of many MiB (the batched
// pipeline's staging) stay on the copy engines (hostlink.hpp).
#pragma once
#include "common.h"

namespace loamx {

__global__ __launch_bounds__(256) static void k_fetch_pinned(float4* __restrict__ dst, const float4* __restrict__ host_src, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = host_src[i];
}

// dst[0..n) <- host_src[0..n), asynchronous on `st`; host_src must stay unchanged until the stream has passed this point.  host_src in
// memory the runtime has pinned (the handles' own staging blocks always are) and n <= 256 Ki points: by kernel; otherwise hipMemcpyAsync.
inline void fetch_from_pinned(float4* dst, const void* host_src, size_t n, hipStream_t st) {
  if (!n) return;
  void* d = nullptr;
  if (n <= ((size_t)1 << 18) && hipHostGetDevicePointer(&d, const_cast<void*>(host_src), 0) == hipSuccess && d) {
    hipLaunchKernelGGL(k_fetch_pinned, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, dst, (const float4*)d, (uint32_t)n);
    return;
  }
  (void)hipGetLastError();
  LX_HIP(hipMemcpyAsync(dst, host_src, sizeof(float4) * n, hipMemcpyHostToDevice, st));
}

// Device -> host of a few KB of 32-bit words by a kernel that stores into the pinned block (visible to the host once the stream has passed
// the launch).  Round 6: the sequential-SLAM chain's 19 KB cube histogram went down by hipMemcpyAsync, and that call BLOCKED the calling
// thread for 6.5-9 ms once at the start of every process and again every few hundred sweeps (LOAMX_MAP_TRACE, profiles/r06_ab.md
// section 16) — the "slow window" of the sequential figures.
__global__ __launch_bounds__(256) static void k_store_pinned_u32(uint32_t* __restrict__ host_dst, const uint32_t* __restrict__ src, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) host_dst[i] = src[i];
}
inline void store_to_pinned_u32(uint32_t* host_dst, const uint32_t* src, size_t n, hipStream_t st) {
  if (!n) return;
  void* d = nullptr;
  if (n <= ((size_t)1 << 20) && hipHostGetDevicePointer(&d, host_dst, 0) == hipSuccess && d) {
    hipLaunchKernelGGL(k_store_pinned_u32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)d, src, (uint32_t)n);
    return;
  }
  (void)hipGetLastError();
  LX_HIP(hipMemcpyAsync(host_dst, src, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
}

}  // namespace loamx
