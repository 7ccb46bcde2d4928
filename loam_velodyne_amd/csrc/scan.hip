// Device-wide exclusive scan kernels (see scan.hpp).
#include "scan.hpp"
#include "common.h"
#include <mutex>
#include <unordered_map>

namespace loamx {

void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* tile_sums /* SCAN_SCRATCH_WORDS, zero-filled once */, const uint32_t* d_n,
                               uint32_t* d_total, uint32_t max_n, hipStream_t st, uint32_t* out2, uint32_t* zero_in) {
  // one launch (the scratch holds the chained scan's state: 64-bit words)
  exclusive_scan_u32_chained(in, out, (unsigned long long*)tile_sums, d_n, d_total, max_n, st, out2, zero_in);
}

// ---- one-launch variant (see scan.hpp) ---------------------------------------------------------------------------------------------
// state[0]: epoch of the last finished launch (bumped by the last tile to finish), state[1]: tiles finished, state[8 + b]: tile b's word
//   word = epoch << 34 | flag << 32 | value;  flag 1: value = the tile's own sum, flag 2: value = inclusive prefix up to and with the tile
constexpr int CS_HDR = 8;

__global__ __launch_bounds__(256) void k_scan_chained(const uint32_t* in, uint32_t* __restrict__ out, unsigned long long* __restrict__ state,
                                                      const uint32_t* __restrict__ d_n, uint32_t n_host, uint32_t* __restrict__ d_total,
                                                      uint32_t* __restrict__ out2, uint32_t* zero_in, uint32_t* __restrict__ err_flag) {
  __shared__ uint32_t lds[17];
  __shared__ uint32_t s_excl;
  uint32_t n = n_host != 0xffffffffu ? n_host : *d_n;
  // a count beyond what this launch's grid covers (a caller's bound that was too small) is cut to the grid: every tile that counts
  // exists, the launch ends and leaves the state usable (the raised error word tells the host; out[] is then incomplete)
  if ((unsigned long long)n > (unsigned long long)gridDim.x * SCAN_TILE) {
    n = gridDim.x * (uint32_t)SCAN_TILE;
    if (blockIdx.x == 0 && threadIdx.x == 0 && err_flag) *err_flag = 1u;
  }
  const uint32_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  const uint32_t b = blockIdx.x;
  if (b >= ntiles && !(n == 0 && b == 0)) return;   // (nobody waits for a tile beyond the data)
  const unsigned long long epoch = (__hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull) & ((1ull << 30) - 1ull);
  if (n == 0) {   // empty input: out[0] = 0, total 0 — one tile, no chain
    if (threadIdx.x == 0) {
      out[0] = 0u;
      if (out2) out2[0] = 0u;
      if (d_total) *d_total = 0u;
      __hip_atomic_store(&state[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const uint32_t i0 = b * SCAN_TILE + threadIdx.x * 8;
  uint32_t v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
  if (zero_in) {
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (i0 + k < n && v[k]) zero_in[i0 + k] = 0u;
  }
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { uint32_t t = v[k]; v[k] = s; s += t; }
  uint32_t total;
  const uint32_t off = block_excl_scan(s, lds, total);
  // ---- the chain: publish the sum, look back (the wave's 64 lanes poll 64 predecessors at a time), publish the inclusive prefix
  if (threadIdx.x < 64) {
    bool failed = false;
    const uint32_t excl = chain_lookback(state + CS_HDR, epoch, b, total, failed);
    if (failed && threadIdx.x == 0 && err_flag) *err_flag = 1u;
    if (threadIdx.x == 0) s_excl = excl;
  }
  __syncthreads();
  const uint32_t base = s_excl + off;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (i0 + k < n) {
      const uint32_t r = v[k] + base;
      out[i0 + k] = r;
      if (out2) out2[i0 + k] = r;
    }
  if (b == ntiles - 1 && threadIdx.x == 0) {   // the last tile holds the grand total
    const uint32_t all = s_excl + total;
    out[n] = all;
    if (out2) out2[n] = all;
    if (d_total) *d_total = all;
  }
  // ---- the launch is over when every tile is: the last one to get here opens the next epoch
  if (threadIdx.x == 0) {
    const unsigned long long done = __hip_atomic_fetch_add(&state[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    if (done == ntiles) {
      __hip_atomic_store(&state[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static uint32_t* scan_err_word() {   // one pinned word per process
  static uint32_t* p = [] { uint32_t* q = nullptr; if (hipHostMalloc((void**)&q, 64, hipHostMallocDefault) != hipSuccess) return (uint32_t*)nullptr; *q = 0u; return q; }();
  return p;
}
void scan_check_errors() {
  uint32_t* p = scan_err_word();
  if (p && *(volatile uint32_t*)p) { *p = 0u; throw Error(LOAMX_E_HIP, "a chained scan gave up waiting for an earlier tile"); }
}

void exclusive_scan_u32_chained(const uint32_t* in, uint32_t* out, unsigned long long* state, const uint32_t* d_n, uint32_t* d_total,
                                uint32_t max_n, hipStream_t st, uint32_t* out2, uint32_t* zero_in, uint32_t n_host) {
  const uint32_t bound = n_host != 0xffffffffu ? n_host : max_n;
  const uint32_t ntiles = (bound + SCAN_TILE - 1) / SCAN_TILE;
  // The tiles' words carry a 30-bit epoch and are never cleared: after 2^30 launches on one state buffer (hours of live operation) a word
  // of a tile that has not been written since the epoch's previous life would pass for current.  The host counts the launches per state
  // buffer and clears the buffer on the stream before the epoch comes round (the kernel continues at epoch 1 from a zeroed state).
  {
    static std::mutex mu;
    static std::unordered_map<const void*, uint32_t> launches;
    bool clear = false;
    {
      std::lock_guard<std::mutex> lk(mu);
      uint32_t& k = launches[state];
      if (++k >= (1u << 30) - 8u) { k = 0; clear = true; }
    }
    if (clear) LX_HIP(hipMemsetAsync(state, 0, sizeof(unsigned long long) * chained_scan_state_words(), st));
  }
  hipLaunchKernelGGL(k_scan_chained, dim3(ntiles ? ntiles : 1u), dim3(256), 0, st, in, out, state, d_n, n_host, d_total, out2, zero_in, scan_err_word());
}

void exclusive_scan_u32_n(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, uint32_t* scratch2, uint32_t n, hipStream_t st) {
  // the count travels as a kernel argument
  exclusive_scan_u32_chained(in, out, (unsigned long long*)tile_sums, nullptr, scratch2 + 1, n, st, nullptr, nullptr, n);
}

}  // namespace loamx
