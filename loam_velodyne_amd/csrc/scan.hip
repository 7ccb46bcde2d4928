// Device-wide exclusive scan kernels (see scan.cuh).
#include "scan.cuh"
#include "common.h"

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

void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* tile_sums /* SCAN_SCRATCH_WORDS, zero-filled once */, const uint32_t* d_n,
                               uint32_t* d_total, uint32_t max_n, hipStream_t st, uint32_t* out2, uint32_t* zero_in) {
  if (scan_use_chained()) {   // one launch (the scratch doubles as the chained scan's state: 64-bit words)
    exclusive_scan_u32_chained(in, out, (unsigned long long*)tile_sums, d_n, d_total, max_n, st, out2, zero_in);
    return;
  }
  const uint32_t ntiles = (max_n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, st, in, out, tile_sums, d_n, zero_in);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, tile_sums, d_n, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(256), 0, st, out, tile_sums, d_n, d_total, out2);
}

// ---- one-launch variant (see scan.cuh) ---------------------------------------------------------------------------------------------
// state[0]: epoch of the last finished launch (bumped by the last tile to finish), state[1]: tiles finished, state[8 + b]: tile b's word
//   word = epoch << 34 | flag << 32 | value;  flag 1: value = the tile's own sum, flag 2: value = inclusive prefix up to and with the tile
constexpr int CS_HDR = 8;
__device__ inline unsigned long long cs_word(unsigned long long epoch, unsigned flag, uint32_t v) { return (epoch << 34) | ((unsigned long long)flag << 32) | v; }

__global__ __launch_bounds__(256) void k_scan_chained(const uint32_t* in, uint32_t* __restrict__ out, unsigned long long* __restrict__ state,
                                                      const uint32_t* __restrict__ d_n, uint32_t n_host, uint32_t* __restrict__ d_total,
                                                      uint32_t* __restrict__ out2, uint32_t* zero_in, uint32_t* __restrict__ err_flag) {
  __shared__ uint32_t lds[17];
  __shared__ uint32_t s_excl;
  const uint32_t n = n_host != 0xffffffffu ? n_host : *d_n;
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
    unsigned long long* w = state + CS_HDR;
    const int lane = (int)threadIdx.x;
    uint32_t excl = 0u;
    if (b == 0) {
      if (lane == 0) __hip_atomic_store(&w[0], cs_word(epoch, 2u, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&w[b], cs_word(epoch, 1u, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int hi = (int)b;   // tiles [.., hi) are still to be accounted for
      bool failed = false;
      while (hi > 0 && !failed) {
        const int j = hi - 1 - lane;   // lane 0 looks at the nearest predecessor
        unsigned long long x = 0ull;
        uint32_t spins = 0;
        for (;;) {
          x = j >= 0 ? __hip_atomic_load(&w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cs_word(epoch, 2u, 0u);
          const bool ok = (x >> 34) == epoch && ((x >> 32) & 3ull) != 0ull;
          // usable: every lane up to the first inclusive prefix has a word of this epoch
          const unsigned long long okm = __ballot(ok), incm = __ballot(ok && ((x >> 32) & 3ull) == 2ull);
          const int first_inc = incm ? __builtin_ctzll(incm) : 64;
          const unsigned long long need = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
          if ((okm & need) == need) {   // lanes 0 .. first_inc are all in: add them up
            uint32_t part = (lane <= first_inc) ? (uint32_t)x : 0u;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
            excl += part;
            hi = first_inc < 64 ? 0 : hi - 64;   // an inclusive prefix closes the chain
            break;
          }
          if (++spins > (1u << 22)) { failed = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (failed && lane == 0 && err_flag) *err_flag = 1u;
      if (lane == 0) __hip_atomic_store(&w[b], cs_word(epoch, 2u, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_excl = excl;
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
bool scan_use_chained() { static const bool on = getenv("LOAMX_SCAN_3PASS") == nullptr; return on; }

void exclusive_scan_u32_chained(const uint32_t* in, uint32_t* out, unsigned long long* state, const uint32_t* d_n, uint32_t* d_total,
                                uint32_t max_n, hipStream_t st, uint32_t* out2, uint32_t* zero_in, uint32_t n_host) {
  const uint32_t bound = n_host != 0xffffffffu ? n_host : max_n;
  const uint32_t ntiles = (bound + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_chained, dim3(ntiles ? ntiles : 1u), dim3(256), 0, st, in, out, state, d_n, n_host, d_total, out2, zero_in, scan_err_word());
}

__global__ void k_scan_set_n(uint32_t* p, uint32_t v) { *p = v; }

void exclusive_scan_u32_n(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, uint32_t* scratch2, uint32_t n, hipStream_t st) {
  if (scan_use_chained()) {   // one launch instead of four: the count travels as a kernel argument
    exclusive_scan_u32_chained(in, out, (unsigned long long*)tile_sums, nullptr, scratch2 + 1, n, st, nullptr, nullptr, n);
    return;
  }
  hipLaunchKernelGGL(k_scan_set_n, dim3(1), dim3(1), 0, st, scratch2, n);
  exclusive_scan_u32(in, out, tile_sums, scratch2, scratch2 + 1, n, st);
}

}  // namespace loamx
