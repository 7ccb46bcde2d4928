// Device-wide exclusive scan of uint32 with the element count read from device memory (no host sync): one launch, tiles of 2048
// elements per 256-thread block chained by a decoupled look-back, up to 8192 tiles = 16 Mi elements.  wave64 shuffles inside a wave.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace loamx {

constexpr int SCAN_TILE = 2048;
constexpr uint32_t SCAN_MAX_N = 8192u * SCAN_TILE;
// the scratch every scan call is handed ("tile_sums"): uint32 words, ZERO-FILLED ONCE by its owner — the one-launch scan keeps its
// per-tile 64-bit words (epoch-tagged) in it
constexpr size_t SCAN_SCRATCH_WORDS = 2 * (8192 + 8);

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

// ---- decoupled look-back over a chain of workgroups (the chained scan's core; also how the feature extractor's per-ring outputs find
// their place without prefix / copy launches).  w[b]: workgroup b's 64-bit word = epoch << 34 | flag << 32 | value — flag 1: value = its
// own count, flag 2: value = the inclusive prefix up to and with it.  Called by the 64 lanes of ONE wave with uniform arguments:
// publishes `own`, adds up the words of the workgroups before b (64 at a time; stops at the first inclusive prefix), publishes the
// inclusive prefix and returns the exclusive one in every lane.  The words carry the launch's epoch (30 bits, never 0), so nothing is
// cleared between launches; everything travels inside the words (relaxed agent-scope atomics, no cache-wide fence).  Workgroups must
// start in index order; a predecessor that does not show up within ~1 s sets `failed` instead of hanging the device.
__device__ inline unsigned long long cs_word(unsigned long long epoch, unsigned flag, uint32_t v) { return (epoch << 34) | ((unsigned long long)flag << 32) | v; }
__device__ inline uint32_t chain_lookback(unsigned long long* __restrict__ w, unsigned long long epoch, uint32_t b, uint32_t own, bool& failed) {
  const int lane = (int)(threadIdx.x & 63);
  uint32_t excl = 0u;
  if (b == 0) {
    if (lane == 0) __hip_atomic_store(&w[0], cs_word(epoch, 2u, own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0u;
  }
  if (lane == 0) __hip_atomic_store(&w[b], cs_word(epoch, 1u, own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int hi = (int)b;   // workgroups [.., hi) are still to be accounted for
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
  if (lane == 0) __hip_atomic_store(&w[b], cs_word(epoch, 2u, excl + own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// host launcher.  max_n bounds the launch; the real count is *d_n (<= max_n).  out needs max_n+1 entries and receives
// out[n] = total; in may alias out.  tile_sums needs 8192 entries.
void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, const uint32_t* d_n, uint32_t* d_total,
                        uint32_t max_n, hipStream_t st, uint32_t* out2 = nullptr,   // out2: optional second copy of the result
                        uint32_t* zero_in = nullptr);   // zero_in (= in, when in != out): the input is cleared behind the scan

// ---- the scan itself (round 4; the three-launch version it replaced is in the history): chained tiles with a decoupled look-back.  Tile b publishes its sum in a 64-bit word, walks back over the published sums / inclusive prefixes of the tiles
// before it, publishes its own inclusive prefix and writes its slice.  The words carry the launch's epoch, so nothing is cleared between
// launches; `state` (chained_scan_state_words() uint64 words) is zero-filled ONCE by its owner and may be shared by all scans of one HIP
// stream.  Everything a tile needs from another one travels INSIDE those words: relaxed agent-scope atomics, no cache-wide fence
// (dev_math.hpp "exchange").  Workgroups start in index order (as everywhere in this library where a workgroup waits for a
// lower-numbered one); a predecessor that does not show up within ~1 s raises the process-wide error word (scan_check_errors()) instead
// of hanging the device.  n_host != UINT32_MAX: the element count is this value and d_n is ignored (no k_scan_set_n launch).
constexpr uint32_t CHAINED_SCAN_MAX_TILES = 8192;
inline size_t chained_scan_state_words() { return (size_t)CHAINED_SCAN_MAX_TILES + 8; }
void exclusive_scan_u32_chained(const uint32_t* in, uint32_t* out, unsigned long long* state, const uint32_t* d_n, uint32_t* d_total,
                                uint32_t max_n, hipStream_t st, uint32_t* out2 = nullptr, uint32_t* zero_in = nullptr,
                                uint32_t n_host = 0xffffffffu);
void scan_check_errors();   // throws LOAMX_E_HIP when a chained scan gave up waiting (after a synchronisation point)

// same with a host-known element count; scratch2 = two device words
void exclusive_scan_u32_n(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, uint32_t* scratch2, uint32_t n, hipStream_t st);

}  // namespace loamx
