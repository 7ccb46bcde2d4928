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

// host launcher.  max_n bounds the launch; the real count is *d_n (<= max_n).  out needs max_n+1 entries and receives
// out[n] = total; in may alias out.  tile_sums needs 8192 entries.
void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* tile_sums, const uint32_t* d_n, uint32_t* d_total,
                        uint32_t max_n, hipStream_t st, uint32_t* out2 = nullptr,   // out2: optional second copy of the result
                        uint32_t* zero_in = nullptr);   // zero_in (= in, when in != out): the input is cleared behind the scan

// ---- the scan itself (round 4; the three-launch version it replaced is in the history): chained tiles with a decoupled look-back.  Tile b publishes its sum in a 64-bit word, walks back over the published sums / inclusive prefixes of the tiles
// before it, publishes its own inclusive prefix and writes its slice.  The words carry the launch's epoch, so nothing is cleared between
// launches; `state` (chained_scan_state_words() uint64 words) is zero-filled ONCE by its owner and may be shared by all scans of one HIP
// stream.  Everything a tile needs from another one travels INSIDE those words: relaxed agent-scope atomics, no cache-wide fence
// (dev_math.cuh "exchange").  Workgroups start in index order (as everywhere in this library where a workgroup waits for a
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
