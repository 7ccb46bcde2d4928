// Feature extraction kernels for gfx950 — reference src/lib/BasicScanRegistration.cpp:155-386.
//
//   k_feat_ring    one workgroup of six waves per scan ring.  Prologue (rounds 1-4: a kernel of its own): per point the curvature
//                  stencil (+-curvatureRegion neighbours on the ring, :293-306), the gap test of markAsPicked and the unreliable-point
//                  masks of setScanBuffersFor (:321-363), straight into LDS.  Then, per feature region, one wave: stable sort of the
//                  curvatures (:311-317, bitonic network in registers) and the order-dependent greedy picks (:198-235) done
//                  cooperatively: the 64 lanes test the next 64 candidates in sorted order, a ballot finds the first admissible one,
//                  and markAsPicked (:367-386) suppresses its neighbours before the next round — the sequential semantics are kept
//                  exactly.  The regions' picks run side by side with exact settling of the marks that cross a region boundary.
//   k_feat_compact the per-ring pick slots -> the three compact output clouds + per-sweep offsets, one launch (every workgroup sums the
//                  counts of the rings before it itself)
//   k_feat_lf_voxel / k_feat_lf_compact   per-ring pcl::VoxelGrid(0.2 m) of the less-flat candidates (:246-252) and its compaction
//   VoxelPipeline  the same voxel grid for rings longer than 4096 points.
#include "features.hpp"
#include "pinned_copy.hpp"
#include "scan.hpp"

namespace loamx {

__device__ inline float sqdiff3(const float4& a, const float4& b) {
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;
}
__device__ inline float sqdiff3w(const float4& a, const float4& b, float wb) {
  const float dx = a.x - b.x * wb, dy = a.y - b.y * wb, dz = a.z - b.z * wb;
  return dx * dx + dy * dy + dz * dz;
}

struct RingLds {
  uint8_t* flags;
  float* c;
  uint32_t* sorted;
  int8_t* label;
};

// LDS writes of this wave become visible to its own later LDS reads (one wave works on the pick lists at a time)
__device__ inline void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

// suppress the neighbours of a picked point (markAsPicked :367-386); one whole wave participates.
// gaps[k] = 1 when |p[k+1] - p[k]|^2 > 0.05 (ring-relative k)
__device__ inline void mark_as_picked(uint32_t scan_i, int cr, uint8_t* flags, const uint8_t* gaps, int lane) {
  bool brk = false;
  if (lane < cr) brk = gaps[scan_i + lane] != 0;                               // forward step i = lane+1: p[idx+i] vs p[idx+i-1]
  else if (lane >= 32 && lane < 32 + cr) brk = gaps[scan_i - (lane - 32) - 1] != 0;   // backward step i: p[idx-i] vs p[idx-i+1]
  const unsigned long long m = __ballot(brk);
  const uint32_t mf = (uint32_t)(m & 0xffffffffull), mb = (uint32_t)(m >> 32);
  const int nf = mf ? __builtin_ctz(mf) : cr;
  const int nb = mb ? __builtin_ctz(mb) : cr;
  if (lane == 0) flags[scan_i] = 1;
  if (lane < nf) flags[scan_i + lane + 1] = 1;
  if (lane >= 32 && lane - 32 < nb) flags[scan_i - (lane - 32) - 1] = 1;
  wave_lds_sync();
}

// The same, for a region whose picks run beside the other regions' (k_feat_ring): marks inside the wave's own region [r_lo, r_hi) go to
// `flags`, marks beyond its end go to `fwd` (they matter to the NEXT region only), marks in front of its start are dropped — the
// region before has made its picks before this one in the reference's order, so they can no longer change anything.
__device__ inline void mark_as_picked_split(uint32_t scan_i, int cr, uint8_t* flags, uint8_t* fwd, const uint8_t* gaps, int lane, uint32_t r_lo,
                                            uint32_t r_hi) {
  bool brk = false;
  if (lane < cr) brk = gaps[scan_i + lane] != 0;
  else if (lane >= 32 && lane < 32 + cr) brk = gaps[scan_i - (lane - 32) - 1] != 0;
  const unsigned long long m = __ballot(brk);
  const uint32_t mf = (uint32_t)(m & 0xffffffffull), mb = (uint32_t)(m >> 32);
  const int nf = mf ? __builtin_ctz(mf) : cr;
  const int nb = mb ? __builtin_ctz(mb) : cr;
  if (lane == 0) flags[scan_i] = 1;
  if (lane < nf) {
    const uint32_t t = scan_i + lane + 1;
    if (t < r_hi) flags[t] = 1;
    else fwd[t] = 1;
  }
  if (lane >= 32 && lane - 32 < nb) {
    const uint32_t t = scan_i - (lane - 32) - 1;
    if (t >= r_lo) flags[t] = 1;
  }
  wave_lds_sync();
}

// The order-dependent greedy picks of ONE region by one wave (BasicScanRegistration.cpp:197-235): corners from the largest curvature
// down, then flat points from the smallest up; 64 lanes test the next 64 candidates in sorted order, a ballot finds the first
// admissible one, its neighbours are suppressed before the next round.  The picks go to the region's own slots of the pick lists
// (outS / outLS / outF) and their counts to cnt3[0..2].  fwd == nullptr: every mark goes to `flags` (the sequential walk over the regions).
__device__ inline void region_picks(const FeatParams& P, int cr, uint32_t rn, uint32_t rgsp, uint32_t rscan, const float* rc, const uint32_t* rsorted,
                                    int8_t* rlabel, uint8_t* flags, uint8_t* fwd, const uint8_t* gaps, uint32_t* outS, uint32_t* outLS, uint32_t* outF,
                                    uint32_t* cnt3, int lane) {
  const uint32_t r_lo = rscan, r_hi = rscan + rn;
  uint32_t nS = 0, nLS = 0, nF = 0;
  {
    int picked = 0;
    int pos = (int)rn;
    while (pos > 0 && picked < P.max_less_sharp) {
      const int kk = pos - 1 - lane;
      const bool in = kk >= 0;
      const uint32_t e = in ? rsorted[kk] : 0u;
      const float ce = in ? rc[e] : 0.f;
      const bool above = in && (ce > P.curv_thr);
      const bool ok = above && flags[rscan + e] == 0;
      const unsigned long long mok = __ballot(ok), mstop = __ballot(in && !above);
      const int fo = mok ? __builtin_ctzll(mok) : 64, fs = mstop ? __builtin_ctzll(mstop) : 64;
      if (fo < fs) {
        const uint32_t pe = __shfl(e, fo, 64);
        picked++;
        if (lane == 0) {
          if (picked <= P.max_sharp) {
            rlabel[pe] = 2;
            outS[nS] = rgsp + pe;
          } else {
            rlabel[pe] = 1;
          }
          outLS[nLS] = rgsp + pe;
        }
        if (picked <= P.max_sharp) nS++;
        nLS++;
        if (fwd) mark_as_picked_split(rscan + pe, cr, flags, fwd, gaps, lane, r_lo, r_hi);
        else mark_as_picked(rscan + pe, cr, flags, gaps, lane);
        pos = pos - 1 - fo;
      } else if (fs < 64) {
        break;   // sorted: nothing further exceeds the threshold
      } else {
        pos -= 64;
      }
    }
  }
  {
    int picked = 0;
    int pos = 0;
    while (pos < (int)rn && picked < P.max_flat) {
      const int kk = pos + lane;
      const bool in = kk < (int)rn;
      const uint32_t e = in ? rsorted[kk] : 0u;
      const float ce = in ? rc[e] : 0.f;
      const bool below = in && (ce < P.curv_thr);
      const bool ok = below && flags[rscan + e] == 0;
      const unsigned long long mok = __ballot(ok), mstop = __ballot(in && !below);
      const int fo = mok ? __builtin_ctzll(mok) : 64, fs = mstop ? __builtin_ctzll(mstop) : 64;
      if (fo < fs) {
        const uint32_t pe = __shfl(e, fo, 64);
        picked++;
        if (lane == 0) {
          rlabel[pe] = -1;
          outF[nF] = rgsp + pe;
        }
        nF++;
        if (fwd) mark_as_picked_split(rscan + pe, cr, flags, fwd, gaps, lane, r_lo, r_hi);
        else mark_as_picked(rscan + pe, cr, flags, gaps, lane);
        pos = pos + fo + 1;
      } else if (fs < 64) {
        break;
      } else {
        pos += 64;
      }
    }
  }
  if (lane == 0) { cnt3[0] = nS; cnt3[1] = nLS; cnt3[2] = nF; }
  wave_lds_sync();
}

// One wave sorts up to 64 * KPL unique 64-bit keys (curvature bits << 32 | position) entirely in registers: element i of the
// network lives in lane i / KPL, register i % KPL.  Compare-exchanges whose partner distance is below KPL stay inside the lane;
// the others swap whole registers with the partner lane by shuffles — no LDS round trip and no fence per stage (the LDS version
// this replaces spent 23 us of the ring's 48 us here).  The keys are unique, so any correct network yields the reference's
// stable ascending order (:311-317); padding keys (~0) sort last.
template <int KPL>
__device__ inline void wave_sort_curvature(const float* __restrict__ c, uint32_t n, uint32_t* __restrict__ sorted, int lane) {
  unsigned long long a[KPL];
#pragma unroll
  for (int r = 0; r < KPL; r++) {   // (initial placement is arbitrary for a sorting network: conflict-free reads)
    const uint32_t e = (uint32_t)(r * 64 + lane);
    a[r] = e < n ? ((unsigned long long)__float_as_uint(c[e]) << 32) | e : ~0ull;
  }
  constexpr uint32_t N = 64u * KPL;
  // (k and the cross-lane j are run-time loop variables on purpose: fully unrolled, the 45-55 stages of five instantiations no longer
  // fit the instruction cache and the kernel ran 1.5x slower than the LDS version; only the register indices have to be static)
#pragma nounroll
  for (uint32_t k = 2; k <= N; k <<= 1) {
#pragma nounroll
    for (uint32_t j = k >> 1; j >= (uint32_t)KPL; j >>= 1) {   // partner in another lane
      const int pl = (int)(j / KPL);
      const bool lower = (lane & pl) == 0;
#pragma unroll
      for (int r = 0; r < KPL; r++) {
        const unsigned long long x = a[r];
        const unsigned long long y = __shfl_xor(x, pl, 64);
        const uint32_t i = (uint32_t)lane * KPL + (uint32_t)r;
        const bool up = (i & k) == 0;           // the same for both ends of the pair (j < k)
        const bool want_min = lower == up;
        a[r] = want_min ? (x < y ? x : y) : (x < y ? y : x);
      }
    }
#pragma unroll
    for (int j = KPL / 2; j >= 1; j >>= 1) {   // partner in the same lane (static register indices)
      if ((uint32_t)j <= (k >> 1)) {
#pragma unroll
        for (int r = 0; r < KPL; r++) {
          if ((r & j) == 0) {
            const uint32_t i = (uint32_t)lane * KPL + (uint32_t)r;
            const bool up = (i & k) == 0;
            const unsigned long long x = a[r], y = a[r | j];
            const bool sw = (x > y) == up;
            a[r] = sw ? y : x;
            a[r | j] = sw ? x : y;
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KPL; r++) {
    const uint32_t i = (uint32_t)lane * KPL + (uint32_t)r;
    if (i < n) sorted[i] = (uint32_t)a[r];
  }
}

#ifdef LOAMX_PROF_FEAT
__device__ unsigned long long g_feat_ts[8];
#define FT_TS(k) do { if (blockIdx.x == 100 && threadIdx.x == 0) g_feat_ts[k] = wall_clock64(); } while (0)
#else
#define FT_TS(k) do { } while (0)
#endif
constexpr int FEAT_WAVES = 6;   // regions sorted concurrently per ring

// one workgroup of FEAT_WAVES waves per ring.  Dynamic LDS: flags[flag_bytes] | gaps[flag_bytes] | per wave { c | sorted | label }[nmax]
// 5 waves per SIMD (<= 96 VGPRs): three 6-wave workgroups share a CU, and 512 rings on 256 CUs are not dealt two apiece
__global__ __launch_bounds__(64 * FEAT_WAVES) __attribute__((amdgpu_waves_per_eu(5))) void k_feat_ring(
    const float4* __restrict__ cloud, const uint32_t* __restrict__ ring_off, const uint32_t* __restrict__ ring_sweep_base, FeatParams P,
    uint32_t flag_bytes, uint32_t nmax, uint32_t sortP, float4* __restrict__ slotS,
    float4* __restrict__ slotLS, float4* __restrict__ slotF, uint32_t* __restrict__ cntS, uint32_t* __restrict__ cntLS,
    uint32_t* __restrict__ cntF, uint8_t* __restrict__ lf_valid, int force_sequential, uint32_t* __restrict__ bad_word) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint8_t* flags = (uint8_t*)smem;
  uint8_t* gaps = flags + flag_bytes;
  uint8_t* fwd = gaps + flag_bytes;   // marks a region's picks leave BEYOND its end (they concern the next region only)
  uint8_t* flags0 = fwd + flag_bytes; // the masks of setScanBuffersFor as they were before any pick (a region that is made again starts from them)
  float* curv_all = (float*)(smem + 4 * (size_t)flag_bytes);   // the ring's curvatures (flag_bytes entries)
  // picked point indices (global), region j's in its own slots [j * max, (j + 1) * max) of the three lists, compacted in region order
  // at the end; the points themselves are copied out after that
  uint32_t* pickS = (uint32_t*)(smem + 8 * (size_t)flag_bytes);
  uint32_t* pickLS = pickS + P.max_sharp * P.n_regions;
  uint32_t* pickF = pickLS + P.max_less_sharp * P.n_regions;
  char* wave_base = smem + ((8 * (size_t)flag_bytes + 4 * (size_t)((P.max_sharp + P.max_less_sharp + P.max_flat) * P.n_regions) + 15) & ~(size_t)15);
  const size_t wave_bytes = (size_t)nmax * 9;
  __shared__ uint32_t reg_n[FEAT_WAVES], reg_gsp[FEAT_WAVES], reg_scan[FEAT_WAVES], npick[3];
  __shared__ uint32_t rcnt[64][3];    // picks per region (n_regions <= 64)
  __shared__ int s_simple;

  const uint32_t r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t s0g = ring_off[r], len = ring_off[r + 1] - s0g;
  const int cr = P.curv_region, nreg = P.n_regions;
  const uint32_t capS = P.max_sharp * nreg, capLS = P.max_less_sharp * nreg, capF = P.max_flat * nreg;
  uint32_t nS = 0, nLS = 0, nF = 0;   // maintained by wave 0
  FT_TS(0);
  // the less-flat marks of the ring start from zero: the regions below write theirs (points outside every region — the first and last
  // curv_region of a ring, a region of one point — keep the zero); ordered before those writes by the prologue's barriers
  for (uint32_t k = tid; k < len; k += blockDim.x) lf_valid[s0g + k] = 0;
  if (len <= 2u * cr + 1u) {          // (too short for a curvature stencil: nothing is extracted, but the finite-input contract still holds)
    for (uint32_t k = tid; k < len; k += blockDim.x) {
      const float4 p = cloud[s0g + k];
      if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) __hip_atomic_store(bad_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (len > 2u * cr + 1u) {           // block-uniform
    const uint32_t base = ring_sweep_base[r];
    // indices relative to the sweep's cloud, exactly the values the reference's integer region formula sees (:180-183)
    const unsigned long long s0 = s0g - base, e0 = s0 + len - 1;
    // ---- the per-point pass of the ring (rounds 1-4: a kernel of its own, k_feat_point, with its results — 5 B per point — written to
    // HBM and read back here): curvature stencil (:293-306), the gap test of markAsPicked (:372, :380) and the unreliable-point masks of
    // setScanBuffersFor (:321-363), straight into LDS.  The sweep is read ONCE; the +-curvature_region neighbours come from L1.
    for (uint32_t k = tid; k < len; k += blockDim.x) { flags[k] = 0; fwd[k] = 0; }
    // Round 6: the ring's points go through LDS in chunks (each point is needed by 2 * curv_region + 3 points of the stencil and mask
    // tests: 13 loads per point through the L1 became one coalesced load + LDS reads; in-kernel stamps: this pass was 17-20 of a ring's
    // 41 us).  The chunks live in the per-wave sort buffers, which nothing touches before the prologue is over.  Same values, same
    // operations in the same order: bit-identical.
    float4* stg = (float4*)wave_base;
    const uint32_t stg_cap = (uint32_t)((FEAT_WAVES * wave_bytes) / sizeof(float4));
    const uint32_t halo = (uint32_t)(cr > 1 ? cr : 1);
    const bool staged = stg_cap >= 2u * halo + 64u;          // (block-uniform; otherwise — a huge curvature region — straight from memory)
    const uint32_t chunk = staged ? stg_cap - 2u * halo : len;
    for (uint32_t c0 = 0; c0 < len; c0 += chunk) {
    const uint32_t lo = c0 >= halo ? c0 - halo : 0u;
    const uint32_t c1 = c0 + chunk < len ? c0 + chunk : len;
    const uint32_t hi = c1 + halo < len ? c1 + halo : len;
    __syncthreads();                                         // (the flags above are cleared / the previous chunk has been read)
    if (staged) {
      for (uint32_t k = lo + tid; k < hi; k += blockDim.x) stg[k - lo] = cloud[s0g + k];
      __syncthreads();
    }
    // (two instantiations of the pass: LDS reads stay ds_read, not flat loads; point k of the ring is src[k - off])
    auto point_pass = [&](const auto* __restrict__ src_, const uint32_t off) {
    auto src = [&](uint32_t kk) -> float4 { return src_[kk - off]; };
    for (uint32_t k = c0 + tid; k < c1; k += blockDim.x) {
      const float4 p = src(k);
      // (binned rings are finite by contract; a caller that breaks it is told — LOAMX_E_INVALID at the call's synchronisation point)
      if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) __hip_atomic_store(bad_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // gap: the step to the next point exceeds markAsPicked's 0.05 m^2 limit
      gaps[k] = k + 1 < len ? (((double)sqdiff3(src(k + 1), p) > 0.05) ? 1 : 0) : 1;
      if (k < (uint32_t)cr || k + (uint32_t)cr > len - 1u) continue;
      // curvature (:293-306): diff = -2*cr*p + sum_j (p[i+j] + p[i-j])
      const float w = (float)(-2 * cr);
      float dx = w * p.x, dy = w * p.y, dz = w * p.z;
      for (int j = 1; j <= cr; j++) {
        const float4 a = src(k + j), b = src(k - j);
        dx += a.x + b.x;
        dy += a.y + b.y;
        dz += a.z + b.z;
      }
      curv_all[k] = dx * dx + dy * dy + dz * dz;
      if (k + (uint32_t)cr >= len - 1u) continue;   // the mask loop stops one short (:328)
      // setScanBuffersFor (:329-361); flags are only ever set to 1, so concurrent writers are benign
      const float4 prev = src(k - 1), next = src(k + 1);
      const float diffNext = sqdiff3(next, p);
      bool skip_beam_test = false;
      if ((double)diffNext > 0.1) {
        const float depth1 = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
        const float depth2 = sqrtf(next.x * next.x + next.y * next.y + next.z * next.z);
        if (depth1 > depth2) {
          const float wd = sqrtf(sqdiff3w(next, p, depth2 / depth1)) / depth2;
          if ((double)wd < 0.1) {
            for (int q = 0; q <= cr; q++) flags[k - cr + q] = 1;
            skip_beam_test = true;   // `continue` in the reference: the parallel-beam test is skipped
          }
        } else {
          const float wd = sqrtf(sqdiff3w(p, next, depth1 / depth2)) / depth1;
          if ((double)wd < 0.1)
            for (int q = 0; q <= cr; q++) flags[k + 1 + q] = 1;
        }
      }
      if (!skip_beam_test) {
        const float diffPrev = sqdiff3(p, prev);
        const float dis = p.x * p.x + p.y * p.y + p.z * p.z;
        if ((double)diffNext > 0.0002 * (double)dis && (double)diffPrev > 0.0002 * (double)dis) flags[k] = 1;
      }
    }
    };
    if (staged) point_pass((const float4*)stg, lo); else point_pass(cloud + s0g, 0u);
    }
    __syncthreads();
    for (uint32_t k = tid; k < len; k += blockDim.x) flags0[k] = flags[k];
    if (tid < 64) { rcnt[tid][0] = 0; rcnt[tid][1] = 0; rcnt[tid][2] = 0; }
    if (tid == 0) {
      // Regions side by side need every region to hold at least curv_region points: a pick's marks then reach into the NEXT region at
      // most.  Shorter (or skipped) regions — rings of a few dozen points — take the sequential walk.
      int simple = 1;
      for (int j = 0; j < nreg; j++) {
        const unsigned long long sp = ((s0 + cr) * (unsigned long long)(nreg - j) + (e0 - cr) * (unsigned long long)j) / nreg;
        const unsigned long long ep = ((s0 + cr) * (unsigned long long)(nreg - 1 - j) + (e0 - cr) * (unsigned long long)(j + 1)) / nreg - 1;
        if (!(ep > sp) || ep - sp + 1 < (unsigned long long)cr) simple = 0;
      }
      s_simple = simple && !force_sequential;
    }
    float* c = (float*)(wave_base + wid * wave_bytes);
    uint32_t* sorted = (uint32_t*)(c + nmax);
    int8_t* label = (int8_t*)(sorted + nmax);
    for (int jb = 0; jb < nreg; jb += FEAT_WAVES) {
      const int j = jb + wid;
      uint32_t n = 0, gsp = 0, scan_sp = 0;
      if (j < nreg) {
        const unsigned long long sp = ((s0 + cr) * (unsigned long long)(nreg - j) + (e0 - cr) * (unsigned long long)j) / nreg;
        const unsigned long long ep = ((s0 + cr) * (unsigned long long)(nreg - 1 - j) + (e0 - cr) * (unsigned long long)(j + 1)) / nreg - 1;
        if (ep > sp) {
          n = (uint32_t)(ep - sp + 1);
          gsp = base + (uint32_t)sp;      // global index of the region's first point
          scan_sp = (uint32_t)(sp - s0);  // ring-relative index of the region's first point
        }
      }
      if (lane == 0) { reg_n[wid] = n; reg_gsp[wid] = gsp; reg_scan[wid] = scan_sp; }
      for (uint32_t e = lane; e < n; e += 64) {
        c[e] = curv_all[scan_sp + e];
        label[e] = 0;   // SURFACE_LESS_FLAT
      }
      if (lane < 4 && n) c[n + lane] = __builtin_inff();   // padding for the 4-wide reads below (never counted)
      __syncthreads();
      FT_TS(1);
      // stable ascending order (:311-317): every wave sorts its own region with a bitonic network over unique 64-bit keys
      // (curvature bits << 32 | position: curvatures are sums of squares, so their bit patterns order like the values and
      // equal curvatures keep their input order) — in registers for regions of up to 512 points, through LDS beyond that
      // (16 keys per lane would cost the kernel its third workgroup per CU).
      if (sortP == 64) wave_sort_curvature<1>(c, n, sorted, lane);
      else if (sortP == 128) wave_sort_curvature<2>(c, n, sorted, lane);
      else if (sortP == 256) wave_sort_curvature<4>(c, n, sorted, lane);
      else if (sortP == 512) wave_sort_curvature<8>(c, n, sorted, lane);
      else {
        unsigned long long* keys = (unsigned long long*)(wave_base + FEAT_WAVES * wave_bytes + (size_t)wid * sortP * 8);
        for (uint32_t e = lane; e < sortP; e += 64)
          keys[e] = e < n ? ((unsigned long long)__float_as_uint(c[e]) << 32) | e : ~0ull;
        wave_lds_sync();
        for (uint32_t k = 2; k <= sortP; k <<= 1) {
          for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < sortP / 2; t += 64) {
              const uint32_t i = ((t / j) * 2 * j) + (t % j), l = i + j;
              const unsigned long long a = keys[i], b = keys[l];
              const bool up = (i & k) == 0;
              if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            wave_lds_sync();
          }
        }
        for (uint32_t e = lane; e < n; e += 64) sorted[e] = (uint32_t)keys[e];
      }
      __syncthreads();
      FT_TS(2);
      // The order-dependent greedy picks.  In the reference the regions of a ring are walked one after the other and a pick suppresses
      // its +-curv_region neighbours in a flag array of the whole RING (:367-386) — so a region is not independent of the one before it:
      // marks of picks near that region's end reach into this one's first points.  Nothing else couples them.  So every wave makes
      // the picks of its own region at once, as if no such marks came in, and leaves the marks that go out beyond its end in `fwd`;
      // then the boundaries are settled in order: a region is made again (by wave 0, with the incoming marks in place) exactly when one
      // of the points it picked carries a mark of the — by then final — region before it.  An incoming mark on a point that was
      // not picked cannot change anything: the walk skipped it or never reached it.  Bit-identical to the sequential walk
      // (tests/test_gpu_features.py), which rings with regions shorter than curv_region still take.
      if (s_simple) {
        const uint32_t rn = reg_n[wid];
        if (rn) {
          if (wid == 0 && jb > 0)   // the previous group's last region is final: its marks come first
            for (uint32_t e = lane; e < (uint32_t)cr && e < rn; e += 64) flags[reg_scan[0] + e] |= fwd[reg_scan[0] + e];
          wave_lds_sync();
          region_picks(P, cr, rn, reg_gsp[wid], reg_scan[wid], c, sorted, label, flags, fwd, gaps, pickS + (size_t)j * P.max_sharp,
                       pickLS + (size_t)j * P.max_less_sharp, pickF + (size_t)j * P.max_flat, rcnt[j], lane);
        }
        __syncthreads();
        if (wid == 0) {
          for (int w = 1; w < FEAT_WAVES && jb + w < nreg; w++) {
            const uint32_t rn2 = reg_n[w], rscan2 = reg_scan[w], rgsp2 = reg_gsp[w];
            const float* rc = (const float*)(wave_base + w * wave_bytes);
            const uint32_t* rsorted = (const uint32_t*)(rc + nmax);
            int8_t* rlabel = (int8_t*)(rsorted + nmax);
            const bool hit = lane < cr && (uint32_t)lane < rn2 && fwd[rscan2 + lane] != 0 && rlabel[lane] != 0;
            if (__ballot(hit)) {   // one of its picks was not admissible: again, with the marks of the region before it in place
              for (uint32_t e = lane; e < rn2; e += 64) {
                flags[rscan2 + e] = flags0[rscan2 + e] | (e < (uint32_t)cr ? fwd[rscan2 + e] : (uint8_t)0);
                rlabel[e] = 0;
              }
              for (uint32_t e = lane; e < (uint32_t)cr; e += 64) fwd[rscan2 + rn2 + e] = 0;   // (its own marks beyond its end: made anew)
              wave_lds_sync();
              region_picks(P, cr, rn2, rgsp2, rscan2, rc, rsorted, rlabel, flags, fwd, gaps, pickS + (size_t)(jb + w) * P.max_sharp,
                           pickLS + (size_t)(jb + w) * P.max_less_sharp, pickF + (size_t)(jb + w) * P.max_flat, rcnt[jb + w], lane);
            }
          }
        }
      } else if (wid == 0) {
        for (int w = 0; w < FEAT_WAVES && jb + w < nreg; w++) {
          const uint32_t rn2 = reg_n[w];
          if (rn2 == 0) continue;
          const float* rc = (const float*)(wave_base + w * wave_bytes);
          const uint32_t* rsorted = (const uint32_t*)(rc + nmax);
          int8_t* rlabel = (int8_t*)(rsorted + nmax);
          region_picks(P, cr, rn2, reg_gsp[w], reg_scan[w], rc, rsorted, rlabel, flags, nullptr, gaps, pickS + (size_t)(jb + w) * P.max_sharp,
                       pickLS + (size_t)(jb + w) * P.max_less_sharp, pickF + (size_t)(jb + w) * P.max_flat, rcnt[jb + w], lane);
        }
      }
      __syncthreads();
      FT_TS(3);
      // less-flat candidates: everything in the region that is not a corner (:238-242)
      for (uint32_t e = lane; e < n; e += 64) lf_valid[gsp + e] = label[e] <= 0 ? 1 : 0;
      __syncthreads();
    }
  }
  // the regions' pick lists, one behind the other in region order (wave 0; a list only ever moves towards the front)
  if (wid == 0 && len > 2u * cr + 1u) {
    uint32_t* lists[3] = {pickS, pickLS, pickF};
    const uint32_t per[3] = {(uint32_t)P.max_sharp, (uint32_t)P.max_less_sharp, (uint32_t)P.max_flat};
    uint32_t tot[3] = {0, 0, 0};
    for (int j = 0; j < nreg; j++) {
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const uint32_t cnt = rcnt[j][q], src = (uint32_t)j * per[q];
        if (src != tot[q]) {
          for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
            const uint32_t k = k0 + lane;
            const uint32_t v = k < cnt ? lists[q][src + k] : 0u;
            wave_lds_sync();
            if (k < cnt) lists[q][tot[q] + k] = v;
            wave_lds_sync();
          }
        }
        tot[q] += cnt;
      }
    }
    nS = tot[0]; nLS = tot[1]; nF = tot[2];
  }
  if (tid == 0) {
    cntS[r] = nS;
    cntLS[r] = nLS;
    cntF[r] = nF;
    npick[0] = nS; npick[1] = nLS; npick[2] = nF;
  }
  __syncthreads();
  for (uint32_t k = tid; k < npick[0]; k += blockDim.x) slotS[(size_t)r * capS + k] = cloud[pickS[k]];
  for (uint32_t k = tid; k < npick[1]; k += blockDim.x) slotLS[(size_t)r * capLS + k] = cloud[pickLS[k]];
  for (uint32_t k = tid; k < npick[2]; k += blockDim.x) slotF[(size_t)r * capF + k] = cloud[pickF[k]];
  FT_TS(4);
}

// ----------------------------------------------------------------------------------------------------------------
// Per-ring pcl::VoxelGrid of the less-flat candidates (:246-252) in ONE workgroup per ring: the ring's candidates
// (<= 4096) are compacted, keyed (PCL's linear voxel index << 12 | position), sorted in LDS by a stable LSD radix sort on
// the voxel index (8-bit digits, only as many passes as the index has bits — typically 3; the bitonic network this replaces
// needed 66 stages), and every voxel run is averaged by one thread in input order.  Same membership, order and float
// arithmetic as the generic VoxelPipeline (which remains the fallback for longer rings); output goes to the ring's own slots.
// ----------------------------------------------------------------------------------------------------------------
constexpr int LFV_THREADS = 1024;
constexpr uint32_t LFV_MAX = 4096;

__global__ __launch_bounds__(LFV_THREADS) void k_feat_lf_voxel(const float4* __restrict__ cloud, const uint32_t* __restrict__ ring_off,
                                                               const uint8_t* __restrict__ lf_valid, float inv_leaf, uint32_t P,
                                                               float4* __restrict__ slots, uint32_t* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = (unsigned long long*)smem;   // P entries
  __shared__ uint32_t sc[17];
  __shared__ int mm[6];
  __shared__ uint32_t s_wcnt[LFV_THREADS / 64][256];   // radix passes: per wave digit counts, then exclusive prefixes over the waves
  __shared__ uint32_t s_base[256];
  const uint32_t r = blockIdx.x, tid = threadIdx.x;
  const uint32_t s0 = ring_off[r], len = ring_off[r + 1] - s0;
  if (tid < 3) mm[tid] = 2147483647;
  else if (tid < 6) mm[tid] = -2147483647 - 1;
  for (uint32_t k = tid; k < P; k += LFV_THREADS) keys[k] = ~0ull;
  __syncthreads();
  // ---- compact the candidates (order preserved): keys[j] temporarily holds the ring-relative index
  uint32_t base = 0;
  for (uint32_t b = 0; b < len; b += LFV_THREADS) {
    const uint32_t i = b + tid;
    const uint32_t v = (i < len && lf_valid[s0 + i]) ? 1u : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan(v, sc, total);
    if (v) keys[base + ex] = i;
    base += total;
  }
  const uint32_t m = base;
  __syncthreads();
  // ---- voxel coordinates and their bounds
  {   // per thread, then per wave (shuffles), then one LDS atomic per wave and bound: 2048 candidates hitting six LDS words
      // one atomic at a time was the longest part of this kernel
    int lo[3] = {2147483647, 2147483647, 2147483647}, hi[3] = {-2147483647 - 1, -2147483647 - 1, -2147483647 - 1};
    for (uint32_t j = tid; j < m; j += LFV_THREADS) {
      const float4 p = cloud[s0 + (uint32_t)keys[j]];
      const int c[3] = {(int)floorf(p.x * inv_leaf), (int)floorf(p.y * inv_leaf), (int)floorf(p.z * inv_leaf)};
#pragma unroll
      for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], c[a]); hi[a] = max(hi[a], c[a]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        lo[a] = min(lo[a], __shfl_xor(lo[a], off, 64));
        hi[a] = max(hi[a], __shfl_xor(hi[a], off, 64));
      }
    }
    if ((tid & 63u) == 0) {
#pragma unroll
      for (int a = 0; a < 3; a++) { atomicMin(&mm[a], lo[a]); atomicMax(&mm[3 + a], hi[a]); }
    }
  }
  __syncthreads();
  const long long dx = (long long)mm[3] - mm[0] + 1, dy = (long long)mm[4] - mm[1] + 1, dz = (long long)mm[5] - mm[2] + 1;
  // PCL: "leaf size is too small" (more than INT_MAX voxels) -> the cloud passes through unfiltered
  const bool pass = m > 0 && (dx > 2147483647LL || dy > 2147483647LL || dx * dy > 2147483647LL || dz > 2147483647LL / (dx * dy) + 1 || dx * dy * dz > 2147483647LL);
  for (uint32_t j = tid; j < m; j += LFV_THREADS) {
    const uint32_t li = (uint32_t)keys[j];
    unsigned long long k;
    if (pass) {
      k = (unsigned long long)j;
    } else {
      const float4 p = cloud[s0 + li];
      const int jx = (int)floorf(p.x * inv_leaf), jy = (int)floorf(p.y * inv_leaf), jz = (int)floorf(p.z * inv_leaf);
      k = (unsigned long long)((long long)(jx - mm[0]) + (long long)(jy - mm[1]) * dx + (long long)(jz - mm[2]) * dx * dy);
    }
    keys[j] = (k << 12) | li;   // li < 4096: unique elements; equal voxels are in input order and a stable sort keeps them so
  }
  __syncthreads();
  // ---- stable LSD radix sort on the voxel index.  Wave w owns the elements [w * per_wave, (w + 1) * per_wave) in slots of 64
  // consecutive ones, so (wave, slot, lane) is the current order: per slot 8 ballots find the lanes with the same digit, the
  // wave's running digit counts give the rank inside the wave, an exclusive prefix over the 16 waves and over the 256 digit
  // totals the destination.  The second key buffer is the area that later holds the gathered points.
  unsigned long long* kb[2] = {keys, (unsigned long long*)(smem + (size_t)P * 8)};
  uint32_t npass = 0;
  if (!pass && m > 1) {
    const unsigned long long top = (unsigned long long)(dx * dy * dz - 1);
    const uint32_t bits = top ? 64u - (uint32_t)__builtin_clzll(top) : 0u;
    npass = (bits + 7u) / 8u;
  }
  {
    const uint32_t lane = tid & 63u, wid = tid >> 6;
    const uint32_t per_wave = ((m + LFV_THREADS - 1) / LFV_THREADS) * 64u;   // <= 256
    for (uint32_t ps = 0; ps < npass; ps++) {
      const uint32_t shift = 12u + 8u * ps;
      const unsigned long long* src = kb[ps & 1];
      unsigned long long* dst = kb[(ps & 1) ^ 1];
      for (uint32_t e = tid; e < (LFV_THREADS / 64) * 256; e += LFV_THREADS) (&s_wcnt[0][0])[e] = 0u;
      __syncthreads();
      unsigned long long key[4];
      uint32_t rank[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t i = wid * per_wave + (uint32_t)j * 64u + lane;
        const bool in = (uint32_t)j * 64u < per_wave && i < m;
        key[j] = in ? src[i] : ~0ull;
        const uint32_t d = (uint32_t)((key[j] >> shift) & 255ull);
        unsigned long long mt = __ballot(in);
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const unsigned long long bal = __ballot((d >> b) & 1u);
          mt &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = mt & ((1ull << lane) - 1ull);
        const int leader = __builtin_ctzll(mt | (1ull << 63));
        uint32_t old = 0u;
        if (in && (int)lane == leader) {
          old = s_wcnt[wid][d];
          s_wcnt[wid][d] = old + (uint32_t)__popcll(mt);
        }
        old = __shfl(old, leader, 64);
        rank[j] = old + (uint32_t)__popcll(below);
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
      uint32_t run = 0;
      if (tid < 256) {
#pragma unroll
        for (int w = 0; w < LFV_THREADS / 64; w++) { const uint32_t c = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += c; }
      }
      uint32_t tot;
      const uint32_t ex = block_excl_scan(run, sc, tot);   // (threads >= 256 contribute 0 after the 256 digits)
      if (tid < 256) s_base[tid] = ex;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t i = wid * per_wave + (uint32_t)j * 64u + lane;
        if ((uint32_t)j * 64u < per_wave && i < m) {
          const uint32_t d = (uint32_t)((key[j] >> shift) & 255ull);
          dst[s_base[d] + s_wcnt[wid][d] + rank[j]] = key[j];
        }
      }
      __syncthreads();
    }
    if (npass & 1u) {   // the result sits in the area the points are gathered into next
      for (uint32_t j = tid; j < m; j += LFV_THREADS) keys[j] = kb[1][j];
      __syncthreads();
    }
  }
  // ---- voxel heads -> output positions -> means.  The sorted points are first gathered into LDS (one parallel gather)
  // so that the in-order sums below do not chase cloud[] one dependent load after the other.
  float4* sp = (float4*)(smem + (size_t)P * 8);
  for (uint32_t j = tid; j < m; j += LFV_THREADS) sp[j] = cloud[s0 + (uint32_t)(keys[j] & 4095ull)];
  __syncthreads();
  uint32_t obase = 0;
  for (uint32_t b = 0; b < m; b += LFV_THREADS) {
    const uint32_t j = b + tid;
    const bool head = j < m && (j == 0 || (keys[j] >> 12) != (keys[j - 1] >> 12));
    uint32_t total;
    const uint32_t ex = block_excl_scan(head ? 1u : 0u, sc, total);
    if (head) {
      const unsigned long long vk = keys[j] >> 12;
      float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
      uint32_t e = j;
      do {
        const float4 p = sp[e];
        sx += p.x; sy += p.y; sz += p.z; si += p.w;
        e++;
      } while (e < m && (keys[e] >> 12) == vk);
      const float c = (float)(e - j);
      slots[s0 + obase + ex] = make_float4(sx / c, sy / c, sz / c, si / c);
    }
    obase += total;
  }
  if (tid == 0) cnt[r] = obase;
}

// ---- compaction of the per-ring slots into the output clouds, ONE launch per family (rounds 1-4: a prefix launch, a copy launch and a
// per-sweep offset launch for the picks; a prefix and a copy launch for the less-flat voxels).  A workgroup finds its ring's place
// itself: the sum of the counts of the rings before it (<= a few thousand words from L2, one wave-wide reduction) — no workgroup waits
// for another.  The first ring of a sweep records the sweep's offsets, the last ring the totals.
__device__ inline uint32_t rings_before(const uint32_t* __restrict__ c, uint32_t r, int lane) {
  uint32_t part = 0u;
  for (uint32_t i = (uint32_t)lane; i < r; i += 64u) part += c[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
  return part;
}
// grid = (nring, 3), 64 threads
__global__ __launch_bounds__(64) void k_feat_compact(const float4* __restrict__ s0, const float4* __restrict__ s1, const float4* __restrict__ s2,
                                                     const uint32_t* __restrict__ c0, const uint32_t* __restrict__ c1, const uint32_t* __restrict__ c2,
                                                     uint32_t cap0, uint32_t cap1, uint32_t cap2, float4* __restrict__ o0, float4* __restrict__ o1,
                                                     float4* __restrict__ o2, uint32_t* __restrict__ offs, const uint32_t* __restrict__ sweep_ring_base,
                                                     uint32_t nsw, uint32_t* __restrict__ host_offs) {
  const uint32_t r = blockIdx.x, kind = blockIdx.y, nring = gridDim.x;
  const int lane = (int)threadIdx.x;
  const float4* s = kind == 0 ? s0 : (kind == 1 ? s1 : s2);
  const uint32_t* c = kind == 0 ? c0 : (kind == 1 ? c1 : c2);
  const uint32_t cap = kind == 0 ? cap0 : (kind == 1 ? cap1 : cap2);
  float4* o = kind == 0 ? o0 : (kind == 1 ? o1 : o2);
  const uint32_t dst = rings_before(c, r, lane), n = c[r];
  for (uint32_t k = (uint32_t)lane; k < n; k += 64u) o[dst + k] = s[(size_t)r * cap + k];
  if (lane == 0) {
    uint32_t* off = offs + (size_t)kind * (nsw + 1);
    // host_offs (linked single-sweep handles): the same table in pinned host memory — the host reads the three sizes behind an event
    // recorded right after this launch, while the less-flat voxel grid is still running (FeatureExtractor::device_results_front)
    uint32_t* hoff = host_offs ? host_offs + (size_t)kind * (nsw + 1) : nullptr;
    uint32_t lo = 0, hi = nsw + 1;   // first sweep whose first ring is >= r
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sweep_ring_base[mid] < r) lo = mid + 1; else hi = mid; }
    for (uint32_t sw = lo; sw <= nsw && sweep_ring_base[sw] == r; sw++) { off[sw] = dst; if (hoff) hoff[sw] = dst; }
    if (r + 1 == nring)   // (sweeps that start behind the last ring — the entry [nsw] among them — hold the total)
      for (uint32_t sw = nsw; sweep_ring_base[sw] >= nring; sw--) { off[sw] = dst + n; if (hoff) hoff[sw] = dst + n; if (sw == 0) break; }
  }
}
// grid = nring, 256 threads
__global__ __launch_bounds__(256) void k_feat_lf_compact(const float4* __restrict__ slots, const uint32_t* __restrict__ ring_off,
                                                         const uint32_t* __restrict__ cnt, uint32_t* __restrict__ lf_off, float4* __restrict__ out,
                                                         uint32_t* __restrict__ host_lf_off) {
  __shared__ uint32_t s_dst;
  const uint32_t r = blockIdx.x;
  if (threadIdx.x < 64) {
    const uint32_t d = rings_before(cnt, r, (int)threadIdx.x);
    if (threadIdx.x == 0) {
      s_dst = d;
      lf_off[r] = d;
      if (r + 1 == gridDim.x) lf_off[gridDim.x] = d + cnt[r];
      if (host_lf_off) {   // (pinned host mirror: FeatureExtractor::device_results_lf)
        host_lf_off[r] = d;
        if (r + 1 == gridDim.x) host_lf_off[gridDim.x] = d + cnt[r];
      }
    }
  }
  __syncthreads();
  const uint32_t n = cnt[r], src = ring_off[r], dst = s_dst;
  for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) out[dst + k] = slots[src + k];
}

// ----------------------------------------------------------------------------------------------------------------
FeatureExtractor::FeatureExtractor(int device, hipStream_t shared_stream) : device_(device) {
  select_device(device);
  if (shared_stream) {
    st_ = shared_stream;
  } else {
    LX_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    own_stream_ = true;
  }
  vox_.init(st_);
  h_bad_.reserve(16);
  h_bad_.p[0] = 0u;
  h_bad_.p[1] = 0u;
}

void FeatureExtractor::check_finite_input() {
  if (*(volatile uint32_t*)h_bad_.p) {
    *h_bad_.p = 0u;
    throw Error(LOAMX_E_INVALID, "a sweep holds non-finite coordinates (binned rings must be finite: BasicLaserOdometry.cpp:230 / MultiScanRegistration.cpp:187-196 drop such points before this stage)");
  }
}

FeatureExtractor::~FeatureExtractor() {
  if (ev_front_) (void)hipEventDestroy(ev_front_);
  if (ev_lf_) (void)hipEventDestroy(ev_lf_);
  if (own_stream_ && st_) (void)hipStreamDestroy(st_);
}

void FeatureExtractor::check_params_() const {
  LX_REQUIRE(params.curv_region >= 1 && params.curv_region <= 16, "curvature_region must be in [1,16]");
  LX_REQUIRE(params.n_regions >= 1 && params.n_regions <= 64, "n_feature_regions must be in [1,64]");
  LX_REQUIRE(params.max_sharp >= 0 && params.max_flat >= 0 && params.max_less_sharp >= params.max_sharp, "invalid pick limits");
  LX_REQUIRE(params.less_flat_leaf > 0.f, "less_flat_filter_size must be positive");
}

// host bookkeeping of a batch: ring offsets, sweep bases, longest ring
void FeatureExtractor::upload_async(uint32_t nsw, const loamx_cloud* clouds, const uint32_t* const* ring_size, const uint32_t* n_rings,
                                    hipStream_t copy_stream, hipEvent_t done, const std::function<bool(void*, const void*, size_t)>& big_copy) {
  LX_REQUIRE(nsw >= 1 && clouds && ring_size && n_rings && copy_stream && done, "invalid sweep batch");
  check_params_();
  LX_HIP(hipSetDevice(device_));
  bool packed = true;
  for (uint32_t s = 0; s < nsw; s++) {
    check_cloud(&clouds[s], false);
    uint32_t sum = 0;
    for (uint32_t r = 0; r < n_rings[s]; r++) sum += ring_size[s][r];
    LX_REQUIRE(sum == clouds[s].count, "ring sizes do not add up to the cloud size");
    packed = packed && clouds[s].stride == 16 && clouds[s].intensity_offset == 12;
  }
  layout_(nsw, ring_size, n_rings);
  allocate_(copy_stream);
  if (packed) {
    // clouds that lie back to back in the caller's memory (one block per step) go over in ONE copy: a copy call costs ~5 us of host
    // time, and the hand-over of a step must not take longer than the step
    for (uint32_t s = 0; s < nsw;) {
      uint32_t e = s + 1;
      size_t cnt = clouds[s].count;
      while (e < nsw && (const char*)clouds[e].data == (const char*)clouds[s].data + sizeof(float4) * cnt) cnt += clouds[e++].count;
      if (cnt && !(big_copy && big_copy(cloud_.p + h_pt_base_[s], clouds[s].data, sizeof(float4) * cnt)))
        LX_HIP(hipMemcpyAsync(cloud_.p + h_pt_base_[s], clouds[s].data, sizeof(float4) * cnt, hipMemcpyHostToDevice, copy_stream));
      s = e;
    }
  } else {
    h_cloud_.reserve(n_ + 1);
    for (uint32_t s = 0; s < nsw; s++) pack_cloud(&clouds[s], h_cloud_.p + h_pt_base_[s]);
    if (n_) LX_HIP(hipMemcpyAsync(cloud_.p, h_cloud_.p, sizeof(float4) * n_, hipMemcpyHostToDevice, copy_stream));
  }
  LX_HIP(hipEventRecord(done, copy_stream));
}
void FeatureExtractor::upload_device(uint32_t nsw, const float4* d_src, const uint32_t* src_off, const uint32_t* const* ring_size,
                                     const uint32_t* n_rings, hipStream_t copy_stream, hipEvent_t done) {
  LX_REQUIRE(nsw >= 1 && d_src && src_off && ring_size && n_rings && copy_stream && done, "invalid sweep batch");
  check_params_();
  LX_HIP(hipSetDevice(device_));
  layout_(nsw, ring_size, n_rings);
  allocate_(copy_stream);
  for (uint32_t s = 0; s < nsw; s++) {
    const uint32_t cnt = h_pt_base_[s + 1] - h_pt_base_[s];
    if (cnt) LX_HIP(hipMemcpyAsync(cloud_.p + h_pt_base_[s], d_src + src_off[s], sizeof(float4) * cnt, hipMemcpyDeviceToDevice, copy_stream));
  }
  LX_HIP(hipEventRecord(done, copy_stream));
}
void FeatureExtractor::layout_(uint32_t nsw, const uint32_t* const* ring_size, const uint32_t* n_rings) {
  nsw_ = nsw;
  h_ring_off_.assign(1, 0);
  h_ring_base_.assign(nsw + 1, 0);
  h_pt_base_.assign(nsw + 1, 0);
  h_ring_sweep_base_.clear();
  max_ring_len_ = 0;
  uint32_t pt = 0;
  for (uint32_t s = 0; s < nsw; s++) {
    uint32_t sum = 0;
    for (uint32_t r = 0; r < n_rings[s]; r++) {
      sum += ring_size[s][r];
      h_ring_off_.push_back(pt + sum);
      h_ring_sweep_base_.push_back(pt);
      max_ring_len_ = std::max(max_ring_len_, ring_size[s][r]);
    }
    pt += sum;
    h_pt_base_[s + 1] = pt;
    h_ring_base_[s + 1] = h_ring_base_[s] + n_rings[s];
  }
  n_ = pt;
  nring_ = h_ring_base_[nsw];
  LX_REQUIRE(nring_ >= 1, "no scan rings");
}

// device buffers for the laid-out batch + the small tables (the cloud itself is already in / copied to cloud_)
void FeatureExtractor::allocate_(hipStream_t table_stream) {
  const uint32_t nsw = nsw_;
  cloud_.reserve(n_ + 1);
  lf_valid_.reserve(n_ + 1);
  lf_out_.reserve(n_ + 1);
  lf_slots_.reserve(n_ + 1);
  // the four offset tables lie back to back — [sharp | less sharp | flat][nsw + 1], then the less-flat cloud's per-ring offsets
  // [nring + 1] — so that a consumer fetches them with ONE copy (Pipeline::launch_features)
  offs_.reserve((size_t)3 * (nsw + 1) + nring_ + 2);
  off_stride_ = nsw + 1;
  lf_cnt_.reserve(nring_ + 2);
  const uint32_t caps[3] = {(uint32_t)(params.max_sharp * params.n_regions), (uint32_t)(params.max_less_sharp * params.n_regions),
                            (uint32_t)(params.max_flat * params.n_regions)};
  for (int k = 0; k < 3; k++) {
    slots_[k].reserve((size_t)nring_ * caps[k] + 1);
    out_[k].reserve((size_t)nring_ * caps[k] + 1);
    slot_cnt_[k].reserve((size_t)nring_ + 2);
  }
  if (max_ring_len_ > 4096) vox_.reserve(n_ + 1, nring_);
  // the three layout tables lie back to back in one device block and travel through pinned memory in ONE copy — and only when the
  // layout changed: a sensor's sweeps usually repeat their ring sizes, and the block on the device is then already the right one
  hipStream_t ts = table_stream ? table_stream : st_;
  const size_t words = (size_t)2 * nring_ + nsw + 2;
  std::vector<uint32_t> tab(words);
  memcpy(tab.data(), h_ring_off_.data(), sizeof(uint32_t) * (nring_ + 1));
  memcpy(tab.data() + nring_ + 1, h_ring_sweep_base_.data(), sizeof(uint32_t) * nring_);
  memcpy(tab.data() + 2 * (size_t)nring_ + 1, h_ring_base_.data(), sizeof(uint32_t) * (nsw + 1));
  const bool same = tab_dev_.p && tab_dev_.cap >= words + 8 && tab == tab_last_;
  if (!same) {
    tab_dev_.reserve(words + 8);
    h_tab_.reserve(words + 8);
    memcpy(h_tab_.p, tab.data(), sizeof(uint32_t) * words);
    LX_HIP(hipMemcpyAsync(tab_dev_.p, h_tab_.p, sizeof(uint32_t) * words, hipMemcpyHostToDevice, ts));
    tab_last_.swap(tab);
  }
  ring_off_.p = tab_dev_.p;
  ring_sweep_base_.p = tab_dev_.p + nring_ + 1;
  sweep_ring_base_.p = tab_dev_.p + 2 * (size_t)nring_ + 1;
}

void FeatureExtractor::upload(uint32_t nsw, const loamx_cloud* clouds, const uint32_t* const* ring_size, const uint32_t* n_rings, bool allow_direct) {
  LX_REQUIRE(nsw >= 1 && clouds && ring_size && n_rings, "invalid sweep batch");
  check_params_();
  LX_HIP(hipSetDevice(device_));
  for (uint32_t s = 0; s < nsw; s++) {
    check_cloud(&clouds[s], false);
    uint32_t sum = 0;
    for (uint32_t r = 0; r < n_rings[s]; r++) sum += ring_size[s][r];
    LX_REQUIRE(sum == clouds[s].count, "ring sizes do not add up to the cloud size");
  }
  LX_HIP(hipStreamSynchronize(st_));   // (the staging block and the tables of the previous sweep are free: nothing to wait for unless the caller skipped its results)
  layout_(nsw, ring_size, n_rings);
  // one sweep of packed x y z intensity records in memory the runtime has pinned goes up straight from where it lies — only for the
  // callers that ask for it (allow_direct): the fetch kernel reads the CALLER's block after this function has returned, so the entry
  // point must either wait for the sweep's results itself (loamx_scanreg_process) or state the lifetime in its contract
  // (loamx_scanreg_process_linked).  Anything else goes through this object's own pinned staging block, packed before this returns.
  const bool direct = allow_direct && nsw == 1 && n_ && packed_layout(&clouds[0]) && host_pinned(clouds[0].data, sizeof(float4) * n_);
  if (!direct) {
    h_cloud_.reserve(n_ + 1);
    for (uint32_t s = 0; s < nsw; s++) pack_cloud(&clouds[s], h_cloud_.p + h_pt_base_[s]);
  }
  allocate_();
  // (by a kernel that reads the pinned block: no copy engine in front of the extraction — pinned_copy.hpp; no wait: the kernels follow
  // on the same stream)
  fetch_from_pinned(cloud_.p, direct ? clouds[0].data : (const void*)h_cloud_.p, n_, st_);
}

// One raw revolution (MultiScanRegistration::process, src/lib/MultiScanRegistration.cpp:160-238): records with x, y, z
// float32 at byte offsets 0/4/8 in sensor axes, firing order.  The sweep is binned into rings on the device (ingest.hip);
// only the ring sizes come back to the host, which needs them to lay out the per-ring work.
void FeatureExtractor::upload_raw(const void* raw_xyz, uint32_t count, uint32_t stride, float lower_deg, float upper_deg, uint32_t n_scan_rings) {
  LX_REQUIRE(raw_xyz || count == 0, "NULL raw cloud");
  LX_REQUIRE(stride >= 12 && stride % 4 == 0, "raw stride must be a multiple of 4 and at least 12");
  LX_REQUIRE(n_scan_rings >= 1 && n_scan_rings <= RawBinner::MAX_RINGS, "n_scan_rings must be in [1, 256]");
  LX_REQUIRE(upper_deg != lower_deg, "vertical bounds must differ");
  check_params_();
  LX_HIP(hipSetDevice(device_));
  h_cloud_.reserve(count + 1);
  for (uint32_t i = 0; i < count; i++) {
    const float* p = (const float*)((const char*)raw_xyz + (size_t)i * stride);
    h_cloud_.p[i] = make_float4(p[0], p[1], p[2], 0.f);
  }
  raw_.reserve(count + 1);
  cloud_.reserve(count + 1);
  raw_ring_cnt_.reserve(RawBinner::MAX_RINGS);
  h_raw_ring_cnt_.reserve(RawBinner::MAX_RINGS);
  if (count) LX_HIP(hipMemcpyAsync(raw_.p, h_cloud_.p, sizeof(float4) * count, hipMemcpyHostToDevice, st_));
  MapperParams M;
  M.lower = lower_deg; M.upper = upper_deg; M.n_rings = n_scan_rings;
  M.factor = (float)((int)n_scan_rings - 1) / (upper_deg - lower_deg);   // (nScanRings - 1) / (upperBound - lowerBound), :41-50
  binner_.init(st_);
  // IMU de-skew (projectPointToStartOfSweep, :231): the state the PREVIOUS reset() left behind — the reference projects the
  // points of a sweep before processScanlines resets the state with this sweep's scan time
  ImuTable I;
  const uint32_t H = imu_.size();
  if (H) {
    h_imu_d_.reserve(2 * (size_t)H);
    h_imu_f_.reserve(9 * (size_t)H);
    imu_.fill_table(h_imu_d_.p, h_imu_f_.p, I);
    imu_dt_.reserve(2 * (size_t)H);
    imu_state_.reserve(9 * (size_t)H);
    imu_last_.reserve(1);
    h_imu_last_.reserve(1);
    LX_HIP(hipMemcpyAsync(imu_dt_.p, h_imu_d_.p, sizeof(double) * 2 * H, hipMemcpyHostToDevice, st_));
    LX_HIP(hipMemcpyAsync(imu_state_.p, h_imu_f_.p, sizeof(float) * 9 * H, hipMemcpyHostToDevice, st_));
    LX_HIP(hipMemsetAsync(imu_last_.p, 0, sizeof(ImuLast), st_));
    I.dt = imu_dt_.p;
    I.dstamp = imu_dt_.p + H;
    I.state = imu_state_.p;
  }
  binner_.run(raw_.p, count, M, params.scan_period, cloud_.p, raw_ring_cnt_.p, H ? &I : nullptr, H ? imu_last_.p : nullptr);
  LX_HIP(hipMemcpyAsync(h_raw_ring_cnt_.p, raw_ring_cnt_.p, sizeof(uint32_t) * n_scan_rings, hipMemcpyDeviceToHost, st_));
  if (H) LX_HIP(hipMemcpyAsync(h_imu_last_.p, imu_last_.p, sizeof(ImuLast), hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
  if (H && h_imu_last_.p->valid) imu_.apply_last(*h_imu_last_.p);
  begin_sweep();   // processScanlines: reset(scanTime) ... updateIMUTransform()
  const uint32_t* rs[1] = {h_raw_ring_cnt_.p};
  layout_(1, rs, &n_scan_rings);
  allocate_();   // (cloud_ already holds the binned sweep; reserve() keeps the contents when the capacity suffices)
  LX_HIP(hipStreamSynchronize(st_));
}

// ---- IMU state machine (host): the reference keeps it in BasicScanRegistration; the per-point part runs in ingest.hip
void ImuTracker::update(double stamp, float roll, float pitch, float yaw, const float acc_in[3]) {
  State st;
  st.stamp = stamp; st.roll = HAngle(roll); st.pitch = HAngle(pitch); st.yaw = HAngle(yaw);
  st.acceleration = {acc_in[0], acc_in[1], acc_in[2]};
  if (!hist_.empty()) {   // accumulate IMU position and velocity over time (:84-95)
    HVec3 acc = st.acceleration;
    h_rot_zxy(acc, st.roll, st.pitch, st.yaw);
    const State& prev = hist_.back();
    const float dt = (float)(stamp - prev.stamp);
    st.position = {prev.position.x + prev.velocity.x * dt + 0.5f * acc.x * dt * dt, prev.position.y + prev.velocity.y * dt + 0.5f * acc.y * dt * dt,
                   prev.position.z + prev.velocity.z * dt + 0.5f * acc.z * dt * dt};
    st.velocity = {prev.velocity.x + acc.x * dt, prev.velocity.y + acc.y * dt, prev.velocity.z + acc.z * dt};
  }
  if (hist_.size() >= (size_t)std::max(history_size, 1)) hist_.pop_front();   // CircularBuffer::push, CircularBuffer.h:111-119
  hist_.push_back(st);
}

void ImuTracker::interpolate_for_(float rel, State& out) {   // interpolateIMUStateFor :133-147
  double td = (scan_time_ - hist_[idx_].stamp) + rel;
  while (idx_ < hist_.size() - 1 && td > 0) {
    idx_++;
    td = (scan_time_ - hist_[idx_].stamp) + rel;
  }
  if (idx_ == 0 || td > 0) {
    out = hist_[idx_];
  } else {
    const State &a = hist_[idx_], &b = hist_[idx_ - 1];   // IMUState::interpolate(a, b, ratio), .h:107-131
    const float ratio = (float)(-td / (a.stamp - b.stamp)), inv = 1 - ratio;
    out.roll = HAngle(a.roll.r * inv + b.roll.r * ratio);
    out.pitch = HAngle(a.pitch.r * inv + b.pitch.r * ratio);
    if (a.yaw.r - b.yaw.r > M_PI) out.yaw = HAngle((float)(a.yaw.r * inv + (b.yaw.r + 2 * M_PI) * ratio));
    else if (a.yaw.r - b.yaw.r < -M_PI) out.yaw = HAngle((float)(a.yaw.r * inv + (b.yaw.r - 2 * M_PI) * ratio));
    else out.yaw = HAngle(a.yaw.r * inv + b.yaw.r * ratio);
    out.velocity = {a.velocity.x * inv + b.velocity.x * ratio, a.velocity.y * inv + b.velocity.y * ratio, a.velocity.z * inv + b.velocity.z * ratio};
    out.position = {a.position.x * inv + b.position.x * ratio, a.position.y * inv + b.position.y * ratio, a.position.z * inv + b.position.z * ratio};
  }
}

// reset(scanTime) (:55-79) followed by updateIMUTransform() (:258-281): the latter only needs _imuStart (set by the reset) and
// _imuCur / _imuPositionShift (left behind by the projection loop), so both are done up front
void ImuTracker::begin_sweep() {
  scan_time_ = next_scan_time_;
  idx_ = 0;
  if (!hist_.empty()) interpolate_for_(0.f, start_);
  sweep_start_ = scan_time_;
  imu_trans_[0] = start_.pitch.r; imu_trans_[1] = start_.yaw.r; imu_trans_[2] = start_.roll.r;
  imu_trans_[3] = cur_.pitch.r; imu_trans_[4] = cur_.yaw.r; imu_trans_[5] = cur_.roll.r;
  HVec3 sh = shift_;
  h_rot_yxz(sh, -start_.yaw, -start_.pitch, -start_.roll);
  imu_trans_[6] = sh.x; imu_trans_[7] = sh.y; imu_trans_[8] = sh.z;
  HVec3 v{cur_.velocity.x - start_.velocity.x, cur_.velocity.y - start_.velocity.y, cur_.velocity.z - start_.velocity.z};
  h_rot_yxz(v, -start_.yaw, -start_.pitch, -start_.roll);
  imu_trans_[9] = v.x; imu_trans_[10] = v.y; imu_trans_[11] = v.z;
}

void ImuTracker::fill_table(double* h_d, float* h_f, ImuTable& I) const {
  // projectPointToStartOfSweep (:231) uses the state the PREVIOUS reset() left behind — the reference projects the points of a
  // sweep before processScanlines resets the state with this sweep's scan time
  const uint32_t H = (uint32_t)hist_.size();
  for (uint32_t j = 0; j < H; j++) {
    const State& s = hist_[j];
    h_d[j] = scan_time_ - s.stamp;
    h_d[H + j] = j ? s.stamp - hist_[j - 1].stamp : 1.0;
    float* o = h_f + 9 * (size_t)j;
    o[0] = s.roll.r; o[1] = s.pitch.r; o[2] = s.yaw.r;
    o[3] = s.position.x; o[4] = s.position.y; o[5] = s.position.z;
    o[6] = s.velocity.x; o[7] = s.velocity.y; o[8] = s.velocity.z;
  }
  I.H = H;
  I.idx0 = H ? (uint32_t)std::min(idx_, (size_t)H - 1) : 0u;
  const HAngle* sa[3] = {&start_.roll, &start_.pitch, &start_.yaw};
  for (int k = 0; k < 3; k++) { I.start_c[k] = sa[k]->c; I.start_s[k] = sa[k]->s; }
  I.start_pos[0] = start_.position.x; I.start_pos[1] = start_.position.y; I.start_pos[2] = start_.position.z;
  I.start_vel[0] = start_.velocity.x; I.start_vel[1] = start_.velocity.y; I.start_vel[2] = start_.velocity.z;
  I.rel_sweep_base = scan_time_ - sweep_start_;
}

void ImuTracker::apply_last(const ImuLast& L) {
  cur_.roll = HAngle(L.roll); cur_.pitch = HAngle(L.pitch); cur_.yaw = HAngle(L.yaw);
  cur_.position = {L.pos[0], L.pos[1], L.pos[2]};
  cur_.velocity = {L.vel[0], L.vel[1], L.vel[2]};
  shift_ = {L.shift[0], L.shift[1], L.shift[2]};
  idx_ = L.idx;
}

// the (binned) input cloud of a sweep and its ring sizes
int FeatureExtractor::download_cloud(uint32_t sweep, loamx_cloud* full, uint32_t* ring_size_out) {
  LX_REQUIRE(sweep < nsw_, "sweep index out of range");
  const uint32_t r0 = h_ring_base_[sweep], r1 = h_ring_base_[sweep + 1];
  if (ring_size_out)
    for (uint32_t r = r0; r < r1; r++) ring_size_out[r - r0] = h_ring_off_[r + 1] - h_ring_off_[r];
  if (!full) return LOAMX_OK;
  check_cloud(full, false);
  const uint32_t a = h_pt_base_[sweep], b = h_pt_base_[sweep + 1];
  h_pack_.reserve((size_t)(b - a) + 2);   // (pinned: a copy into pageable memory is staged by the runtime)
  if (b > a) LX_HIP(hipMemcpyAsync(h_pack_.p, cloud_.p + a, sizeof(float4) * (b - a), hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
  return unpack_cloud(h_pack_.p, b - a, full);
}

void FeatureExtractor::run_async(bool mirror_offsets) {
  TraceRange trace_range("loamx:features:extract");
  LX_REQUIRE(nsw_ > 0, "run() before upload()");
  LX_HIP(hipSetDevice(device_));
  // mirror_offsets (the linked single-sweep chain): the compaction kernels write their offset tables into pinned host memory as well,
  // and an event is recorded behind each — the consumer reads the sharp / less-sharp / flat sizes ~50 us before the less-flat voxel grid
  // has finished (device_results_front / device_results_lf) and starts the odometry's iterations, which do not need that cloud
  split_ = mirror_offsets && max_ring_len_ <= LFV_MAX;
  uint32_t* hmir = nullptr;
  if (split_) {
    h_mirror_off_.reserve(n_offsets());
    hmir = h_mirror_off_.p;
    if (!ev_front_) { LX_HIP(hipEventCreateWithFlags(&ev_front_, hipEventDisableTiming)); LX_HIP(hipEventCreateWithFlags(&ev_lf_, hipEventDisableTiming)); }
  }
  const int cr = params.curv_region;
  (void)cr;   // (lf_valid_ is cleared ring by ring in k_feat_ring's prologue: no memset)
  const uint32_t caps[3] = {(uint32_t)(params.max_sharp * params.n_regions), (uint32_t)(params.max_less_sharp * params.n_regions),
                            (uint32_t)(params.max_flat * params.n_regions)};
  const uint32_t flag_bytes = (max_ring_len_ + 15u) & ~15u;
  const uint32_t nmax = (max_ring_len_ / (uint32_t)params.n_regions + 8u + 15u) & ~15u;
  uint32_t sortP = 64;   // bitonic sort size of one region
  while (sortP < nmax) sortP <<= 1;
  const size_t lds = ((8 * (size_t)flag_bytes + 4 * (size_t)(caps[0] + caps[1] + caps[2]) + 15) & ~(size_t)15) + (size_t)FEAT_WAVES * nmax * (4 + 4 + 1) +
                     (sortP > 512 ? (size_t)FEAT_WAVES * sortP * 8 : 0) + 16;   // (regions of up to 512 points are sorted in registers)
  LX_REQUIRE(lds <= 160 * 1024, "scan ring too long for the LDS staging of k_feat_ring");
  static const bool force_seq = diag_env("LOAMX_FEAT_SEQUENTIAL") && atoi(diag_env("LOAMX_FEAT_SEQUENTIAL")) != 0;   // (diagnostic: the regions one after the other)
  if (lds > 64 * 1024)
    LX_HIP(hipFuncSetAttribute((const void*)k_feat_ring, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_feat_ring, dim3(nring_), dim3(64 * FEAT_WAVES), lds, st_, cloud_.p, ring_off_.p, ring_sweep_base_.p, params, flag_bytes, nmax,
                     sortP, slots_[0].p, slots_[1].p, slots_[2].p, slot_cnt_[0].p, slot_cnt_[1].p, slot_cnt_[2].p, lf_valid_.p, force_seq ? 1 : 0,
                     h_bad_.p);
  hipLaunchKernelGGL(k_feat_compact, dim3(nring_, 3), dim3(64), 0, st_, slots_[0].p, slots_[1].p, slots_[2].p, slot_cnt_[0].p, slot_cnt_[1].p,
                     slot_cnt_[2].p, caps[0], caps[1], caps[2], out_[0].p, out_[1].p, out_[2].p, offs_.p, sweep_ring_base_.p, nsw_, hmir);
  if (split_) LX_HIP(hipEventRecord(ev_front_, st_));
  // per-ring voxel grid of the less-flat candidates
  const float inv = 1.0f / params.less_flat_leaf;
  if (max_ring_len_ <= LFV_MAX) {
    uint32_t P = 2;
    while (P < max_ring_len_) P <<= 1;
    if ((size_t)P * 24 > 64 * 1024)
      LX_HIP(hipFuncSetAttribute((const void*)k_feat_lf_voxel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)P * 24)));
    hipLaunchKernelGGL(k_feat_lf_voxel, dim3(nring_), dim3(LFV_THREADS), (size_t)P * 24, st_, cloud_.p, ring_off_.p, lf_valid_.p, inv, P,
                       lf_slots_.p, lf_cnt_.p);
    hipLaunchKernelGGL(k_feat_lf_compact, dim3(nring_), dim3(256), 0, st_, lf_slots_.p, ring_off_.p, lf_cnt_.p, lf_off_(), lf_out_.p,
                       hmir ? hmir + (size_t)3 * off_stride_ : nullptr);
    if (split_) LX_HIP(hipEventRecord(ev_lf_, st_));
  } else {   // very long rings: generic segmented pipeline (global radix sort)
    vox_.compute_ijk(cloud_.p, lf_valid_.p, n_, ring_off_.p, nring_, inv, inv);
    vox_.sort_reduce(cloud_.p, lf_valid_.p, n_, ring_off_.p, nring_, lf_out_.p, lf_off_());
  }
#ifdef LOAMX_PROF_FEAT
  {
    unsigned long long ts[8];
    LX_HIP(hipStreamSynchronize(st_));
    LX_HIP(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_feat_ts), sizeof(ts)));
    fprintf(stderr, "[feat_ring ts, 10ns ticks] load %lld  rank-sort %lld  picks %lld  tail %lld\n", (long long)(ts[1] - ts[0]), (long long)(ts[2] - ts[1]),
            (long long)(ts[3] - ts[2]), (long long)(ts[4] - ts[3]));
  }
#endif
  LX_HIP(hipGetLastError());
}

void FeatureExtractor::sync() {
  LX_HIP(hipStreamSynchronize(st_));
  vox_.check();   // a timed-out wait inside the long-ring fallback's voxel kernel raises instead of passing garbage on
  check_finite_input();
}

// The four feature clouds of one sweep, written back to back into pinned (device-mapped) memory by ONE launch, sizes in a header in
// front: the host needs one wait and no copy command (four offset copies + four cloud copies, each with its own wait, were ~0.2 ms
// of a 1.8 ms VLP-16 sweep through the single-stream entry points).  dst[0] = header (the four counts), dst[1 ...] = the points.
__global__ __launch_bounds__(256) void k_feat_pack_host(const float4* __restrict__ o0, const float4* __restrict__ o1, const float4* __restrict__ o2,
                                                        const float4* __restrict__ lf, const uint32_t* __restrict__ off0,
                                                        const uint32_t* __restrict__ off1, const uint32_t* __restrict__ off2,
                                                        const uint32_t* __restrict__ lf_off, uint32_t sweep, uint32_t ring_a, uint32_t ring_b,
                                                        uint32_t capacity, float4* __restrict__ dst) {
  const uint32_t a0 = off0[sweep], n0 = off0[sweep + 1] - a0, a1 = off1[sweep], n1 = off1[sweep + 1] - a1;
  const uint32_t a2 = off2[sweep], n2 = off2[sweep + 1] - a2, a3 = lf_off[ring_a], n3 = lf_off[ring_b] - a3;
  const uint32_t total = n0 + n1 + n2 + n3;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    uint4 h = make_uint4(n0, n1, n2, n3);
    if (total > capacity) h = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);   // (cannot happen: every cloud is a subset of the sweep)
    *reinterpret_cast<uint4*>(dst) = h;
  }
  if (total > capacity) return;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float4 v;
    if (i < n0) v = o0[a0 + i];
    else if (i < n0 + n1) v = o1[a1 + (i - n0)];
    else if (i < n0 + n1 + n2) v = o2[a2 + (i - n0 - n1)];
    else v = lf[a3 + (i - n0 - n1 - n2)];
    dst[1 + i] = v;
  }
}

void FeatureExtractor::device_results(uint32_t sweep, const float4* ptr[4], uint32_t count[4]) {
  LX_REQUIRE(sweep < nsw_, "sweep index out of range");
  h_link_off_.reserve(n_offsets());
  LX_HIP(hipMemcpyAsync(h_link_off_.p, offs_.p, sizeof(uint32_t) * n_offsets(), hipMemcpyDeviceToHost, st_));
  LX_HIP(hipStreamSynchronize(st_));
  vox_.check();
  check_finite_input();
  const uint32_t* o = h_link_off_.p;
  for (int k = 0; k < 3; k++) {
    const uint32_t a = o[(size_t)k * off_stride_ + sweep], b = o[(size_t)k * off_stride_ + sweep + 1];
    ptr[k] = out_[k].p + a;
    count[k] = b - a;
  }
  const uint32_t* lf = o + (size_t)3 * off_stride_;
  const uint32_t a = lf[h_ring_base_[sweep]], b = lf[h_ring_base_[sweep + 1]];
  ptr[3] = lf_out_.p + a;
  count[3] = b - a;
}

// The two halves of device_results() for a run_async(true): the sharp / less-sharp / flat clouds as soon as k_feat_compact has finished
// (input check included: k_feat_ring raises the non-finite mark) ...
void FeatureExtractor::device_results_front(uint32_t sweep, const float4* ptr[3], uint32_t count[3]) {
  LX_REQUIRE(sweep < nsw_, "sweep index out of range");
  LX_REQUIRE(split_, "internal: device_results_front() without run_async(true)");
  spin_event(ev_front_);
  check_finite_input();
  const uint32_t* o = h_mirror_off_.p;
  for (int k = 0; k < 3; k++) {
    const uint32_t a = o[(size_t)k * off_stride_ + sweep], b = o[(size_t)k * off_stride_ + sweep + 1];
    ptr[k] = out_[k].p + a;
    count[k] = b - a;
  }
}
// ... and the less-flat cloud once its per-ring voxel grid has been compacted
void FeatureExtractor::device_results_lf(uint32_t sweep, const float4*& ptr, uint32_t& count) {
  LX_REQUIRE(sweep < nsw_, "sweep index out of range");
  LX_REQUIRE(split_, "internal: device_results_lf() without run_async(true)");
  spin_event(ev_lf_);
  const uint32_t* lf = h_mirror_off_.p + (size_t)3 * off_stride_;
  const uint32_t a = lf[h_ring_base_[sweep]], b = lf[h_ring_base_[sweep + 1]];
  ptr = lf_out_.p + a;
  count = b - a;
}

int FeatureExtractor::download(uint32_t sweep, loamx_cloud* sharp, loamx_cloud* less_sharp, loamx_cloud* flat, loamx_cloud* less_flat) {
  LX_REQUIRE(sweep < nsw_, "sweep index out of range");
  loamx_cloud* outs[4] = {sharp, less_sharp, flat, less_flat};
  for (auto* o : outs) if (o) check_cloud(o, false);
  const uint32_t n_pts = h_pt_base_[sweep + 1] - h_pt_base_[sweep];
  const uint32_t capacity = 4 * n_pts;   // each of the four clouds is a subset of the sweep's points
  h_pack_.reserve((size_t)capacity + 2);
  hipLaunchKernelGGL(k_feat_pack_host, dim3(std::min<uint32_t>((capacity + 1023) / 1024 + 1, 128u)), dim3(256), 0, st_, out_[0].p, out_[1].p, out_[2].p,
                     lf_out_.p, d_feat_off(0), d_feat_off(1), d_feat_off(2), lf_off_(), sweep, h_ring_base_[sweep], h_ring_base_[sweep + 1], capacity,
                     h_pack_.p);
  LX_HIP(hipStreamSynchronize(st_));
  vox_.check();
  check_finite_input();
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(h_pack_.p);
  LX_REQUIRE(hdr[0] != 0xffffffffu, "internal: the feature clouds of a sweep exceed four times its points");
  int rc = LOAMX_OK;
  const float4* src = h_pack_.p + 1;
  for (int k = 0; k < 4; k++) {
    if (outs[k]) {
      const int r = unpack_cloud(src, hdr[k], outs[k]);
      if (r != LOAMX_OK) rc = r;
    }
    src += hdr[k];
  }
  return rc;
}

}  // namespace loamx
