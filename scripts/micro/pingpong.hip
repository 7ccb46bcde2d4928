// Micro-benchmark (MI355X): round trip of a flag between two workgroups — on the SAME XCD (workgroup ids 0 and 8 of one dispatch: ids are
// dealt round-robin over the 8 XCDs) and on DIFFERENT XCDs (ids 0 and 1) — for the store / load flavours an exchange could use, alone and
// beside a kernel that streams through HBM.  Every wait gives up after ~20 ms (a flavour that never sees the partner's store reports
// "timeout" instead of hanging the device).  hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long wclk() { return wall_clock64(); }   // 100 MHz
template <int S> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
  if (S == 0) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (S == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if (S == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (S == 3) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_atomic_swap %0, %1, off" ::"v"(p), "v"(v) : "memory");   // (no return: executed in the XCD's L2)
}
template <int L> __device__ __forceinline__ unsigned ld(const unsigned* p) {
  unsigned v;
  if (L == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (L == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (L == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (L == 3) asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (L == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (L == 5) { const unsigned z = 0u; asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory"); }   // returning atomic in the local L2
  else { const unsigned z = 0u; asm volatile("global_atomic_or %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory"); }       // ... at agent scope
  return v;
}
// out[0] = ticks of block `a`'s loop, out[1] = timeouts, out[2] / out[3] = XCC ids of the two blocks
template <int S, int L> __global__ void k_pp(unsigned* A, unsigned* B, int rounds, unsigned base, unsigned a, unsigned b, unsigned long long* out) {
  if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
  const bool first = blockIdx.x == a;
  out[first ? 2 : 3] = xcc;
  unsigned* mine = first ? A : B;
  const unsigned* theirs = first ? B : A;
  const unsigned long long t0 = wclk();
  unsigned long long fails = 0;
  for (int r = 1; r <= rounds; r++) {
    const unsigned v = base + (unsigned)r;
    if (first) st<S>(mine, v);
    const unsigned long long tw = wclk();
    while (ld<L>(theirs) != v) {
      if (wclk() - tw > 2000000ull) { fails++; break; }   // 20 ms
    }
    if (fails) break;
    if (!first) st<S>(mine, v);
  }
  if (first) { out[0] = wclk() - t0; out[1] = fails; } else if (fails) out[1] = fails;
}
__global__ void k_stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; r++)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}


struct Ctx { hipStream_t s1, s2; unsigned *A, *B; unsigned long long *out; float4 *src, *dst; size_t n; unsigned base; };
template <int S, int L> int run(const char* name, Ctx& c, unsigned partner, const char* where, bool load) {
  const int rounds = load ? 400 : 2000;
  unsigned long long h[8];
  CK(hipMemsetAsync(c.out, 0, 64, c.s1));
  CK(hipStreamSynchronize(c.s1));
  if (load) {   // ~8 ms of streaming at full bandwidth, started first
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, c.s2, c.src, c.dst, c.n, 8);
    hipLaunchKernelGGL((k_pp<0, 0>), dim3(16), dim3(64), 0, c.s1, c.A, c.B, 50, c.base, 0u, partner, c.out);   // (lets the stream kernel get going)
    c.base += 57u;
  }
  hipLaunchKernelGGL((k_pp<S, L>), dim3(16), dim3(64), 0, c.s1, c.A, c.B, rounds, c.base, 0u, partner, c.out);
  CK(hipStreamSynchronize(c.s1));
  const bool still = load && hipStreamQuery(c.s2) == hipErrorNotReady;
  CK(hipStreamSynchronize(c.s2));
  CK(hipMemcpy(h, c.out, 64, hipMemcpyDeviceToHost));
  c.base += (unsigned)rounds + 7u;
  if (h[1]) printf("  %-40s %-10s TIMEOUT (xcc %llu / %llu)\n", name, where, h[2], h[3]);
  else printf("  %-40s %-10s round trip %6.0f ns (xcc %llu / %llu)%s\n", name, where, h[0] * 10.0 / rounds, h[2], h[3], load && !still ? "  [stream kernel had ended]" : "");
  return 0;
}

int main() {
  Ctx c;
  CK(hipStreamCreateWithFlags(&c.s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.s2, hipStreamNonBlocking));
  char* blk;
  CK(hipMalloc(&blk, 4096));
  CK(hipMemset(blk, 0, 4096));
  c.A = (unsigned*)blk; c.B = (unsigned*)(blk + 1024);
  CK(hipMalloc(&c.out, 64));
  c.n = (size_t)64 << 20;   // 1 GiB each
  CK(hipMalloc(&c.src, c.n * sizeof(float4)));
  CK(hipMalloc(&c.dst, c.n * sizeof(float4)));
  CK(hipMemset(c.src, 1, c.n * sizeof(float4)));
  c.base = 1000;
  for (int load = 0; load < 2; load++) {
    printf(load ? "beside a kernel streaming through HBM at full bandwidth:\n" : "alone on the device:\n");
    for (int same = 1; same >= 0; same--) {
      const unsigned partner = same ? 8u : 1u;
      const char* where = same ? "same XCD" : "other XCD";
      if (run<0, 0>("store sc1, load sc1 (today)", c, partner, where, load)) return 1;
      if (run<2, 2>("store sc0 sc1, load sc0 sc1", c, partner, where, load)) return 1;
      if (run<0, 2>("store sc1, load sc0 sc1", c, partner, where, load)) return 1;
      if (run<2, 0>("store sc0 sc1, load sc1", c, partner, where, load)) return 1;
      if (run<0, 4>("store sc1, load nt", c, partner, where, load)) return 1;
      if (run<0, 6>("store sc1, atomic_or sc0 sc1 (agent)", c, partner, where, load)) return 1;
      if (same) {   // (flavours that can only work inside one XCD's L2)
        if (run<1, 5>("store plain, atomic_or sc0 (local L2)", c, partner, where, load)) return 1;
        if (run<0, 5>("store sc1, atomic_or sc0 (local L2)", c, partner, where, load)) return 1;
        if (run<4, 5>("atomic_swap (L2), atomic_or sc0 (L2)", c, partner, where, load)) return 1;
      }
    }
  }
  return 0;
}
