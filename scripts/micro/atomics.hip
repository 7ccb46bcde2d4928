// Micro-benchmarks (MI355X): what do device-scope atomics and kernel boundaries cost?  hipcc --offload-arch=gfx950 -O3 atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_same_ret(unsigned* c, unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = atomicAdd(c, 1u); }
__global__ void k_same_noret(unsigned* c) { if (threadIdx.x == 0) atomicAdd(c, 1u); }
__global__ void k_distinct_ret(unsigned* c, unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = atomicAdd(c + 64 * blockIdx.x, 1u); }
__global__ void k_wave_same_ret(unsigned* c, unsigned* out) { out[blockIdx.x * blockDim.x + threadIdx.x] = atomicAdd(c + (threadIdx.x >> 6), 1u); }   // every lane, one address per wave
__global__ void k_scatter_noret(unsigned* c, unsigned nbins) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; atomicAdd(c + (i * 2654435761u) % nbins, 1u); }
__global__ void k_empty() {}
__global__ void k_load_chain(const unsigned* p, unsigned* out, int n) {   // dependent loads: latency of one global load (L2 hit)
  unsigned i = threadIdx.x;
  for (int k = 0; k < n; k++) i = p[i];
  out[threadIdx.x] = i;
}
__global__ void k_store16(unsigned short* p, int per) { unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * per; for (int k = 0; k < per; k++) p[i + k] = (unsigned short)k; }

template <class F> float timeit(hipStream_t st, int reps, F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipStreamSynchronize(st);
  std::vector<float> t;
  for (int r = 0; r < reps; r++) { hipEventRecord(a, st); f(); hipEventRecord(b, st); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms * 1000.f); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned *c, *out; unsigned short* s16;
  CK(hipMalloc(&c, 64 << 20)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&s16, 64 << 20));
  CK(hipMemset(c, 0, 64 << 20));
  const float base = timeit(st, 20, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); });
  printf("event-bracketed empty kernel: %.1f us\n", base);
  printf("10 empty kernels back to back: %.1f us\n", timeit(st, 20, [&] { for (int k = 0; k < 10; k++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }));
  printf("10 x 2000-block empty kernels: %.1f us\n", timeit(st, 20, [&] { for (int k = 0; k < 10; k++) hipLaunchKernelGGL(k_empty, dim3(2000), dim3(256), 0, st); }));
  for (int n : {16, 150, 300, 2000, 20000}) {
    printf("blocks %6d: same addr + return %7.1f us | same addr no return %7.1f | distinct addr + return %7.1f\n", n,
           timeit(st, 10, [&] { hipLaunchKernelGGL(k_same_ret, dim3(n), dim3(256), 0, st, c, out); }),
           timeit(st, 10, [&] { hipLaunchKernelGGL(k_same_noret, dim3(n), dim3(256), 0, st, c); }),
           timeit(st, 10, [&] { hipLaunchKernelGGL(k_distinct_ret, dim3(n), dim3(256), 0, st, c, out); }));
  }
  printf("1200 blocks x 256 lanes, every lane atomicAdd+return on its wave's address (4 addresses): %.1f us\n",
         timeit(st, 10, [&] { hipLaunchKernelGGL(k_wave_same_ret, dim3(1200), dim3(256), 0, st, c, out); }));
  for (unsigned nb : {256u, 32768u, 524288u})
    printf("300k scattered atomics without return over %7u bins: %.1f us\n", nb, timeit(st, 10, [&] { hipLaunchKernelGGL(k_scatter_noret, dim3(1172), dim3(256), 0, st, c, nb); }));
  std::vector<unsigned> h(1 << 20);
  for (unsigned i = 0; i < h.size(); i++) h[i] = (i * 40503u + 12345u) & ((1u << 20) - 1);
  CK(hipMemcpy(c, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const float t100 = timeit(st, 10, [&] { hipLaunchKernelGGL(k_load_chain, dim3(1), dim3(64), 0, st, c, out, 100); });
  const float t1100 = timeit(st, 10, [&] { hipLaunchKernelGGL(k_load_chain, dim3(1), dim3(64), 0, st, c, out, 1100); });
  printf("dependent global load (4 MB working set): %.0f ns per load\n", (t1100 - t100));
  printf("16 x 1024 threads x 32 consecutive 2-byte stores each: %.1f us\n", timeit(st, 10, [&] { hipLaunchKernelGGL(k_store16, dim3(16), dim3(1024), 0, st, s16, 32); }));
  return 0;
}
