// Micro-benchmark (MI355X): what does a 16 MiB device -> pinned-host copy cost the REST of the device while it runs?
//   hipcc --offload-arch=gfx950 -O3 d2h.hip -o d2h
// Measures, for each way of doing the copy (hipMemcpyAsync alone; hipMemcpyAsync while a host -> device copy is in flight on another
// stream, which is when the runtime has been seen to fall back to its 256-workgroup blit kernel; an own copy kernel of G workgroups):
//   * the copy's throughput,
//   * the round trip of a one-word device -> host write from another stream while the copy runs (launch, write, host sees it),
//   * the duration of a small memory-bound kernel (scattered atomics over 4 MiB) on another stream while the copy runs.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_word(volatile unsigned* host_word, unsigned v) { *host_word = v; }
__global__ void k_scatter(unsigned* c, unsigned nbins) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; atomicAdd(c + (i * 2654435761u) % nbins, 1u); }
// G workgroups of 256 threads, U float4 per thread in flight
template <int U>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = src[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; k++) dst[i + k * stride] = v[k];
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;   // run one mode only (for a profiler run per mode)
  const size_t bytes = 16u << 20, n4 = bytes / 16;
  float4 *d_src, *d_dst, *h_out, *h_in;
  unsigned *d_bins, *h_word;
  CK(hipMalloc(&d_src, bytes)); CK(hipMalloc(&d_dst, bytes)); CK(hipMalloc(&d_bins, 4u << 20));
  CK(hipHostMalloc(&h_out, bytes)); CK(hipHostMalloc(&h_in, bytes)); CK(hipHostMalloc(&h_word, 64));
  CK(hipMemset(d_src, 1, bytes)); CK(hipMemset(d_bins, 0, 4u << 20));
  memset(h_in, 2, bytes);
  hipStream_t s_copy, s_h2d, s_probe, s_prod, s_prio;
  CK(hipStreamCreateWithFlags(&s_prod, hipStreamNonBlocking));
  { int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&s_prio, hipStreamNonBlocking, (lo + hi) / 2)); }
  hipEvent_t e_prod; CK(hipEventCreateWithFlags(&e_prod, hipEventDisableTiming));
  CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_h2d, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_probe, hipStreamNonBlocking));
  hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));

  struct Mode { const char* name; int kind; int G; bool h2d; };
  std::vector<Mode> modes = {{"idle (no copy)", 0, 0, false}, {"hipMemcpyAsync D2H", 1, 0, false}, {"hipMemcpyAsync D2H + H2D in flight", 1, 0, true},
                             {"own kernel  8 WG x4", 2, 8, true}, {"own kernel 16 WG x4", 2, 16, true}, {"own kernel 32 WG x4", 2, 32, true},
                             {"own kernel 64 WG x4", 2, 64, true}, {"own kernel 256 WG x4", 2, 256, true}, {"own kernel 16 WG x4, no H2D", 2, 16, false},
                             {"hipMemcpyAsync D2H behind hipStreamWaitEvent (+H2D)", 3, 0, true}, {"hipMemcpyAsync D2H on a priority stream (+H2D)", 4, 0, true},
                             {"hipMemcpyAsync D2H after hipEventSynchronize (+H2D)", 5, 0, true}};
  for (size_t mi = 0; mi < modes.size(); mi++) {
    const Mode& m = modes[mi];
    if (only >= 0 && (int)mi != only) continue;
    std::atomic<bool> stop{false};
    std::atomic<long> copies{0};
    double t_begin = now_us();
    std::thread bg([&] {   // keeps copies going back to back
      (void)hipSetDevice(0);
      while (!stop.load()) {
        if (m.h2d) (void)hipMemcpyAsync(d_dst, h_in, bytes, hipMemcpyHostToDevice, s_h2d);
        if (m.kind == 1) (void)hipMemcpyAsync(h_out, d_src, bytes, hipMemcpyDeviceToHost, s_copy);
        else if (m.kind == 2) hipLaunchKernelGGL(k_copy<4>, dim3(m.G), dim3(256), 0, s_copy, d_src, h_out, n4);
        else if (m.kind == 3 || m.kind == 5) {   // the copy depends on a kernel of another stream
          hipLaunchKernelGGL(k_scatter, dim3(64), dim3(256), 0, s_prod, d_bins, 1u << 20);
          (void)hipEventRecord(e_prod, s_prod);
          if (m.kind == 3) (void)hipStreamWaitEvent(s_copy, e_prod, 0); else (void)hipEventSynchronize(e_prod);
          (void)hipMemcpyAsync(h_out, d_src, bytes, hipMemcpyDeviceToHost, s_copy);
        } else if (m.kind == 4) { (void)hipMemcpyAsync(h_out, d_src, bytes, hipMemcpyDeviceToHost, s_prio); (void)hipStreamSynchronize(s_prio); }
        if (m.kind) (void)hipStreamSynchronize(s_copy);
        if (m.h2d) (void)hipStreamSynchronize(s_h2d);
        if (!m.kind && !m.h2d) std::this_thread::sleep_for(std::chrono::microseconds(200));
        copies++;
      }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    const long c0 = copies.load(); const double t0 = now_us();
    std::vector<double> rt, kd;
    for (int r = 0; r < 300; r++) {
      *(volatile unsigned*)h_word = 0;
      const double a = now_us();
      hipLaunchKernelGGL(k_word, dim3(1), dim3(1), 0, s_probe, h_word, (unsigned)r + 1);
      while (*(volatile unsigned*)h_word != (unsigned)r + 1) {}
      rt.push_back(now_us() - a);
      CK(hipEventRecord(ea, s_probe));
      hipLaunchKernelGGL(k_scatter, dim3(1024), dim3(256), 0, s_probe, d_bins, 1u << 20);
      CK(hipEventRecord(eb, s_probe));
      CK(hipEventSynchronize(eb));
      float ms; CK(hipEventElapsedTime(&ms, ea, eb));
      kd.push_back(ms * 1e3);
    }
    const long c1 = copies.load(); const double t1 = now_us();
    stop = true; bg.join();
    std::sort(rt.begin(), rt.end()); std::sort(kd.begin(), kd.end());
    const double per_copy_us = (c1 > c0) ? (t1 - t0) / (double)(c1 - c0) : 0.0;
    printf("%-52s copy %7.1f us (%5.1f GB/s)   word round trip median %6.1f p95 %6.1f us   scatter kernel median %6.1f p95 %6.1f us\n", m.name, per_copy_us,
           per_copy_us > 0 && m.kind ? bytes / per_copy_us / 1e3 : 0.0, rt[rt.size() / 2], rt[rt.size() * 95 / 100], kd[kd.size() / 2], kd[kd.size() * 95 / 100]);
    (void)t_begin;
  }
  return 0;
}
