// Shared host/device definitions of libloamx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/loamx.h"
#ifndef LOAMX_NO_ROCTX   // (make NO_ROCTX=1: a ROCm install without the rocprofiler SDK; the trace ranges become no-ops)
#include <rocprofiler-sdk-roctx/roctx.h>
#endif

namespace loamx {

// ---- error plumbing: exceptions never cross the ABI; each extern "C" entry wraps its body in guard() ------------
void set_last_error(const std::string& s);
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& s) : std::runtime_error(s), code(c) {}
};
// -DLOAMX_API_TRACE (diagnostic build): every runtime call made through LX_HIP and every kernel launch is timed on the host; one that
// takes longer than 300 us — and is not a wait by its name — is reported on stderr with its place.  How round 6 found the copies that
// blocked their callers for milliseconds (profiles/r06_ab.md section 16).
#ifdef LOAMX_API_TRACE
struct ApiTimer {
  const char* what; const char* file; int line;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ApiTimer(const char* w, const char* f, int l) : what(w), file(f), line(l) {}
  ~ApiTimer() {
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (us > 300.0 && !strstr(what, "Synchronize") && !strstr(what, "Malloc") && !strstr(what, "Free") && !strstr(what, "hipGetDeviceProperties") && !strstr(what, "Occupancy"))
      fprintf(stderr, "[api trace] %.0f us in %.80s (%s:%d)\n", us, what, file, line);
  }
};
#define LX_API_TIMER(what) ::loamx::ApiTimer api_timer_(what, __FILE__, __LINE__)
#else
#define LX_API_TIMER(what) do { } while (0)
#endif
#define LX_HIP(expr)                                                                                         \
  do {                                                                                                       \
    LX_API_TIMER(#expr);                                                                                     \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess)                                                                                    \
      throw ::loamx::Error(LOAMX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                                            ":" + std::to_string(__LINE__) + ")");                          \
  } while (0)
#define LX_REQUIRE(cond, msg)                                         \
  do {                                                                \
    if (!(cond)) throw ::loamx::Error(LOAMX_E_INVALID, (msg));        \
  } while (0)

template <class F> int guard(F&& f) {
  try {
    return f();
  } catch (const Error& e) {
    set_last_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return LOAMX_E_INVALID;
  } catch (...) {
    set_last_error("unknown error");
    return LOAMX_E_INVALID;
  }
}

// ---- device buffer with grow-only capacity ------------------------------------------------------------------------
template <class T> struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  // grow (contents discarded unless keep)
  void reserve(size_t n, hipStream_t st = nullptr, bool keep = false) {
    if (n <= cap) return;
    size_t ncap = n + n / 4 + 64;
    T* np = nullptr;
    if (getenv("LOAMX_ALLOC_TRACE")) fprintf(stderr, "[alloc] device buffer %zu -> %zu bytes%s\n", cap * sizeof(T), ncap * sizeof(T), keep ? " (contents kept)" : "");
    LX_HIP(hipMalloc((void**)&np, ncap * sizeof(T)));
    if (keep && p && cap) {
      LX_HIP(hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, st));
      LX_HIP(hipStreamSynchronize(st));
    }
    if (p) (void)hipFree(p);
    p = np;
    cap = ncap;
  }
};

// pinned host staging buffer
template <class T> struct PinBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~PinBuf() {
    if (p) (void)hipHostFree(p);
  }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipHostFree(p);
    cap = n + n / 4 + 64;
    LX_HIP(hipHostMalloc((void**)&p, cap * sizeof(T), hipHostMallocDefault));
  }
};

// ---- packing between caller records and the device float4 layout ------------------------------------------------
inline void check_cloud(const loamx_cloud* c, bool allow_null_data = true) {
  LX_REQUIRE(c != nullptr, "cloud descriptor is NULL");
  LX_REQUIRE(c->stride >= 16 && (c->stride % 4) == 0, "cloud stride must be a multiple of 4 and >= 16");
  LX_REQUIRE(c->intensity_offset >= 12 && c->intensity_offset + 4 <= c->stride && (c->intensity_offset % 4) == 0,
             "cloud intensity_offset must be 4-aligned, >= 12 and inside the record");
  LX_REQUIRE(allow_null_data || c->data != nullptr || c->count == 0, "cloud data is NULL");
}
// true when every x, y, z of n packed points is finite (x * 0 is NaN for NaN and +-Inf, 0 otherwise: a loop the compiler vectorises)
inline bool packed_all_finite(const float4* p, size_t n) {
  float acc = 0.f;
  for (size_t i = 0; i < n; i++) acc += p[i].x * 0.f + p[i].y * 0.f + p[i].z * 0.f;
  return acc == 0.f;
}
// true when [p, p + bytes) is host memory the HIP runtime has pinned (hipHostMalloc, hipHostRegister; torch's pin_memory()): an
// asynchronous copy from / to it is a DMA straight from / to the caller's buffer — no staging copy by this library or by the runtime
inline bool host_pinned(const void* p, size_t bytes) {
  if (!p || !bytes) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (a.type != hipMemoryTypeHost) return false;
  if (hipPointerGetAttributes(&a, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}
inline bool packed_layout(const loamx_cloud* c) { return c->stride == 16 && c->intensity_offset == 12; }
inline void pack_cloud(const loamx_cloud* c, float4* dst) {
  const char* src = (const char*)c->data;
  if (c->stride == 16 && c->intensity_offset == 12) {   // x y z intensity records: the device layout itself
    if (c->count) memcpy(dst, src, (size_t)c->count * 16);
    return;
  }
  char* d = (char*)dst;   // (byte copies: the destination is not always 16-byte aligned storage, and the compiler must not assume it is)
  for (uint32_t i = 0; i < c->count; i++) {
    const char* r = src + (size_t)i * c->stride;
    memcpy(d + (size_t)i * 16, r, 12);
    memcpy(d + (size_t)i * 16 + 12, r + c->intensity_offset, 4);
  }
}
// writes min(n, capacity) points; returns LOAMX_E_CAPACITY (after writing what fits) when n > capacity
inline int unpack_cloud(const float4* src, uint32_t n, loamx_cloud* c) {
  uint32_t cap = c->count;
  uint32_t m = n < cap ? n : cap;
  char* dst = (char*)c->data;
  if (c->stride == 16 && c->intensity_offset == 12) {
    if (m) memcpy(dst, src, (size_t)m * 16);
    c->count = n;
    return n > cap ? LOAMX_E_CAPACITY : LOAMX_OK;
  }
  for (uint32_t i = 0; i < m; i++) {
    float* r = (float*)(dst + (size_t)i * c->stride);
    r[0] = src[i].x; r[1] = src[i].y; r[2] = src[i].z;
    if (c->stride >= 32 && c->intensity_offset == 16) r[3] = 1.0f;   // PCL's homogeneous w
    *(float*)((char*)r + c->intensity_offset) = src[i].w;
  }
  c->count = n;
  return n > cap ? LOAMX_E_CAPACITY : LOAMX_OK;
}

void select_device(int device);   // throws LOAMX_E_NOGPU

// roctx range around a host-side stage (tracing): shows up in `rocprofv3 --marker-trace` next to the kernels the
// stage enqueues; costs two calls into an unloaded tool library otherwise
struct TraceRange {
#ifndef LOAMX_NO_ROCTX
  explicit TraceRange(const char* name) { roctxRangePush(name); }
  ~TraceRange() { roctxRangePop(); }
#else
  explicit TraceRange(const char*) {}
#endif
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

// Diagnostic switches — timing experiments, A/B runs of retired code paths, anything that changes RESULTS (LOAMX_ODOM_MAXIT,
// LOAMX_NO_MAP_IMU_BLEND, LOAMX_EIG_JACOBI ...) — exist only in a build made with `make EXTRA=-DLOAMX_DIAG`; the product library does
// not read them (loamx_build_info() says which kind a library is).  What the product does read from the environment is listed in
// tracing (LOAMX_*_TRACE), transport selection and the look-ahead / chain tuning knobs, none of which changes a result.
#ifdef LOAMX_DIAG
inline const char* diag_env(const char* name) { return getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif

// Wait for a stream by polling (hipStreamQuery) for up to `spin_ms` before falling back to hipStreamSynchronize.  The runtime's blocking
// wait sleeps on an interrupt after a short spin; on the sequential-SLAM path — a wait every ~0.3 ms — one wake-up in a few hundred
// arrived ~10 ms late on some hosts (one such call in a window of 100 sweeps is 15 % of the window: profiles/r06_ab.md section 8).
inline void spin_sync(hipStream_t st, int spin_ms = 4) {
  const auto t_in = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) { LX_HIP(e); }
    if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t_in > std::chrono::milliseconds(spin_ms)) break;
    __builtin_ia32_pause();
  }
  LX_HIP(hipStreamSynchronize(st));
}

inline void spin_event(hipEvent_t ev, int spin_ms = 4) {   // the same for an event
  const auto t_in = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) { LX_HIP(e); }
    if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t_in > std::chrono::milliseconds(spin_ms)) break;
    __builtin_ia32_pause();
  }
  LX_HIP(hipEventSynchronize(ev));
}

// The steady-state waits of the chains (the odometry's pose event, the features' event in front of a pass, the registration's results):
// the runtime's blocking waits by default; LOAMX_WAIT_SPIN=1: polling for up to 4 ms before they block (spin_event / spin_sync) — +1-1.5 %
// on the sequential chain in a process of its own, neutral on the batched one, but -3 % on the HDL-32 chain beside busy host cores (the
// bench's child processes): profiles/r06_ab.md section 20
inline bool wait_spins() { static const bool on = getenv("LOAMX_WAIT_SPIN") != nullptr && atoi(getenv("LOAMX_WAIT_SPIN")) != 0; return on; }
inline void wait_event(hipEvent_t ev) { if (wait_spins()) spin_event(ev); else LX_HIP(hipEventSynchronize(ev)); }
inline void wait_stream(hipStream_t st) { if (wait_spins()) spin_sync(st); else LX_HIP(hipStreamSynchronize(st)); }

// HIP stream with a relative priority: +1 = highest the device offers, 0 = default, -1 = lowest.  The stages of the
// pipeline run on streams of their own; the latency-critical ones (registration, odometry) outrank feature extraction,
// whose wide kernels would otherwise delay their short dependent launches.
// env_priority (diagnostic): the stream priority a stage asked for can be overridden with LOAMX_PRIO_REG / _ODOM / _FEAT = -1, 0, 1
inline int env_priority(const char* name, int dflt) {
  const char* e = diag_env(name);
  if (!e) return dflt;
  const int v = atoi(e);
  return v > 0 ? 1 : (v < 0 ? -1 : 0);
}
// cu_stride > 1 (diagnostic, LOAMX_FEAT_CU_STRIDE): a stream whose kernels only run on every cu_stride-th compute unit
// cu_split (diagnostic, LOAMX_CU_SPLIT=n): the device's compute units in two sets — mask bits [0, n) for the streams created with part 0
// (the odometry chains), the rest for those created with part 1 (registration, features): do the chains' persistent workgroups run
// their iterations faster when nothing else shares their compute units?  (a masked stream has no priority: the runtime offers no call for both)
inline hipStream_t create_stream(int rel_priority, int cu_stride = 0, int part = -1) {
  if (part >= 0 && diag_env("LOAMX_CU_SPLIT") && atoi(diag_env("LOAMX_CU_SPLIT")) > 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    LX_HIP(hipGetDevice(&dev));
    LX_HIP(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount, n = std::min(atoi(diag_env("LOAMX_CU_SPLIT")), ncu - 8);
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int i = (part == 0 ? 0 : n); i < (part == 0 ? n : ncu); i++) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipStream_t st = nullptr;
    LX_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    return st;
  }
  if (cu_stride > 1) {
    hipDeviceProp_t prop;
    int dev = 0;
    LX_HIP(hipGetDevice(&dev));
    LX_HIP(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu; i += cu_stride) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipStream_t st = nullptr;
    LX_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    return st;
  }
  int lo = 0, hi = 0;   // numerically lower = higher priority
  LX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  const int prio = rel_priority > 0 ? hi : (rel_priority < 0 ? lo : (lo + hi) / 2);
  hipStream_t st = nullptr;
  LX_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
  return st;
}

}  // namespace loamx

#ifdef LOAMX_API_TRACE
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                 \
  do {                                                      \
    LX_API_TIMER(#kernelName);                              \
    hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);  \
  } while (0)
#endif
