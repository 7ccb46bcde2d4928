// Process-wide pieces of the C-ABI: last-error slot, device selection, version.
#include "common.h"

namespace loamx {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }

void select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw Error(LOAMX_E_NOGPU, "no HIP device visible: libloamx has no CPU fallback (hipGetDeviceCount: " +
                                   std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
  if (device < 0 || device >= n) throw Error(LOAMX_E_INVALID, "device ordinal out of range");
  LX_HIP(hipSetDevice(device));
}
}  // namespace loamx

extern "C" {
const char* loamx_last_error(void) { return loamx::g_last_error.c_str(); }
int loamx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int loamx_abi_version(void) { return LOAMX_ABI_VERSION; }
void* loamx_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void loamx_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
#define LX_STR2(x) #x
#define LX_STR(x) LX_STR2(x)
const char* loamx_build_info(void) {
  return "abi=" LX_STR(LOAMX_ABI_VERSION)
#ifdef LOAMX_DIAG
         ";diag=1"
#else
         ";diag=0"
#endif
#ifdef LOAMX_NO_RCCL
         ";rccl=0"
#else
         ";rccl=1"
#endif
#ifdef LOAMX_NO_ROCTX
         ";roctx=0"
#else
         ";roctx=1"
#endif
      ;
}
}
