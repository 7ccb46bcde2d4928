// Device -> pinned-host block copies on the GPU's SDMA engine, issued through ROCr (hsa_amd_memory_async_copy) instead of
// hipMemcpyAsync.  Why: the HIP runtime chooses the engine of a pinned copy itself and has been measured (profiles/r03_pcie.md) to
// carry the pipeline's 16 MiB result download with its 256-workgroup blit KERNEL while a host -> device copy is in flight.  Every
// other kernel that ends with a store to host memory (flags, counts, the small result copies of the three chains) then queues
// behind 16 MiB of posted PCIe writes: k_feat_lf_voxel 74 -> 329 us, k_odom_corr 62 -> 344 us, the step 0.63 -> 0.94 ms.  The same
// copy on an SDMA engine does not disturb them (scripts/micro/d2h.hip: one-word device -> host round trip 6.5 us p95 7.6 beside an
// SDMA copy, p95 250 us beside a 256-workgroup copy kernel).  ROCr keeps separate engines for the two directions of the host link.
//
// Only memory that ROCr itself allocated is taken (hipMalloc / hipHostMalloc and what sits on them: torch's pinned tensors);
// anything else — registered or pageable host memory — is refused and the caller goes through hipMemcpyAsync as before.
#pragma once
#include "common.h"
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

namespace loamx {

template <int SLOTS> class HostLinkT {
 public:
  ~HostLinkT() {
    for (int s = 0; s < SLOTS; s++) {
      if (!made_[s]) continue;
      try { wait(s); } catch (...) {}
      (void)hsa_signal_destroy(sig_[s]);
    }
  }
  // true when [dev_src, +bytes) is device memory of a GPU agent and [host_dst, +bytes) host memory of a CPU agent, both ROCr's own
  bool can_copy(void* host_dst, const void* dev_src, size_t bytes) {
    if (!init_()) return false;
    hsa_agent_t g, c;
    return owner_(dev_src, bytes, HSA_DEVICE_TYPE_GPU, &g) && owner_(host_dst, bytes, HSA_DEVICE_TYPE_CPU, &c);
  }
  // Begin a group of n copies that complete slot's signal together.  The data must be complete and visible (the caller has
  // synchronised with the kernels that wrote it): the engine starts at once.
  void begin(int slot, uint32_t n) {
    LX_REQUIRE(init_(), "ROCr is not available");
    wait(slot);
    if (!made_[slot]) {
      LX_REQUIRE(hsa_signal_create(0, 0, nullptr, &sig_[slot]) == HSA_STATUS_SUCCESS, "hsa_signal_create failed");
      made_[slot] = true;
    }
    hsa_signal_store_relaxed(sig_[slot], (hsa_signal_value_t)n);
    pending_[slot] = n > 0;
    unissued_[slot] = n;
  }
  // A group that cannot be completed (a copy failed: the caller falls back to hipMemcpyAsync): the copies that were never issued are
  // taken off the signal, the issued ones are waited for, the slot is free again.  (a failing copy used to leave
  // the signal above zero for good and every later wait(slot) blocked for ever.)
  void abandon(int slot) {
    if (!pending_[slot]) return;
    if (unissued_[slot]) hsa_signal_subtract_relaxed(sig_[slot], (hsa_signal_value_t)unissued_[slot]);
    unissued_[slot] = 0;
    try { wait(slot); } catch (...) { pending_[slot] = false; }
  }
  void copy_d2h(int slot, void* host_dst, const void* dev_src, size_t bytes) {
    hsa_agent_t g, c;
    LX_REQUIRE(owner_(dev_src, bytes, HSA_DEVICE_TYPE_GPU, &g) && owner_(host_dst, bytes, HSA_DEVICE_TYPE_CPU, &c), "not ROCr-allocated memory");
    hsa_status_t st = HSA_STATUS_ERROR;
    const uint32_t eng = engine_(c, g);
    if (eng) {
      st = hsa_amd_memory_async_copy_on_engine(host_dst, c, dev_src, g, bytes, 0, nullptr, sig_[slot], (hsa_amd_sdma_engine_id_t)eng, false);
      if (st != HSA_STATUS_SUCCESS) engine_mask_ = 0;   // not available on this device: ROCr chooses from now on
    }
    if (st != HSA_STATUS_SUCCESS) st = hsa_amd_memory_async_copy(host_dst, c, dev_src, g, bytes, 0, nullptr, sig_[slot]);
    if (st != HSA_STATUS_SUCCESS) {   // this copy and the rest of its group will never complete the signal: see abandon()
      abandon(slot);
      throw Error(LOAMX_E_HIP, "hsa_amd_memory_async_copy failed (" + std::to_string((int)st) + ")");
    }
    if (unissued_[slot]) unissued_[slot]--;
  }
  // ... and the other direction: pinned host -> device (the pipeline's staging copies; round 4).  A block copy that the HIP runtime
  // carries on a stream of its own is a FIFTH busy HIP stream next to the four chains — and from five busy streams on every chain of the
  // process slows down (profiles/r04_groups_ab.md; the odometry passes took 0.7-1.1 ms instead of 0.3-0.6 beside the staging stream).
  // Issued through ROCr it is no HIP stream at all.  The engine is ROCr's own choice for this direction.
  bool can_copy_h2d(void* dev_dst, const void* host_src, size_t bytes) {
    if (!init_()) return false;
    hsa_agent_t g, c;
    return owner_(dev_dst, bytes, HSA_DEVICE_TYPE_GPU, &g) && owner_(host_src, bytes, HSA_DEVICE_TYPE_CPU, &c);
  }
  bool host_ok(const void* host, size_t bytes) {   // the host side alone (before the destination exists)
    hsa_agent_t c;
    return init_() && owner_(host, bytes, HSA_DEVICE_TYPE_CPU, &c);
  }
  void copy_h2d(int slot, void* dev_dst, const void* host_src, size_t bytes) {
    hsa_agent_t g, c;
    LX_REQUIRE(owner_(dev_dst, bytes, HSA_DEVICE_TYPE_GPU, &g) && owner_(host_src, bytes, HSA_DEVICE_TYPE_CPU, &c), "not ROCr-allocated memory");
    const hsa_status_t st = hsa_amd_memory_async_copy(dev_dst, g, host_src, c, bytes, 0, nullptr, sig_[slot]);
    if (st != HSA_STATUS_SUCCESS) {
      abandon(slot);
      throw Error(LOAMX_E_HIP, "hsa_amd_memory_async_copy (host -> device) failed (" + std::to_string((int)st) + ")");
    }
    if (unissued_[slot]) unissued_[slot]--;
  }
  uint32_t engine() const { return engine_mask_ == ~0u ? 0u : engine_mask_; }
  bool pending(int slot) const { return pending_[slot]; }
  // blocks until every copy of the slot's group has landed (and is visible to the host)
  void wait(int slot) {
    if (!pending_[slot]) return;
    hsa_signal_value_t v;
    while ((v = hsa_signal_wait_scacquire(sig_[slot], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED)) > 0) {}
    pending_[slot] = false;
    if (v < 0) throw Error(LOAMX_E_HIP, "a device -> host DMA reported an error");
  }

 private:
  static bool init_() {   // (reference-counted: the HIP runtime of this process holds ROCr already; this reference is kept for good)
    static const bool ok = hsa_init() == HSA_STATUS_SUCCESS;
    return ok;
  }
  static bool owner_(const void* p, size_t bytes, hsa_device_type_t want, hsa_agent_t* agent) {
    hsa_amd_pointer_info_t info;
    info.size = sizeof(info);
    if (hsa_amd_pointer_info(p, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA) return false;
    const char* base = (const char*)(want == HSA_DEVICE_TYPE_CPU && info.hostBaseAddress ? info.hostBaseAddress : info.agentBaseAddress);
    if (!base || (const char*)p < base || (const char*)p + bytes > base + info.sizeInBytes) return false;
    hsa_device_type_t type;
    if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != want) return false;
    *agent = info.agentOwner;
    return true;
  }
  // The engine of the downloads.  The HIP runtime gives the first free engine (ENGINE_0) to the first block copy it issues — the
  // pipeline's host -> device staging — and two directions on one engine take turns; so: the first engine ROCr recommends for this
  // direction other than ENGINE_0, else ENGINE_1.  LOAMX_D2H_ENGINE=<mask bit> overrides (0: ROCr's own choice per copy).
  uint32_t engine_(hsa_agent_t cpu, hsa_agent_t gpu) {
    if (engine_mask_ == ~0u) {
      if (const char* e = diag_env("LOAMX_D2H_ENGINE")) engine_mask_ = (uint32_t)strtoul(e, nullptr, 0);
      else {
        uint32_t rec = 0;
        if (hsa_amd_memory_get_preferred_copy_engine(cpu, gpu, &rec) != HSA_STATUS_SUCCESS) rec = 0;
        rec &= ~(uint32_t)HSA_AMD_SDMA_ENGINE_0;
        engine_mask_ = rec ? (rec & (~rec + 1u)) : (uint32_t)HSA_AMD_SDMA_ENGINE_1;
      }
    }
    return engine_mask_;
  }
  uint32_t engine_mask_ = ~0u;
  hsa_signal_t sig_[SLOTS] = {};
  bool made_[SLOTS] = {}, pending_[SLOTS] = {};
  uint32_t unissued_[SLOTS] = {};   // copies of the slot's group that have not been handed to ROCr yet
};
using HostLinkDma = HostLinkT<2>;   // the downloads: two alternating buffers
using HostLinkUp = HostLinkT<8>;    // the staging copies: one slot of the streaming ring each

}  // namespace loamx
