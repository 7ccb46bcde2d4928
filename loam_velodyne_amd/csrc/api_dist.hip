// C-ABI: multi-GPU plumbing of the batched-sweep mode (loamx_dist_*) over RCCL — SURVEY.md §8(e).
//
// The path shards by independent sweeps (rank r registers its contiguous share of the batch against a replica of the
// frozen map): no data-path collective.  Two exchanges exist and both live here, natively, so that a C++ / ROS host gets
// the multi-GPU mode without a Python runtime:
//   map epoch      ncclBroadcast of the corner and surf sub-map buffers root -> all, on the communicator's own HIP stream,
//                  returning an event; loamx_{batch,pipeline}_stage_frozen_device orders its index build behind that event
//                  on the device, so the broadcast of epoch k+1 overlaps the registrations of epoch k (double buffering,
//                  BASELINE configs[4])
//   results        ncclAllGather of n_local x (6 pose floats + iterations + flags) per rank — a few hundred bytes
// xGMI is point to point (7 links x ~153 GB/s per GPU): a ring broadcast of the 16-32 MB map is per-link bound
// (~0.1-0.2 ms); it is issued as ONE collective per buffer (no bucketing needed at this size).
// One process per GPU; the 128-byte ncclUniqueId travels between the processes by the host's own means (a file, MPI,
// a socket — loam_velodyne_amd/launch.py uses a file).
#include "common.h"
#include <rccl/rccl.h>
#include <memory>

namespace loamx {
#define LX_NCCL(expr)                                                                                                 \
  do {                                                                                                                \
    ncclResult_t r_ = (expr);                                                                                         \
    if (r_ != ncclSuccess)                                                                                            \
      throw ::loamx::Error(LOAMX_E_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_) + " (" + __FILE__ + ":" + \
                                            std::to_string(__LINE__) + ")");                                         \
  } while (0)
}  // namespace loamx

using namespace loamx;

struct loamx_dist {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t ev_bcast = nullptr;
  DevBuf<float> d_send, d_recv;
  PinBuf<float> h_send, h_recv;
  ~loamx_dist() {
    if (comm) (void)ncclCommDestroy(comm);
    if (ev_bcast) (void)hipEventDestroy(ev_bcast);
    if (st) (void)hipStreamDestroy(st);
  }
};

static_assert(sizeof(ncclUniqueId) == LOAMX_DIST_ID_BYTES, "LOAMX_DIST_ID_BYTES must be sizeof(ncclUniqueId)");

extern "C" {

int loamx_dist_get_unique_id(unsigned char id[LOAMX_DIST_ID_BYTES]) {
  return guard([&]() {
    LX_REQUIRE(id, "NULL argument");
    ncclUniqueId u;
    LX_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return LOAMX_OK;
  });
}

loamx_dist* loamx_dist_create(const unsigned char id[LOAMX_DIST_ID_BYTES], int rank, int world_size, int device) {
  loamx_dist* h = nullptr;
  guard([&]() {
    LX_REQUIRE(id && world_size >= 1 && rank >= 0 && rank < world_size, "invalid rank / world size");
    select_device(device);
    std::unique_ptr<loamx_dist> d(new loamx_dist());
    d->rank = rank; d->world = world_size; d->device = device;
    d->st = create_stream(0);
    LX_HIP(hipEventCreateWithFlags(&d->ev_bcast, hipEventDisableTiming));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    LX_NCCL(ncclCommInitRank(&d->comm, world_size, u, rank));
    h = d.release();
    return LOAMX_OK;
  });
  return h;
}

void loamx_dist_destroy(loamx_dist* h) { delete h; }
int loamx_dist_rank(const loamx_dist* h) { return h ? h->rank : -1; }
int loamx_dist_world_size(const loamx_dist* h) { return h ? h->world : 0; }

int loamx_dist_shard(const loamx_dist* h, uint32_t batch, uint32_t* begin, uint32_t* end) {
  return guard([&]() {
    LX_REQUIRE(h && begin && end, "NULL argument");
    // GPU g of G takes sweeps [g*B/G, (g+1)*B/G)  (SURVEY.md §8e "Partitioning")
    *begin = (uint32_t)((uint64_t)h->rank * batch / (uint64_t)h->world);
    *end = (uint32_t)((uint64_t)(h->rank + 1) * batch / (uint64_t)h->world);
    return LOAMX_OK;
  });
}

int loamx_dist_broadcast_map(loamx_dist* h, void* d_corner_xyzi, uint32_t n_corner, void* d_surf_xyzi, uint32_t n_surf, int root,
                             void* wait_event, void** done_event) {
  return guard([&]() {
    LX_REQUIRE(h && (d_corner_xyzi || !n_corner) && (d_surf_xyzi || !n_surf), "NULL argument");
    LX_REQUIRE(root >= 0 && root < h->world, "root out of range");
    LX_HIP(hipSetDevice(h->device));
    TraceRange trace_range("loamx:dist:broadcast_map");
    if (wait_event) LX_HIP(hipStreamWaitEvent(h->st, (hipEvent_t)wait_event, 0));   // whatever filled the root's buffers
    LX_NCCL(ncclGroupStart());
    if (n_corner) LX_NCCL(ncclBroadcast(d_corner_xyzi, d_corner_xyzi, (size_t)4 * n_corner, ncclFloat, root, h->comm, h->st));
    if (n_surf) LX_NCCL(ncclBroadcast(d_surf_xyzi, d_surf_xyzi, (size_t)4 * n_surf, ncclFloat, root, h->comm, h->st));
    LX_NCCL(ncclGroupEnd());
    LX_HIP(hipEventRecord(h->ev_bcast, h->st));
    if (done_event) *done_event = (void*)h->ev_bcast;
    return LOAMX_OK;
  });
}

int loamx_dist_allgather_results(loamx_dist* h, const float* poses6, const int* iters_flags2, uint32_t n_local, float* poses6_all,
                                 int* iters_flags2_all) {
  return guard([&]() {
    LX_REQUIRE(h && poses6 && poses6_all, "NULL argument");
    LX_HIP(hipSetDevice(h->device));
    TraceRange trace_range("loamx:dist:allgather_results");
    const size_t rec = 8, nl = (size_t)n_local * rec, na = nl * (size_t)h->world;   // 6 pose floats + iterations + flags
    if (!nl) return (int)LOAMX_OK;
    h->h_send.reserve(nl); h->h_recv.reserve(na); h->d_send.reserve(nl); h->d_recv.reserve(na);
    for (uint32_t i = 0; i < n_local; i++) {
      memcpy(h->h_send.p + rec * i, poses6 + 6 * i, 6 * sizeof(float));
      int tail[2] = {iters_flags2 ? iters_flags2[2 * i] : 0, iters_flags2 ? iters_flags2[2 * i + 1] : 0};
      memcpy(h->h_send.p + rec * i + 6, tail, sizeof(tail));
    }
    LX_HIP(hipMemcpyAsync(h->d_send.p, h->h_send.p, nl * sizeof(float), hipMemcpyHostToDevice, h->st));
    LX_NCCL(ncclAllGather(h->d_send.p, h->d_recv.p, nl, ncclFloat, h->comm, h->st));
    LX_HIP(hipMemcpyAsync(h->h_recv.p, h->d_recv.p, na * sizeof(float), hipMemcpyDeviceToHost, h->st));
    LX_HIP(hipStreamSynchronize(h->st));
    for (size_t i = 0; i < (size_t)n_local * h->world; i++) {
      memcpy(poses6_all + 6 * i, h->h_recv.p + rec * i, 6 * sizeof(float));
      if (iters_flags2_all) memcpy(iters_flags2_all + 2 * i, h->h_recv.p + rec * i + 6, 2 * sizeof(int));
    }
    return (int)LOAMX_OK;
  });
}

int loamx_dist_barrier(loamx_dist* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    LX_HIP(hipSetDevice(h->device));
    h->d_send.reserve(1); h->d_recv.reserve(1);
    LX_HIP(hipMemsetAsync(h->d_send.p, 0, sizeof(float), h->st));
    LX_NCCL(ncclAllReduce(h->d_send.p, h->d_recv.p, 1, ncclFloat, ncclSum, h->comm, h->st));
    LX_HIP(hipStreamSynchronize(h->st));
    return LOAMX_OK;
  });
}

void* loamx_dist_stream(loamx_dist* h) { return h ? (void*)h->st : nullptr; }

}  // extern "C"
