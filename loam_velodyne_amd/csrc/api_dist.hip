// C-ABI: multi-GPU plumbing of the batched-sweep mode (loamx_dist_*) over RCCL — SURVEY.md §8(e).
//
// The path shards by independent sweeps (rank r registers its contiguous share of the batch against a replica of the
// frozen map): no data-path collective.  Two exchanges exist and both live here, natively, so that a C++ / ROS host gets
// the multi-GPU mode without a Python runtime:
//   map epoch      ncclBroadcast of the corner and surf sub-map buffers root -> all, on the communicator's own HIP stream,
//                  returning an event; loamx_{batch,pipeline}_stage_frozen_device orders its index build behind that event
//                  on the device, so the broadcast of epoch k+1 overlaps the registrations of epoch k (double buffering,
//                  BASELINE configs[4])
//   results        two ncclAllGathers: the ranks' record counts (shards are unequal whenever the batch does not divide by the
//                  world size), then ceil(B / G) records of (6 pose floats + iterations + flags) per rank, padded — a few hundred bytes
// xGMI is point to point (7 links x ~153 GB/s per GPU): a ring broadcast of the 16-32 MB map is per-link bound
// (~0.1-0.2 ms); it is issued as ONE collective per buffer (no bucketing needed at this size).
// One process per GPU; the 128-byte ncclUniqueId travels between the processes by the host's own means (a file, MPI,
// a socket — loam_velodyne_amd/launch.py uses a file).
#include "common.h"
#include <memory>
#include <vector>
#ifndef LOAMX_NO_RCCL
#include <rccl/rccl.h>

namespace loamx {
#define LX_NCCL(expr)                                                                                                 \
  do {                                                                                                                \
    ncclResult_t r_ = (expr);                                                                                         \
    if (r_ != ncclSuccess)                                                                                            \
      throw ::loamx::Error(LOAMX_E_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_) + " (" + __FILE__ + ":" + \
                                            std::to_string(__LINE__) + ")");                                         \
  } while (0)
}  // namespace loamx
#else
// make NO_RCCL=1 (a ROCm install without RCCL: the single-GPU drop-in needs none of it): the entry points below stay exported, the two
// exchanges and the communicator answer LOAMX_E_UNSUPPORTED, the host-side layout functions (shard_of / pack / unpack) work as always.
// The few RCCL names the code below mentions are declared here so that it reads the same in both builds (they do nothing).
struct ncclUniqueId { char internal[LOAMX_DIST_ID_BYTES]; };
typedef void* ncclComm_t;
typedef int ncclResult_t;
enum { ncclFloat = 0, ncclUint32 = 1, ncclSum = 0 };
#define LX_NCCL(expr) do { (void)(expr); throw ::loamx::Error(LOAMX_E_UNSUPPORTED, "libloamx was built without RCCL (make NO_RCCL=1): " #expr); } while (0)
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId*) { return 1; }
static inline ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int) { return 1; }
static inline ncclResult_t ncclCommDestroy(ncclComm_t) { return 0; }
static inline ncclResult_t ncclCommCount(ncclComm_t, int*) { return 1; }
static inline ncclResult_t ncclGroupStart() { return 1; }
static inline ncclResult_t ncclGroupEnd() { return 1; }
static inline ncclResult_t ncclBroadcast(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) { return 1; }
static inline ncclResult_t ncclAllGather(const void*, void*, size_t, int, ncclComm_t, hipStream_t) { return 1; }
static inline ncclResult_t ncclAllReduce(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) { return 1; }
static inline ncclResult_t ncclSend(const void*, size_t, int, int, ncclComm_t, hipStream_t) { return 1; }
static inline ncclResult_t ncclRecv(void*, size_t, int, int, ncclComm_t, hipStream_t) { return 1; }
#endif

using namespace loamx;

// ncclGroupStart ... ncclGroupEnd with the end guaranteed: a throw between the two (LX_NCCL on a failed send / recv) would otherwise leave
// the thread's group open and every later collective of the process queued inside it
struct NcclGroup {
  bool open = false;
  NcclGroup() { LX_NCCL(ncclGroupStart()); open = true; }
  void end() { open = false; LX_NCCL(ncclGroupEnd()); }
  ~NcclGroup() { if (open) (void)ncclGroupEnd(); }
};

struct loamx_dist {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t ev_bcast = nullptr;
  DevBuf<float> d_send, d_recv;
  PinBuf<float> h_send, h_recv;
  DevBuf<uint32_t> d_cnt;
  PinBuf<uint32_t> h_cnt;
  DevBuf<uint32_t> d_gsend, d_grecv;   // gatherv: this rank's words / (root) every rank's words back to back
  PinBuf<uint32_t> h_gsend, h_grecv;
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
    return loamx_dist_shard_of(h->rank, h->world, batch, begin, end);
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
    {
      NcclGroup grp;
      if (n_corner) LX_NCCL(ncclBroadcast(d_corner_xyzi, d_corner_xyzi, (size_t)4 * n_corner, ncclFloat, root, h->comm, h->st));
      if (n_surf) LX_NCCL(ncclBroadcast(d_surf_xyzi, d_surf_xyzi, (size_t)4 * n_surf, ncclFloat, root, h->comm, h->st));
      grp.end();
    }
    LX_HIP(hipEventRecord(h->ev_bcast, h->st));
    if (done_event) *done_event = (void*)h->ev_bcast;
    return LOAMX_OK;
  });
}

// ---- host-side record layout of the result exchange (no device, no communicator: a host with its own transport — MPI, a socket —
// uses the same three functions; tests/test_dist_cpu.py drives them over gloo)
int loamx_dist_shard_of(int rank, int world_size, uint32_t batch, uint32_t* begin, uint32_t* end) {
  return guard([&]() {
    LX_REQUIRE(begin && end && world_size >= 1 && rank >= 0 && rank < world_size, "invalid argument");
    // GPU g of G takes sweeps [g*B/G, (g+1)*B/G)  (SURVEY.md §8e "Partitioning")
    *begin = (uint32_t)((uint64_t)rank * batch / (uint64_t)world_size);
    *end = (uint32_t)((uint64_t)(rank + 1) * batch / (uint64_t)world_size);
    return LOAMX_OK;
  });
}
int loamx_dist_pack_results(const float* poses6, const int* iters_flags2, uint32_t n_local, uint32_t n_pad, float* send8) {
  return guard([&]() {
    LX_REQUIRE((poses6 || !n_local) && send8 && n_local <= n_pad, "invalid argument");
    for (uint32_t i = 0; i < n_pad; i++) {
      float* r = send8 + (size_t)LOAMX_DIST_RECORD_FLOATS * i;
      if (i < n_local) {
        memcpy(r, poses6 + 6 * (size_t)i, 6 * sizeof(float));
        const int tail[2] = {iters_flags2 ? iters_flags2[2 * (size_t)i] : 0, iters_flags2 ? iters_flags2[2 * (size_t)i + 1] : 0};
        memcpy(r + 6, tail, sizeof(tail));
      } else {
        memset(r, 0, LOAMX_DIST_RECORD_FLOATS * sizeof(float));   // padding of a shorter shard
      }
    }
    return LOAMX_OK;
  });
}
int loamx_dist_unpack_results(const float* recv8, const uint32_t* counts, int world_size, uint32_t n_pad, float* poses6_all, int* iters_flags2_all) {
  return guard([&]() {
    LX_REQUIRE(recv8 && counts && poses6_all && world_size >= 1, "invalid argument");
    size_t o = 0;
    for (int r = 0; r < world_size; r++) {
      LX_REQUIRE(counts[r] <= n_pad, "a rank's record count exceeds the padded size");
      for (uint32_t i = 0; i < counts[r]; i++, o++) {
        const float* rec = recv8 + (size_t)LOAMX_DIST_RECORD_FLOATS * ((size_t)r * n_pad + i);
        memcpy(poses6_all + 6 * o, rec, 6 * sizeof(float));
        if (iters_flags2_all) memcpy(iters_flags2_all + 2 * o, rec + 6, 2 * sizeof(int));
      }
    }
    return LOAMX_OK;
  });
}

static int dist_gather_counts(loamx_dist* h, uint32_t n_local) {   // -> h->h_cnt.p[0 .. world)
  const int G = h->world;
  h->h_cnt.reserve((size_t)G + 1); h->d_cnt.reserve((size_t)G + 1);
  h->h_cnt.p[G] = n_local;
  LX_HIP(hipMemcpyAsync(h->d_cnt.p + G, h->h_cnt.p + G, sizeof(uint32_t), hipMemcpyHostToDevice, h->st));
  LX_NCCL(ncclAllGather(h->d_cnt.p + G, h->d_cnt.p, 1, ncclUint32, h->comm, h->st));
  LX_HIP(hipMemcpyAsync(h->h_cnt.p, h->d_cnt.p, sizeof(uint32_t) * G, hipMemcpyDeviceToHost, h->st));
  LX_HIP(hipStreamSynchronize(h->st));
  return LOAMX_OK;
}

int loamx_dist_allgather_counts(loamx_dist* h, uint32_t n_local, uint32_t* counts_all) {
  return guard([&]() {
    LX_REQUIRE(h && counts_all, "NULL argument");
    LX_HIP(hipSetDevice(h->device));
    dist_gather_counts(h, n_local);
    memcpy(counts_all, h->h_cnt.p, sizeof(uint32_t) * h->world);
    return (int)LOAMX_OK;
  });
}

int loamx_dist_allgather_results_cap(loamx_dist* h, const float* poses6, const int* iters_flags2, uint32_t n_local, float* poses6_all,
                                     int* iters_flags2_all, uint32_t capacity_records, uint32_t* counts_all) {
  return guard([&]() {
    LX_REQUIRE(h && (poses6 || !n_local) && poses6_all, "NULL argument");
    LX_HIP(hipSetDevice(h->device));
    TraceRange trace_range("loamx:dist:allgather_results");
    const int G = h->world;
    // 1. every rank's record count (shards differ by one whenever the batch does not divide by the world size; a rank with an
    //    empty shard still takes part in both collectives)
    dist_gather_counts(h, n_local);
    uint32_t n_pad = 0;
    unsigned long long total = 0;
    for (int r = 0; r < G; r++) { n_pad = std::max(n_pad, h->h_cnt.p[r]); total += h->h_cnt.p[r]; }
    if (counts_all) memcpy(counts_all, h->h_cnt.p, sizeof(uint32_t) * G);
    if (!n_pad) return (int)LOAMX_OK;   // (every rank sees the same counts: all of them leave here)
    // 2. the records, padded to the longest shard
    const size_t nl = (size_t)n_pad * LOAMX_DIST_RECORD_FLOATS, na = nl * (size_t)G;
    h->h_send.reserve(nl); h->h_recv.reserve(na); h->d_send.reserve(nl); h->d_recv.reserve(na);
    int rc = loamx_dist_pack_results(poses6, iters_flags2, n_local, n_pad, h->h_send.p);
    if (rc != LOAMX_OK) return rc;
    LX_HIP(hipMemcpyAsync(h->d_send.p, h->h_send.p, nl * sizeof(float), hipMemcpyHostToDevice, h->st));
    LX_NCCL(ncclAllGather(h->d_send.p, h->d_recv.p, nl, ncclFloat, h->comm, h->st));
    LX_HIP(hipMemcpyAsync(h->h_recv.p, h->d_recv.p, na * sizeof(float), hipMemcpyDeviceToHost, h->st));
    LX_HIP(hipStreamSynchronize(h->st));
    // (the capacity is a LOCAL matter: it is checked after the collectives, so a rank with too small arrays fails alone)
    if (total > capacity_records) throw Error(LOAMX_E_CAPACITY, "the ranks' records do not fit the receive arrays (counts_all holds the counts)");
    return loamx_dist_unpack_results(h->h_recv.p, h->h_cnt.p, G, n_pad, poses6_all, iters_flags2_all);
  });
}

int loamx_dist_allgather_results(loamx_dist* h, const float* poses6, const int* iters_flags2, uint32_t n_local, float* poses6_all,
                                 int* iters_flags2_all, uint32_t* counts_all) {
  return loamx_dist_allgather_results_cap(h, poses6, iters_flags2, n_local, poses6_all, iters_flags2_all, 0xffffffffu, counts_all);
}

// ---- the epoch's merge step (collective 3): the sweeps a rank has registered travel to the rank that owns the map
// accumulator.  Layout of one rank's message, in 32-bit words (host side, no device: testable without a GPU, usable over any transport):
//   [0] magic 'LXCL'  [1] n_streams  [2 + 2 s] n_corner(s)  [3 + 2 s] n_surf(s)  | 6 n_streams pose floats (transformAftMapped) |
//   the points, x y z intensity each: corner(0), surf(0), corner(1), surf(1), ...
static constexpr uint32_t LX_CLOUD_MAGIC = 0x4c58434cu;
int loamx_dist_pack_clouds(uint32_t n_streams, const loamx_cloud* corner, const loamx_cloud* surf, const float* poses6, uint32_t* words,
                           uint64_t capacity_words, uint64_t* n_words) {
  return guard([&]() {
    LX_REQUIRE(n_words && (n_streams == 0 || (corner && surf && poses6)), "NULL argument");
    uint64_t need = 2 + 2ull * n_streams + 6ull * n_streams;
    for (uint32_t s = 0; s < n_streams; s++) {
      check_cloud(&corner[s], false); check_cloud(&surf[s], false);
      need += 4ull * corner[s].count + 4ull * surf[s].count;
    }
    *n_words = need;
    if (!words) return (int)LOAMX_OK;   // size query
    if (need > capacity_words) throw Error(LOAMX_E_CAPACITY, "the packed clouds do not fit (n_words holds the size)");
    words[0] = LX_CLOUD_MAGIC; words[1] = n_streams;
    for (uint32_t s = 0; s < n_streams; s++) { words[2 + 2 * s] = corner[s].count; words[3 + 2 * s] = surf[s].count; }
    uint32_t* w = words + 2 + 2 * (size_t)n_streams;
    memcpy(w, poses6, sizeof(float) * 6 * n_streams);
    w += 6 * (size_t)n_streams;
    for (uint32_t s = 0; s < n_streams; s++) {   // (the words are 4-byte aligned: plain float copies, no 16-byte vector stores)
      const loamx_cloud* two[2] = {&corner[s], &surf[s]};
      for (const loamx_cloud* c : two) {
        const char* src = (const char*)c->data;
        for (uint32_t i = 0; i < c->count; i++) {
          const char* r = src + (size_t)i * c->stride;
          memcpy(w, r, 12);
          memcpy(w + 3, r + c->intensity_offset, 4);
          w += 4;
        }
      }
    }
    return (int)LOAMX_OK;
  });
}
int loamx_dist_unpack_clouds_header(const uint32_t* words, uint64_t n_words, uint32_t* n_streams, uint32_t* n_corner, uint32_t* n_surf, uint32_t capacity_streams) {
  return guard([&]() {
    LX_REQUIRE(words && n_streams, "NULL argument");
    LX_REQUIRE(n_words >= 2 && words[0] == LX_CLOUD_MAGIC, "not a packed-cloud message");
    const uint32_t ns = words[1];
    *n_streams = ns;
    uint64_t need = 2 + 8ull * ns;
    LX_REQUIRE(n_words >= need, "packed-cloud message truncated (header)");
    for (uint32_t s = 0; s < ns; s++) need += 4ull * words[2 + 2 * s] + 4ull * words[3 + 2 * s];
    LX_REQUIRE(n_words == need, "packed-cloud message size does not match its header");
    if (n_corner || n_surf) {
      if (ns > capacity_streams) throw Error(LOAMX_E_CAPACITY, "more streams in the message than the count arrays hold");
      for (uint32_t s = 0; s < ns; s++) { if (n_corner) n_corner[s] = words[2 + 2 * s]; if (n_surf) n_surf[s] = words[3 + 2 * s]; }
    }
    return (int)LOAMX_OK;
  });
}
int loamx_dist_unpack_clouds_stream(const uint32_t* words, uint64_t n_words, uint32_t stream, float pose6[6], loamx_cloud* corner, loamx_cloud* surf) {
  return guard([&]() {
    LX_REQUIRE(words && pose6 && corner && surf, "NULL argument");
    uint32_t ns = 0;
    int rc = loamx_dist_unpack_clouds_header(words, n_words, &ns, nullptr, nullptr, 0);
    if (rc != LOAMX_OK) return rc;
    LX_REQUIRE(stream < ns, "stream index beyond the message");
    check_cloud(corner, false); check_cloud(surf, false);
    memcpy(pose6, words + 2 + 2 * (size_t)ns + 6 * (size_t)stream, 6 * sizeof(float));
    const uint32_t* w = words + 2 + 8 * (size_t)ns;
    for (uint32_t s = 0; s < stream; s++) w += 4 * ((size_t)words[2 + 2 * s] + words[3 + 2 * s]);
    const uint32_t nc = words[2 + 2 * stream], nsf = words[3 + 2 * stream];
    std::vector<float4> tmp((size_t)std::max(nc, nsf) + 1);
    memcpy(tmp.data(), w, sizeof(float4) * nc);
    const int r1 = unpack_cloud(tmp.data(), nc, corner);
    memcpy(tmp.data(), w + 4 * (size_t)nc, sizeof(float4) * nsf);
    const int r2 = unpack_cloud(tmp.data(), nsf, surf);
    return r1 != LOAMX_OK ? r1 : r2;
  });
}

// Variable-size gather to one rank: every rank's count first (the all-gather of the result exchange), then the words themselves
// point to point — RCCL has no gatherv; the root posts one receive per rank, every rank (the root too) one send, in ONE group.
// recv_words (root only): the ranks' messages back to back in rank order; counts_all (may be NULL): every rank's word count.
int loamx_dist_gatherv(loamx_dist* h, const uint32_t* send_words, uint32_t n_words, int root, uint32_t* recv_words, uint64_t capacity_words,
                       uint32_t* counts_all) {
  return guard([&]() {
    LX_REQUIRE(h && (send_words || !n_words), "NULL argument");
    LX_REQUIRE(root >= 0 && root < h->world, "root out of range");
    LX_HIP(hipSetDevice(h->device));
    TraceRange trace_range("loamx:dist:gatherv");
    const int G = h->world;
    dist_gather_counts(h, n_words);
    if (counts_all) memcpy(counts_all, h->h_cnt.p, sizeof(uint32_t) * G);
    unsigned long long total = 0;
    for (int r = 0; r < G; r++) total += h->h_cnt.p[r];
    if (!total) return (int)LOAMX_OK;   // (every rank sees the same counts: all of them leave here)
    h->h_gsend.reserve((size_t)n_words + 1); h->d_gsend.reserve((size_t)n_words + 1);
    if (n_words) {
      memcpy(h->h_gsend.p, send_words, sizeof(uint32_t) * n_words);
      LX_HIP(hipMemcpyAsync(h->d_gsend.p, h->h_gsend.p, sizeof(uint32_t) * n_words, hipMemcpyHostToDevice, h->st));
    }
    const bool is_root = h->rank == root;
    if (is_root) { h->d_grecv.reserve((size_t)total + 1); h->h_grecv.reserve((size_t)total + 1); }
    {
      NcclGroup grp;
      if (is_root) {
        size_t off = 0;
        for (int r = 0; r < G; r++) {
          if (h->h_cnt.p[r]) LX_NCCL(ncclRecv(h->d_grecv.p + off, h->h_cnt.p[r], ncclUint32, r, h->comm, h->st));
          off += h->h_cnt.p[r];
        }
      }
      if (n_words) LX_NCCL(ncclSend(h->d_gsend.p, n_words, ncclUint32, root, h->comm, h->st));
      grp.end();
    }
    if (is_root) LX_HIP(hipMemcpyAsync(h->h_grecv.p, h->d_grecv.p, sizeof(uint32_t) * total, hipMemcpyDeviceToHost, h->st));
    LX_HIP(hipStreamSynchronize(h->st));
    if (is_root) {
      // (the capacity is the root's own matter: checked after the exchange, so that a root with too small a buffer fails alone)
      LX_REQUIRE(recv_words, "NULL receive buffer on the root");
      if (total > capacity_words) throw Error(LOAMX_E_CAPACITY, "the ranks' words do not fit the receive buffer (counts_all holds the counts)");
      memcpy(recv_words, h->h_grecv.p, sizeof(uint32_t) * total);
    }
    return (int)LOAMX_OK;
  });
}

int loamx_dist_comm_count(loamx_dist* h) {   // ranks the RCCL communicator itself reports (a scaling run proves RCCL saw N ranks)
  int n = -1;
  guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    LX_NCCL(ncclCommCount(h->comm, &n));
    return LOAMX_OK;
  });
  return n;
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
