// C-ABI: batched-sweep registration (loamx_batch_*) — thin shim over loamx::Registrar.
#include "registration.hpp"

using namespace loamx;

struct loamx_batch {
  Registrar reg;
  loamx_batch(int device, uint32_t max_sweeps) : reg(device, max_sweeps) {}
};

static void apply_cfg(Registrar& r, const loamx_map_config* cfg) {
  r.params.max_iterations = cfg->max_iterations;
  r.params.delta_t_abort = cfg->delta_t_abort;
  r.params.delta_r_abort = cfg->delta_r_abort;
  r.params.corner_leaf = cfg->corner_filter_size;
  r.params.surf_leaf = cfg->surf_filter_size;
}

extern "C" {

void loamx_map_default_config(loamx_map_config* cfg) {
  if (!cfg) return;
  cfg->scan_period = 0.1f;
  cfg->max_iterations = 10;
  cfg->delta_t_abort = 0.05f;
  cfg->delta_r_abort = 0.05f;
  cfg->corner_filter_size = 0.2f;
  cfg->surf_filter_size = 0.4f;
  cfg->map_filter_size = 0.0f;
  cfg->device = 0;
}

loamx_batch* loamx_batch_create(const loamx_map_config* cfg, uint32_t max_sweeps) {
  loamx_batch* h = nullptr;
  guard([&]() {
    loamx_map_config c;
    if (cfg) c = *cfg; else loamx_map_default_config(&c);
    LX_REQUIRE(max_sweeps >= 1 && max_sweeps <= 1024, "max_sweeps must be in [1, 1024]");
    LX_REQUIRE(c.max_iterations >= 0 && c.max_iterations <= 64, "max_iterations must be in [0, 64]");
    LX_REQUIRE(c.corner_filter_size > 0.f && c.surf_filter_size > 0.f, "filter sizes must be positive");
    h = new loamx_batch(c.device, max_sweeps);
    apply_cfg(h->reg, &c);
    return LOAMX_OK;
  });
  return h;
}
void loamx_batch_destroy(loamx_batch* h) { delete h; }

int loamx_batch_set_frozen(loamx_batch* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.set_submap_host(corner_map, surf_map);
    return LOAMX_OK;
  });
}
int loamx_batch_set_frozen_device(loamx_batch* h, const void* d_corner, uint32_t nc, const void* d_surf, uint32_t ns) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    LX_REQUIRE((d_corner || nc == 0) && (d_surf || ns == 0), "NULL device pointer");
    h->reg.set_submap_device((const float4*)d_corner, nc, (const float4*)d_surf, ns);
    return LOAMX_OK;
  });
}
int loamx_batch_stage_frozen_device(loamx_batch* h, const void* d_corner, uint32_t nc, const void* d_surf, uint32_t ns, void* wait_event) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    LX_REQUIRE((d_corner || nc == 0) && (d_surf || ns == 0), "NULL device pointer");
    h->reg.stage_submap_device((const float4*)d_corner, nc, (const float4*)d_surf, ns, (hipEvent_t)wait_event);
    return LOAMX_OK;
  });
}
int loamx_batch_stage_frozen(loamx_batch* h, const loamx_cloud* corner_map, const loamx_cloud* surf_map) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.stage_submap_host(corner_map, surf_map);
    return LOAMX_OK;
  });
}
int loamx_batch_swap_frozen(loamx_batch* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    if (!h->reg.submap_staged()) return (int)LOAMX_SKIPPED;
    h->reg.swap_submap();
    return (int)LOAMX_OK;
  });
}
int loamx_batch_upload(loamx_batch* h, uint32_t n_sweeps, const loamx_cloud* corner_last, const loamx_cloud* surf_last,
                       const loamx_cloud* full_res, const float* guess6) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.upload(n_sweeps, corner_last, surf_last, full_res, guess6);
    return LOAMX_OK;
  });
}
int loamx_batch_run(loamx_batch* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.run_async();
    h->reg.sync();
    return h->reg.submap_sufficient() ? LOAMX_OK : LOAMX_SKIPPED;
  });
}
int loamx_batch_run_async(loamx_batch* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.run_async();
    return LOAMX_OK;
  });
}
int loamx_batch_sync(loamx_batch* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.sync();
    return LOAMX_OK;
  });
}
int loamx_batch_download(loamx_batch* h, float* poses6, int* stats4) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.download(poses6, stats4);
    return LOAMX_OK;
  });
}
int loamx_batch_download_full_res(loamx_batch* h, uint32_t sweep, loamx_cloud* out) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    return h->reg.download_full_res(sweep, out);
  });
}
int loamx_batch_set_timing(loamx_batch* h, int on) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.set_timing(on != 0);
    return LOAMX_OK;
  });
}
int loamx_batch_get_timing(loamx_batch* h, float ms[4], uint64_t counts[4]) {
  return guard([&]() {
    LX_REQUIRE(h && ms && counts, "NULL argument");
    h->reg.get_timing(ms, counts);
    return LOAMX_OK;
  });
}
int loamx_batch_knn_probe(loamx_batch* h, int which, const float* queries_xyz, uint32_t n, uint32_t* idx5, float* d2_5) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.knn_probe(which, queries_xyz, n, idx5, d2_5);
    return LOAMX_OK;
  });
}
int loamx_batch_qr6_probe(loamx_batch* h, const float* ata, const float* atb, uint32_t n, float* x_coop, float* x_scalar) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->reg.qr6_probe(ata, atb, n, x_coop, x_scalar);
    return LOAMX_OK;
  });
}
int loamx_batch_xrec_stress(loamx_batch* h, uint32_t pairs, uint32_t rounds, uint64_t out4[4]) {
  return guard([&]() {
    LX_REQUIRE(h && out4, "NULL argument");
    unsigned long long o[4] = {0, 0, 0, 0};
    h->reg.xrec_stress(pairs, rounds, o);
    for (int k = 0; k < 4; k++) out4[k] = (uint64_t)o[k];
    return LOAMX_OK;
  });
}
int loamx_batch_download_ds(loamx_batch* h, uint32_t sweep, loamx_cloud* corner_ds, loamx_cloud* surf_ds) {
  return guard([&]() {
    LX_REQUIRE(h && corner_ds && surf_ds, "NULL argument");
    check_cloud(corner_ds, false);
    check_cloud(surf_ds, false);
    std::vector<float4> c, s;
    h->reg.download_ds(sweep, c, s);
    const int rc = unpack_cloud(c.data(), (uint32_t)c.size(), corner_ds), rs = unpack_cloud(s.data(), (uint32_t)s.size(), surf_ds);
    return rc != LOAMX_OK ? rc : rs;
  });
}
void* loamx_batch_stream(loamx_batch* h) { return h ? (void*)h->reg.stream() : nullptr; }

}  // extern "C"
