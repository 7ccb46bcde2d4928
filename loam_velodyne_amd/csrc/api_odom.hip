// C-ABI: sweep-to-sweep odometry (loamx_odom_*) — shim over loamx::Odometry.
#include "api_handles.h"

using namespace loamx;

extern "C" {

void loamx_odom_default_config(loamx_odom_config* cfg) {
  if (!cfg) return;
  cfg->scan_period = 0.1f;
  cfg->max_iterations = 25;
  cfg->delta_t_abort = 0.1f;
  cfg->delta_r_abort = 0.1f;
  cfg->device = 0;
}

loamx_odom* loamx_odom_create(const loamx_odom_config* cfg) {
  loamx_odom* h = nullptr;
  guard([&]() {
    loamx_odom_config c;
    if (cfg) c = *cfg; else loamx_odom_default_config(&c);
    // same validation as LaserOdometry::setup (LaserOdometry.cpp:70-138)
    LX_REQUIRE(c.scan_period > 0.f, "scan_period must be positive");
    LX_REQUIRE(c.max_iterations >= 1 && c.max_iterations <= 255, "max_iterations must be in [1, 255]");   // (k_odom_lm tags its exchange records with iteration + 1 in 8 bits)
    LX_REQUIRE(c.delta_t_abort > 0.f && c.delta_r_abort > 0.f, "abort thresholds must be positive");
    h = new loamx_odom(c.device);
    h->od.params.scan_period = c.scan_period;
    h->od.params.max_iterations = c.max_iterations;
    h->od.params.delta_t_abort = c.delta_t_abort;
    h->od.params.delta_r_abort = c.delta_r_abort;
    return LOAMX_OK;
  });
  return h;
}
void loamx_odom_destroy(loamx_odom* h) { delete h; }

int loamx_odom_update_imu(loamx_odom* h, const float imu_trans[12]) {
  return guard([&]() {
    LX_REQUIRE(h && imu_trans, "NULL argument");
    h->od.update_imu(0, imu_trans);
    return LOAMX_OK;
  });
}
int loamx_odom_process(loamx_odom* h, const loamx_cloud* sharp, const loamx_cloud* less_sharp, const loamx_cloud* flat,
                       const loamx_cloud* less_flat) {
  return guard([&]() {
    LX_REQUIRE(h && sharp && less_sharp && flat && less_flat, "NULL argument");
    return h->od.process_host(sharp, less_sharp, flat, less_flat);
  });
}
int loamx_odom_process_linked(loamx_odom* h, loamx_scanreg* sr) {
  return guard([&]() {
    LX_REQUIRE(h && sr, "NULL argument");
    LX_REQUIRE(sr->fx.device() == h->od.device(), "linked handles must live on one device");
    const float4* p[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t n[4] = {0, 0, 0, 0};
    const float4* full = sr->fx.d_cloud() + sr->fx.point_base(0);
    const uint32_t n_full = sr->fx.point_base(1) - sr->fx.point_base(0);
    if (sr->fx.split_results()) {
      // the iterations start as soon as the sharp / less-sharp / flat clouds are compacted; the less-flat cloud (its per-ring voxel
      // grid runs on the extraction's stream meanwhile) is fetched when the sweep's tail is about to be enqueued
      sr->fx.device_results_front(0, p, n);
      return h->od.process_linked(p, n, full, n_full, [sr](const float4*& q, uint32_t& c) { sr->fx.device_results_lf(0, q, c); });
    }
    sr->fx.device_results(0, p, n);   // (waits for the extraction; the clouds stay where they are)
    return h->od.process_linked(p, n, full, n_full);
  });
}
int loamx_odom_link_wait(loamx_odom* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    if (h->od.link_ready()) LX_HIP(hipEventSynchronize(h->od.link_ready()));
    return LOAMX_OK;
  });
}
int loamx_odom_get_transform(loamx_odom* h, float t[6]) {
  return guard([&]() { LX_REQUIRE(h && t, "NULL argument"); h->od.stream_state(0).transform.get(t); return LOAMX_OK; });
}
int loamx_odom_get_transform_sum(loamx_odom* h, float t[6]) {
  return guard([&]() { LX_REQUIRE(h && t, "NULL argument"); h->od.stream_state(0).transform_sum.get(t); return LOAMX_OK; });
}
int loamx_odom_set_transform(loamx_odom* h, const float t[6]) {
  return guard([&]() { LX_REQUIRE(h && t, "NULL argument"); h->od.stream_state(0).transform.set(t); return LOAMX_OK; });
}
int loamx_odom_set_transform_sum(loamx_odom* h, const float t[6]) {
  return guard([&]() { LX_REQUIRE(h && t, "NULL argument"); h->od.stream_state(0).transform_sum.set(t); return LOAMX_OK; });
}
int loamx_odom_get_last_clouds(loamx_odom* h, loamx_cloud* last_corner, loamx_cloud* last_surf) {
  return guard([&]() { LX_REQUIRE(h, "NULL handle"); return h->od.get_last_clouds(0, last_corner, last_surf); });
}
int loamx_odom_transform_to_end(loamx_odom* h, loamx_cloud* cloud) {
  return guard([&]() { LX_REQUIRE(h && cloud, "NULL argument"); return h->od.transform_to_end_host(0, cloud); });
}
int loamx_odom_get_stats(loamx_odom* h, int stats[4]) {
  return guard([&]() {
    LX_REQUIRE(h && stats, "NULL argument");
    OdomStats s = h->od.stream_state(0).stats;
    stats[0] = s.iterations; stats[1] = s.sel; stats[2] = s.frame; stats[3] = s.degenerate;
    return LOAMX_OK;
  });
}

}  // extern "C"
