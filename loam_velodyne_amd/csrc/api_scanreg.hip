// C-ABI: feature extraction (loamx_scanreg_*) — shim over loamx::FeatureExtractor.
#include "api_handles.h"
#include <algorithm>
#include <string>

using namespace loamx;

extern "C" {

void loamx_scanreg_default_config(loamx_scanreg_config* cfg) {
  if (!cfg) return;
  cfg->scan_period = 0.1f;
  cfg->n_feature_regions = 6;
  cfg->curvature_region = 5;
  cfg->max_corner_sharp = 2;
  cfg->max_surface_flat = 4;
  cfg->less_flat_filter_size = 0.2f;
  cfg->surface_curvature_threshold = 0.1f;
  cfg->device = 0;
  cfg->max_corner_less_sharp = 20;
  cfg->imu_history_size = 200;
}

// validation as in the reference's parameter parsing (ScanRegistration.cpp:49-138), then RegistrationParams -> FeatParams
static void apply_scanreg_config(FeatureExtractor& fx, const loamx_scanreg_config& c) {
  LX_REQUIRE(c.scan_period > 0.f, "scan_period must be positive");
  LX_REQUIRE(c.n_feature_regions >= 1, "n_feature_regions must be >= 1");
  LX_REQUIRE(c.curvature_region >= 1, "curvature_region must be >= 1");
  LX_REQUIRE(c.max_corner_sharp >= 1, "max_corner_sharp must be >= 1");
  LX_REQUIRE(c.max_surface_flat >= 1, "max_surface_flat must be >= 1");
  LX_REQUIRE(c.less_flat_filter_size >= 0.001f, "less_flat_filter_size must be >= 0.001");
  LX_REQUIRE(c.surface_curvature_threshold >= 0.001f, "surface_curvature_threshold must be >= 0.001");
  const int less_sharp = c.max_corner_less_sharp == 0 ? 10 * c.max_corner_sharp : c.max_corner_less_sharp;   // BasicScanRegistration.cpp:22
  LX_REQUIRE(less_sharp >= c.max_corner_sharp, "max_corner_less_sharp must be >= max_corner_sharp");          // ScanRegistration.cpp:100-109
  LX_REQUIRE(c.imu_history_size >= 1 && c.imu_history_size <= 4096, "imu_history_size must be in [1, 4096]");    // :59-66
  FeatParams& p = fx.params;
  p.scan_period = c.scan_period;
  p.n_regions = c.n_feature_regions;
  p.curv_region = c.curvature_region;
  p.max_sharp = c.max_corner_sharp;
  p.max_less_sharp = less_sharp;
  p.max_flat = c.max_surface_flat;
  p.less_flat_leaf = c.less_flat_filter_size;
  p.curv_thr = c.surface_curvature_threshold;
  fx.imu_history_size = std::max(fx.imu_history_size, std::max(200, c.imu_history_size));   // ensureCapacity only grows (CircularBuffer.h:53-70)
}

loamx_scanreg* loamx_scanreg_create(const loamx_scanreg_config* cfg) {
  loamx_scanreg* h = nullptr;
  guard([&]() {
    loamx_scanreg_config c;
    if (cfg) c = *cfg; else loamx_scanreg_default_config(&c);
    h = new loamx_scanreg(c.device);
    try { apply_scanreg_config(h->fx, c); } catch (...) { delete h; h = nullptr; throw; }
    return LOAMX_OK;
  });
  return h;
}

void loamx_scanreg_destroy(loamx_scanreg* h) { delete h; }

int loamx_scanreg_configure(loamx_scanreg* h, const loamx_scanreg_config* cfg) {
  return guard([&]() {
    LX_REQUIRE(h && cfg, "NULL argument");
    LX_REQUIRE(cfg->device == h->fx.device(), "configure() cannot move a handle to another device");
    apply_scanreg_config(h->fx, *cfg);
    return LOAMX_OK;
  });
}

int loamx_scanreg_process(loamx_scanreg* h, const loamx_cloud* cloud, const uint32_t* ring_size, uint32_t n_rings,
                          loamx_cloud* sharp, loamx_cloud* less_sharp, loamx_cloud* flat, loamx_cloud* less_flat) {
  return guard([&]() {
    LX_REQUIRE(h && cloud && ring_size && n_rings > 0, "NULL / empty argument");
    const uint32_t* rs[1] = {ring_size};
    h->fx.begin_sweep();   // reset(scanTime) / updateIMUTransform() of processScanlines (identities without IMU data)
    h->fx.upload(1, cloud, rs, &n_rings, /*allow_direct=*/true);   // (download() below waits for the sweep: the cloud is free when this returns)
    h->fx.run_async();
    return h->fx.download(0, sharp, less_sharp, flat, less_flat);   // (one wait, behind the launch that packs the clouds into pinned memory)
  });
}

int loamx_scanreg_process_linked(loamx_scanreg* h, const loamx_cloud* cloud, const uint32_t* ring_size, uint32_t n_rings) {
  return guard([&]() {
    LX_REQUIRE(h && cloud && ring_size && n_rings > 0, "NULL / empty argument");
    const uint32_t* rs[1] = {ring_size};
    h->fx.begin_sweep();
    // a packed cloud in runtime-pinned memory is NOT copied out when this returns: a kernel of the extraction fetches it from where it
    // lies, so it must stay unchanged until loamx_odom_process_linked has returned for this sweep (include/loamx.h); any other cloud
    // has been packed into the handle's own staging block by now
    h->fx.upload(1, cloud, rs, &n_rings, /*allow_direct=*/true);
    // no wait: loamx_odom_process_linked waits for (and checks) the extraction — in two steps, the less-flat cloud behind the odometry's
    // first launches (LOAMX_LINK_NO_SPLIT=1: in one step, as before round 6; same results — tests/test_gpu_linked.py)
    const bool no_split = getenv("LOAMX_LINK_NO_SPLIT") != nullptr;   // (read per call: a test toggles it)
    h->fx.run_async(/*mirror_offsets=*/!no_split);
    return LOAMX_OK;
  });
}

// MultiScanMapper presets (include/loam_velodyne/MultiScanRegistration.h:60-75)
int loamx_multiscan_mapper_preset(const char* sensor, loamx_multiscan_mapper* out) {
  return guard([&]() {
    LX_REQUIRE(sensor && out, "NULL argument");
    const std::string s(sensor);
    if (s == "VLP-16") *out = {-15.f, 15.f, 16};
    else if (s == "HDL-32") *out = {-30.67f, 10.67f, 32};
    else if (s == "HDL-64E") *out = {-24.9f, 2.f, 64};
    else throw Error(LOAMX_E_INVALID, "unknown lidar model (VLP-16, HDL-32, HDL-64E)");   // MultiScanRegistration.cpp:100-104
    return LOAMX_OK;
  });
}

int loamx_scanreg_process_raw(loamx_scanreg* h, const loamx_multiscan_mapper* mapper, const void* raw_xyz, uint32_t count, uint32_t stride,
                              loamx_cloud* full, uint32_t* ring_size, loamx_cloud* sharp, loamx_cloud* less_sharp, loamx_cloud* flat,
                              loamx_cloud* less_flat) {
  return guard([&]() {
    LX_REQUIRE(h && mapper, "NULL argument");
    // the reference validates minVerticalAngle/maxVerticalAngle/nScanRings the same way (MultiScanRegistration.cpp:107-127)
    LX_REQUIRE(mapper->upper_bound_deg > mapper->lower_bound_deg, "invalid vertical range (upper <= lower)");
    LX_REQUIRE(mapper->n_scan_rings >= 2, "invalid number of scan rings (n < 2)");
    h->fx.upload_raw(raw_xyz, count, stride, mapper->lower_bound_deg, mapper->upper_bound_deg, mapper->n_scan_rings);
    h->fx.run_async();
    int rc = h->fx.download_cloud(0, full, ring_size);
    const int rc2 = h->fx.download(0, sharp, less_sharp, flat, less_flat);
    return rc != LOAMX_OK ? rc : rc2;
  });
}

// updateIMUData(acc, newState) — BasicScanRegistration.cpp:82-98; acc = local acceleration with gravity removed and axes
// remapped as ScanRegistration::handleIMUMessage does (src/lib/ScanRegistration.cpp:164-184)
int loamx_scanreg_update_imu(loamx_scanreg* h, double stamp_sec, float roll, float pitch, float yaw, const float acc_xyz[3]) {
  return guard([&]() {
    LX_REQUIRE(h && acc_xyz, "NULL argument");
    h->fx.update_imu_data(stamp_sec, roll, pitch, yaw, acc_xyz);
    return LOAMX_OK;
  });
}
int loamx_scanreg_set_time(loamx_scanreg* h, double scan_time_sec) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    h->fx.set_scan_time(scan_time_sec);
    return LOAMX_OK;
  });
}
int loamx_scanreg_get_imu_trans(loamx_scanreg* h, float imu_trans[12]) {
  return guard([&]() {
    LX_REQUIRE(h && imu_trans, "NULL argument");
    for (int k = 0; k < 12; k++) imu_trans[k] = h->fx.imu_trans()[k];
    return LOAMX_OK;
  });
}

}  // extern "C"
