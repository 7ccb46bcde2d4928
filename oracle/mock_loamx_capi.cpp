// ORACLE — TEST INFRASTRUCTURE ONLY.  A TEST DOUBLE of the libloamx C-ABI (include/loamx.h) backed by the CPU oracle, under a
// DIFFERENT library name (libloamx_oracle_mock.so).  It exists for one purpose: to exercise the host-side glue that sits ABOVE the
// C-ABI — loam_velodyne_amd/adapter/loamx_adapter.h, the swapped MultiScanRegistration unit and the reference's own node sources
// compiled against them — on a machine without a GPU (tests/test_four_nodes.py::test_adapter_glue_over_the_oracle_mock).
// It is never linked into, loaded by or shipped with the product: the product library has no CPU path (include/loamx.h), and no
// parity or performance claim rests on this file.  Only the entry points the adapter's node flow reaches are provided.
#include <cstring>
#include <string>
#include "../include/loamx.h"
#include "oracle_mapping.hpp"
#include "oracle_features.hpp"
#include "oracle_ingest.hpp"
#include "oracle_maintenance.hpp"

using namespace loam_oracle;

namespace {
Cloud read_cloud(const loamx_cloud* c) {
  Cloud out;
  if (!c) return out;
  out.resize(c->count);
  for (uint32_t i = 0; i < c->count; i++) {
    const char* r = (const char*)c->data + (size_t)i * c->stride;
    const float* f = (const float*)r;
    out[i] = {f[0], f[1], f[2], *(const float*)(r + c->intensity_offset)};
  }
  return out;
}
// writes as many points as fit; count returns the size needed, LOAMX_E_CAPACITY when it did not fit
int write_cloud(const Cloud& src, loamx_cloud* c) {
  if (!c) return LOAMX_OK;
  const uint32_t cap = c->count;
  c->count = (uint32_t)src.size();
  if (src.size() > cap) return LOAMX_E_CAPACITY;
  for (size_t i = 0; i < src.size(); i++) {
    char* r = (char*)c->data + i * c->stride;
    float* f = (float*)r;
    f[0] = src[i].x; f[1] = src[i].y; f[2] = src[i].z;
    if (c->intensity_offset != 12) f[3] = 1.f;
    *(float*)(r + c->intensity_offset) = src[i].i;
  }
  return LOAMX_OK;
}
void twist_to(const Twist& t, float* o) { o[0] = t.rot_x.rad(); o[1] = t.rot_y.rad(); o[2] = t.rot_z.rad(); o[3] = t.pos.x; o[4] = t.pos.y; o[5] = t.pos.z; }
}  // namespace

struct loamx_scanreg { ScanRegistration s; double next_time = 0; };
struct loamx_odom { LaserOdometry o; };
struct loamx_map { LaserMapping m; };
struct loamx_tm { TransformMaintenance t; };

extern "C" {

const char* loamx_last_error(void) { return "oracle-backed test double"; }
int loamx_device_count(void) { return 0; }
int loamx_abi_version(void) { return LOAMX_ABI_VERSION; }
const char* loamx_build_info(void) { return "abi=5;diag=0;rccl=0;roctx=0;mock=1"; }

// ---- scan registration
void loamx_scanreg_default_config(loamx_scanreg_config* c) { *c = loamx_scanreg_config{0.1f, 6, 5, 2, 4, 0.2f, 0.1f, 0, 20, 200}; }
int loamx_scanreg_configure(loamx_scanreg* h, const loamx_scanreg_config* c) {
  auto& k = h->s.cfg;
  k.scanPeriod = c->scan_period; k.nFeatureRegions = c->n_feature_regions; k.curvatureRegion = c->curvature_region;
  k.maxCornerSharp = c->max_corner_sharp; k.maxSurfaceFlat = c->max_surface_flat;
  k.maxCornerLessSharp = c->max_corner_less_sharp ? c->max_corner_less_sharp : 10 * c->max_corner_sharp;
  k.imuHistorySize = c->imu_history_size;
  k.lessFlatFilterSize = c->less_flat_filter_size; k.surfaceCurvatureThreshold = c->surface_curvature_threshold;
  return LOAMX_OK;
}
loamx_scanreg* loamx_scanreg_create(const loamx_scanreg_config* c) {
  auto* h = new loamx_scanreg();
  loamx_scanreg_configure(h, c);
  return h;
}
void loamx_scanreg_destroy(loamx_scanreg* h) { delete h; }
static int scanreg_outputs(loamx_scanreg* h, loamx_cloud* a, loamx_cloud* b, loamx_cloud* c, loamx_cloud* d) {
  int rc = write_cloud(h->s.cornerSharp, a);
  if (rc == LOAMX_OK) rc = write_cloud(h->s.cornerLessSharp, b);
  if (rc == LOAMX_OK) rc = write_cloud(h->s.surfFlat, c);
  if (rc == LOAMX_OK) rc = write_cloud(h->s.surfLessFlat, d);
  return rc;
}
int loamx_scanreg_process(loamx_scanreg* h, const loamx_cloud* cloud, const uint32_t* ring_size, uint32_t n_rings, loamx_cloud* a, loamx_cloud* b,
                          loamx_cloud* c, loamx_cloud* d) {
  const Cloud all = read_cloud(cloud);
  std::vector<Cloud> rings(n_rings);
  size_t off = 0;
  for (uint32_t r = 0; r < n_rings; r++) { rings[r].assign(all.begin() + off, all.begin() + off + ring_size[r]); off += ring_size[r]; }
  h->s.process_scanlines_at(h->next_time, rings);
  return scanreg_outputs(h, a, b, c, d);
}
int loamx_multiscan_mapper_preset(const char* sensor, loamx_multiscan_mapper* out) {
  const std::string s(sensor);
  if (s == "VLP-16") *out = {-15.f, 15.f, 16};
  else if (s == "HDL-32") *out = {-30.67f, 10.67f, 32};
  else if (s == "HDL-64E") *out = {-24.9f, 2.f, 64};
  else return LOAMX_E_INVALID;
  return LOAMX_OK;
}
int loamx_scanreg_process_raw(loamx_scanreg* h, const loamx_multiscan_mapper* mp, const void* raw_xyz, uint32_t count, uint32_t stride, loamx_cloud* full,
                              uint32_t* ring_size, loamx_cloud* a, loamx_cloud* b, loamx_cloud* c, loamx_cloud* d) {
  std::vector<float> raw(3 * (size_t)count);
  for (uint32_t i = 0; i < count; i++) std::memcpy(&raw[3 * (size_t)i], (const char*)raw_xyz + (size_t)i * stride, 12);
  MultiScanMapper m;
  m.set(mp->lower_bound_deg, mp->upper_bound_deg, (uint16_t)mp->n_scan_rings);
  std::vector<Cloud> scans = bin_sweep(raw.data(), count, m, h->s.cfg.scanPeriod, &h->s);
  if (ring_size) for (uint32_t r = 0; r < mp->n_scan_rings; r++) ring_size[r] = (uint32_t)scans[r].size();
  h->s.process_scanlines_at(h->next_time, scans);
  int rc = write_cloud(h->s.laserCloud, full);
  return rc == LOAMX_OK ? scanreg_outputs(h, a, b, c, d) : rc;
}
int loamx_scanreg_update_imu(loamx_scanreg* h, double stamp, float roll, float pitch, float yaw, const float acc[3]) {
  ScanRegistration::IMUState st;
  st.stamp = stamp; st.roll = Angle(roll); st.pitch = Angle(pitch); st.yaw = Angle(yaw);
  st.acceleration = {acc[0], acc[1], acc[2]};
  h->s.update_imu_data({acc[0], acc[1], acc[2]}, st);
  return LOAMX_OK;
}
int loamx_scanreg_set_time(loamx_scanreg* h, double t) { h->next_time = t; return LOAMX_OK; }
int loamx_scanreg_get_imu_trans(loamx_scanreg* h, float out[12]) { for (int k = 0; k < 12; k++) out[k] = h->s.imuTrans[k]; return LOAMX_OK; }

// ---- odometry
void loamx_odom_default_config(loamx_odom_config* c) { *c = loamx_odom_config{0.1f, 25, 0.1f, 0.1f, 0}; }
loamx_odom* loamx_odom_create(const loamx_odom_config* c) {
  auto* h = new loamx_odom();
  h->o.scanPeriod = c->scan_period; h->o.maxIterations = c->max_iterations; h->o.deltaTAbort = c->delta_t_abort; h->o.deltaRAbort = c->delta_r_abort;
  return h;
}
void loamx_odom_destroy(loamx_odom* h) { delete h; }
int loamx_odom_update_imu(loamx_odom* h, const float t12[12]) { h->o.update_imu(t12); return LOAMX_OK; }
int loamx_odom_process(loamx_odom* h, const loamx_cloud* a, const loamx_cloud* b, const loamx_cloud* c, const loamx_cloud* d) {
  h->o.cornerSharp = read_cloud(a); h->o.cornerLessSharp = read_cloud(b); h->o.surfFlat = read_cloud(c); h->o.surfLessFlat = read_cloud(d);
  const bool first = !h->o.systemInited;
  h->o.process();
  return first ? LOAMX_SKIPPED : LOAMX_OK;
}
int loamx_odom_get_transform(loamx_odom* h, float t[6]) { twist_to(h->o.transform, t); return LOAMX_OK; }
int loamx_odom_get_transform_sum(loamx_odom* h, float t[6]) { twist_to(h->o.transformSum, t); return LOAMX_OK; }
int loamx_odom_get_last_clouds(loamx_odom* h, loamx_cloud* lc, loamx_cloud* ls) {
  int rc = write_cloud(h->o.lastCorner, lc);
  return rc == LOAMX_OK ? write_cloud(h->o.lastSurf, ls) : rc;
}
int loamx_odom_set_transform(loamx_odom* h, const float t[6]) {
  h->o.transform.rot_x = t[0]; h->o.transform.rot_y = t[1]; h->o.transform.rot_z = t[2]; h->o.transform.pos = {t[3], t[4], t[5]};
  return LOAMX_OK;
}
int loamx_odom_set_transform_sum(loamx_odom* h, const float t[6]) {
  h->o.transformSum.rot_x = t[0]; h->o.transformSum.rot_y = t[1]; h->o.transformSum.rot_z = t[2]; h->o.transformSum.pos = {t[3], t[4], t[5]};
  return LOAMX_OK;
}
int loamx_odom_get_stats(loamx_odom* h, int s[4]) {
  s[0] = h->o.lastIterCount; s[1] = h->o.lastSelNum; s[2] = (int)h->o.frameCount; s[3] = 0;   // (the oracle keeps no degeneracy flag for the odometry)
  return LOAMX_OK;
}
int loamx_odom_transform_to_end(loamx_odom* h, loamx_cloud* cloud) {
  Cloud c = read_cloud(cloud);
  h->o.transform_to_end(c);
  return write_cloud(c, cloud);
}

// ---- mapping
void loamx_map_default_config(loamx_map_config* c) { *c = loamx_map_config{0.1f, 10, 0.05f, 0.05f, 0.2f, 0.4f, 0.6f, 0}; }
loamx_map* loamx_map_create(const loamx_map_config* c) {
  auto* h = new loamx_map();
  h->m.scanPeriod = c->scan_period; h->m.maxIterations = c->max_iterations; h->m.deltaTAbort = c->delta_t_abort; h->m.deltaRAbort = c->delta_r_abort;
  h->m.cornerLeaf = c->corner_filter_size; h->m.surfLeaf = c->surf_filter_size;
  return h;
}
void loamx_map_destroy(loamx_map* h) { delete h; }
int loamx_map_update_odometry(loamx_map* h, const float t6[6]) { h->m.update_odometry(t6); return LOAMX_OK; }
int loamx_map_process(loamx_map* h, const loamx_cloud* lc, const loamx_cloud* ls, loamx_cloud* full) {
  h->m.cornerLast = read_cloud(lc); h->m.surfLast = read_cloud(ls); h->m.fullRes = read_cloud(full);
  if (!h->m.process()) return LOAMX_SKIPPED;
  return write_cloud(h->m.fullRes, full);
}
int loamx_map_get_transform(loamx_map* h, int which, float t[6]) {
  const Twist* w[4] = {&h->m.transformAftMapped, &h->m.transformBefMapped, &h->m.transformTobeMapped, &h->m.transformSum};
  twist_to(*w[which], t);
  return LOAMX_OK;
}
int loamx_map_set_transform(loamx_map* h, int which, const float t[6]) {
  Twist* w[4] = {&h->m.transformAftMapped, &h->m.transformBefMapped, &h->m.transformTobeMapped, &h->m.transformSum};
  w[which]->rot_x = t[0]; w[which]->rot_y = t[1]; w[which]->rot_z = t[2]; w[which]->pos = {t[3], t[4], t[5]};
  return LOAMX_OK;
}
int loamx_map_get_stats(loamx_map* h, int s[8]) {
  const MappingStats& m = h->m.stats;
  s[0] = m.iterations; s[1] = m.lastSelNum; s[2] = m.cornerDS; s[3] = m.surfDS; s[4] = m.cornerFromMap; s[5] = m.surfFromMap; s[6] = m.degenerate; s[7] = m.optimized;
  return LOAMX_OK;
}
int loamx_map_load_cubes(loamx_map* h, const loamx_cloud* corner, const loamx_cloud* surf) {
  for (int t = 0; t < 2; t++) {
    const Cloud pts = read_cloud(t == 0 ? corner : surf);
    auto& arr = t == 0 ? h->m.cornerArray : h->m.surfArray;
    for (const Pt& q : pts) {
      const int I = LaserMapping::cube_of(q.x, h->m.cenW), J = LaserMapping::cube_of(q.y, h->m.cenH), K = LaserMapping::cube_of(q.z, h->m.cenD);
      if (I >= 0 && I < LaserMapping::W && J >= 0 && J < LaserMapping::H && K >= 0 && K < LaserMapping::D) arr[LaserMapping::to_index(I, J, K)].push_back(q);
    }
  }
  return LOAMX_OK;
}
int loamx_map_get_cubes(loamx_map* h, int which, loamx_cloud* out) {
  Cloud all;
  for (const Cloud& c : (which == 0 ? h->m.cornerArray : h->m.surfArray)) all.insert(all.end(), c.begin(), c.end());
  return write_cloud(all, out);
}
int loamx_map_update_imu(loamx_map* h, double stamp, float roll, float pitch) { h->m.update_imu(stamp, roll, pitch); return LOAMX_OK; }
int loamx_map_set_time(loamx_map* h, double t) { h->m.laserOdometryTime = t; return LOAMX_OK; }
int loamx_map_has_fresh_map(loamx_map* h) { return h->m.downsizedMapCreated ? 1 : 0; }
int loamx_map_get_surround(loamx_map* h, loamx_cloud* out) { return write_cloud(h->m.surroundDS, out); }

// ---- pose fusion
loamx_tm* loamx_tm_create(void) { return new loamx_tm(); }
void loamx_tm_destroy(loamx_tm* h) { delete h; }
int loamx_tm_update_odometry(loamx_tm* h, const float s[6]) { h->t.update_odometry(s[0], s[1], s[2], s[3], s[4], s[5]); return LOAMX_OK; }
int loamx_tm_update_mapping_transform(loamx_tm* h, const float aft[6], const float bef[6]) {
  double a[6], b[6];
  for (int k = 0; k < 6; k++) { a[k] = aft[k]; b[k] = bef[k]; }
  h->t.update_mapping_transform(a, b);
  return LOAMX_OK;
}
int loamx_tm_associate_to_map(loamx_tm* h) { h->t.transform_associate_to_map(); return LOAMX_OK; }
int loamx_tm_get_mapped(loamx_tm* h, float out[6]) { for (int k = 0; k < 6; k++) out[k] = h->t.transformMapped[k]; return LOAMX_OK; }
int loamx_wire_pose_to_quat(const float rot[3], double q[4]) { wire_pose_to_quat(rot, q); return LOAMX_OK; }
int loamx_wire_quat_to_pose(const double q[4], float rot[3]) { wire_quat_to_pose(q, rot); return LOAMX_OK; }

}  // extern "C"
