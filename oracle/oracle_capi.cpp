// ORACLE — TEST INFRASTRUCTURE ONLY.  Flat C entry points over the oracle headers so tests/, smoke() and
// bench.py's cpu_baseline leg can drive it through ctypes.  Never linked into the product library.
#include "oracle_mapping.hpp"
#include "oracle_features.hpp"
#include "oracle_ingest.hpp"
#include "oracle_maintenance.hpp"
#include <chrono>

using namespace loam_oracle;

namespace {
Cloud to_cloud(const float* p, int n) {
  Cloud c(n);
  for (int i = 0; i < n; i++) c[i] = {p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]};
  return c;
}
int from_cloud(const Cloud& c, float* out, int cap) {
  int n = (int)c.size();
  if (out) {
    int m = std::min(n, cap);
    for (int i = 0; i < m; i++) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = c[i].i; }
  }
  return n;
}
void twist_to(const Twist& t, float* o) {
  o[0] = t.rot_x.rad(); o[1] = t.rot_y.rad(); o[2] = t.rot_z.rad(); o[3] = t.pos.x; o[4] = t.pos.y; o[5] = t.pos.z;
}
void twist_from(Twist& t, const float* o) {
  t.rot_x = o[0]; t.rot_y = o[1]; t.rot_z = o[2]; t.pos = {o[3], o[4], o[5]};
}
}  // namespace

extern "C" {

// ---- primitives ----
int orc_voxel_grid(const float* pts, int n, float leaf, float* out, int cap) {
  Cloud in = to_cloud(pts, n), o;
  voxel_grid(in, leaf, o);
  return from_cloud(o, out, cap);
}
// mode 0: kd-tree (nanoflann contract), 1: brute force
void orc_knn(const float* pts, int n, const float* q, int nq, int k, int* idx, float* d2, int mode) {
  Cloud c = to_cloud(pts, n);
  KdTree t;
  if (mode == 0) t.build(&c);
  for (int i = 0; i < nq; i++) {
    Pt p{q[4 * i], q[4 * i + 1], q[4 * i + 2], 0.f};
    if (mode == 0) t.knn(p, k, idx + (size_t)i * k, d2 + (size_t)i * k);
    else knn_brute(c, p, k, idx + (size_t)i * k, d2 + (size_t)i * k);
  }
}
void orc_eig3(const float* A, float* w, float* V) { eig_sym_jacobi<3>(A, w, V); }
void orc_eig6(const float* A, float* w, float* V) { eig_sym_jacobi<6>(A, w, V); }
void orc_qr53(const float* A, const float* b, float* x) { colpiv_qr_solve<5, 3>(A, b, x); }
void orc_qr66(const float* A, const float* b, float* x) { colpiv_qr_solve<6, 6>(A, b, x); }
int orc_inv6(const float* A, float* inv) { return inverse_lu<6>(A, inv) ? 1 : 0; }
int orc_degeneracy(const float* AtA, float thr, float* P) { return degeneracy_projector(AtA, thr, P) ? 1 : 0; }
void orc_rotate_zxy(float* p3, float rz, float rx, float ry) {
  Vec3 v{p3[0], p3[1], p3[2]};
  rotateZXY(v, Angle(rz), Angle(rx), Angle(ry));
  p3[0] = v.x; p3[1] = v.y; p3[2] = v.z;
}

// ---- feature extraction ----
void* orc_scanreg_create() { return new ScanRegistration(); }
void orc_scanreg_destroy(void* h) { delete (ScanRegistration*)h; }
void orc_scanreg_config(void* h, float scanPeriod, int nFeatureRegions, int curvatureRegion, int maxCornerSharp,
                        int maxSurfaceFlat, float lessFlatFilterSize, float surfaceCurvatureThreshold) {
  auto& c = ((ScanRegistration*)h)->cfg;
  c.scanPeriod = scanPeriod; c.nFeatureRegions = nFeatureRegions; c.curvatureRegion = curvatureRegion;
  c.maxCornerSharp = maxCornerSharp; c.maxCornerLessSharp = 10 * maxCornerSharp; c.maxSurfaceFlat = maxSurfaceFlat;
  c.lessFlatFilterSize = lessFlatFilterSize; c.surfaceCurvatureThreshold = surfaceCurvatureThreshold;
}
// maxCornerLessSharp on its own (ScanRegistration.cpp:100-109); imuHistorySize (:59-66) — the history never shrinks below 200
void orc_scanreg_config2(void* h, int maxCornerLessSharp, int imuHistorySize) {
  auto* s = (ScanRegistration*)h;
  s->cfg.maxCornerLessSharp = maxCornerLessSharp;
  s->cfg.imuHistorySize = imuHistorySize;
}
// pts: all rings concatenated; ring_sizes[n_rings]
void orc_scanreg_process(void* h, const float* pts, const int* ring_sizes, int n_rings) {
  std::vector<Cloud> rings(n_rings);
  size_t off = 0;
  for (int r = 0; r < n_rings; r++) {
    rings[r] = to_cloud(pts + 4 * off, ring_sizes[r]);
    off += ring_sizes[r];
  }
  ((ScanRegistration*)h)->process_scanlines(rings);
}
// which: 0 full, 1 sharp, 2 lessSharp, 3 flat, 4 lessFlat
int orc_scanreg_get(void* h, int which, float* out, int cap) {
  auto* s = (ScanRegistration*)h;
  const Cloud* c[5] = {&s->laserCloud, &s->cornerSharp, &s->cornerLessSharp, &s->surfFlat, &s->surfLessFlat};
  return from_cloud(*c[which], out, cap);
}

// ---- raw-sweep ingestion (MultiScanRegistration::process) ----
// raw: n x (x, y, z) in sensor axes, firing order.  out: binned points (rings concatenated, 4 floats each, cap points),
// ring_sizes[n_rings].  Returns the number of points kept.
int orc_multiscan_bin(const float* raw, int n, float lower_deg, float upper_deg, int n_rings, float scan_period, float* out, int cap,
                      int* ring_sizes) {
  MultiScanMapper m;
  m.set(lower_deg, upper_deg, (uint16_t)n_rings);
  std::vector<Cloud> scans = bin_sweep(raw, (size_t)n, m, scan_period);
  int total = 0;
  for (int r = 0; r < n_rings; r++) {
    ring_sizes[r] = (int)scans[r].size();
    for (const Pt& p : scans[r]) {
      if (total < cap) { out[4 * total] = p.x; out[4 * total + 1] = p.y; out[4 * total + 2] = p.z; out[4 * total + 3] = p.i; }
      total++;
    }
  }
  return total;
}

// ---- scan registration with IMU data ----
// updateIMUData: stamp (s), roll / pitch / yaw, acceleration (local frame, gravity already removed as ScanRegistration.cpp:171-174 does)
void orc_scanreg_update_imu(void* h, double stamp, float roll, float pitch, float yaw, float ax, float ay, float az) {
  auto* s = (ScanRegistration*)h;
  ScanRegistration::IMUState st;
  st.stamp = stamp; st.roll = Angle(roll); st.pitch = Angle(pitch); st.yaw = Angle(yaw);
  st.acceleration = {ax, ay, az};
  s->update_imu_data({ax, ay, az}, st);
}
// MultiScanRegistration::process(laserCloudIn, scanTime) on the handle's IMU state; returns points kept
int orc_scanreg_process_raw(void* h, const float* raw, int n, double scan_time, float lower_deg, float upper_deg, int n_rings, int* ring_sizes) {
  auto* s = (ScanRegistration*)h;
  MultiScanMapper m;
  m.set(lower_deg, upper_deg, (uint16_t)n_rings);
  std::vector<Cloud> scans = bin_sweep(raw, (size_t)n, m, s->cfg.scanPeriod, s);
  int total = 0;
  for (int r = 0; r < n_rings; r++) { ring_sizes[r] = (int)scans[r].size(); total += ring_sizes[r]; }
  s->process_scanlines_at(scan_time, scans);
  return total;
}
void orc_scanreg_get_imu_trans(void* h, float* out12) {
  for (int k = 0; k < 12; k++) out12[k] = ((ScanRegistration*)h)->imuTrans[k];
}

// ---- math primitives as the oracle restates them (pinned against the reference's own headers in tests/) ----
void orc_angle(float rad, int negate, float add, float* out3) {
  Angle a(rad);
  if (add != 0.f) a += add;
  const Angle b = negate ? -a : a;
  out3[0] = b.rad(); out3[1] = b.cos(); out3[2] = b.sin();
}
// which: 0 rotateZXY, 1 rotateYXZ, 2 rotX, 3 rotY, 4 rotZ
void orc_rotate(int which, float* p3, float a0, float a1, float a2) {
  Vec3 v{p3[0], p3[1], p3[2]};
  switch (which) {
    case 0: rotateZXY(v, Angle(a0), Angle(a1), Angle(a2)); break;
    case 1: rotateYXZ(v, Angle(a0), Angle(a1), Angle(a2)); break;
    case 2: rotX(v, Angle(a0)); break;
    case 3: rotY(v, Angle(a0)); break;
    default: rotZ(v, Angle(a0)); break;
  }
  p3[0] = v.x; p3[1] = v.y; p3[2] = v.z;
}

// ---- transform maintenance (BasicTransformMaintenance) + wire conversions ----
void orc_tm_associate(const float* sum6, const float* bef6, const float* aft6, float* mapped6) {
  TransformMaintenance t;
  t.update_odometry(sum6[0], sum6[1], sum6[2], sum6[3], sum6[4], sum6[5]);
  double a[6], b[6];
  for (int k = 0; k < 6; k++) { a[k] = aft6[k]; b[k] = bef6[k]; }
  t.update_mapping_transform(a, b);
  t.transform_associate_to_map();
  for (int k = 0; k < 6; k++) mapped6[k] = t.transformMapped[k];
}
void orc_wire_pose_to_quat(const float* rot3, double* q4) { wire_pose_to_quat(rot3, q4); }
void orc_wire_quat_to_pose(const double* q4, float* rot3) { wire_quat_to_pose(q4, rot3); }

// the kept points of a raw sweep in firing order, before any IMU projection: out5 = (x, y, z, intensity, relTime) per point,
// ring[] their scan ids; returns the count (tests feed these to the reference's projectPointToStartOfSweep)
int orc_multiscan_trace(const float* raw, int n, float lower_deg, float upper_deg, int n_rings, float scan_period, float* out5, int* ring, int cap) {
  MultiScanMapper m;
  m.set(lower_deg, upper_deg, (uint16_t)n_rings);
  BinTrace tr;
  bin_sweep(raw, (size_t)n, m, scan_period, nullptr, &tr);
  const int cnt = (int)tr.points.size();
  for (int i = 0; i < cnt && i < cap; i++) {
    out5[5 * i] = tr.points[i].x; out5[5 * i + 1] = tr.points[i].y; out5[5 * i + 2] = tr.points[i].z; out5[5 * i + 3] = tr.points[i].i;
    out5[5 * i + 4] = tr.relTime[i];
    ring[i] = tr.ring[i];
  }
  return cnt;
}

// ---- odometry ----
void* orc_odom_create() { return new LaserOdometry(); }
void orc_odom_destroy(void* h) { delete (LaserOdometry*)h; }
void orc_odom_config(void* h, float scanPeriod, int maxIterations, float deltaTAbort, float deltaRAbort) {
  auto* o = (LaserOdometry*)h;
  o->scanPeriod = scanPeriod; o->maxIterations = maxIterations; o->deltaTAbort = deltaTAbort; o->deltaRAbort = deltaRAbort;
}
// which: 0 full, 1 sharp, 2 lessSharp, 3 flat, 4 lessFlat
void orc_odom_set_cloud(void* h, int which, const float* pts, int n) {
  auto* o = (LaserOdometry*)h;
  Cloud* c[5] = {&o->laserCloud, &o->cornerSharp, &o->cornerLessSharp, &o->surfFlat, &o->surfLessFlat};
  *c[which] = to_cloud(pts, n);
}
void orc_odom_update_imu(void* h, const float* t12) { ((LaserOdometry*)h)->update_imu(t12); }
void orc_odom_set_transform(void* h, const float* t6) { twist_from(((LaserOdometry*)h)->transform, t6); }
void orc_odom_set_transform_sum(void* h, const float* t6) { twist_from(((LaserOdometry*)h)->transformSum, t6); }
void orc_odom_process(void* h) { ((LaserOdometry*)h)->process(); }
void orc_odom_get_transform(void* h, float* t6) { twist_to(((LaserOdometry*)h)->transform, t6); }
void orc_odom_get_transform_sum(void* h, float* t6) { twist_to(((LaserOdometry*)h)->transformSum, t6); }
// which: 0 lastCorner, 1 lastSurf, 2 full cloud
int orc_odom_get_cloud(void* h, int which, float* out, int cap) {
  auto* o = (LaserOdometry*)h;
  const Cloud* c[3] = {&o->lastCorner, &o->lastSurf, &o->laserCloud};
  return from_cloud(*c[which], out, cap);
}
void orc_odom_transform_full_to_end(void* h) { auto* o = (LaserOdometry*)h; o->transform_to_end(o->laserCloud); }
void orc_odom_stats(void* h, int* s3) {
  auto* o = (LaserOdometry*)h;
  s3[0] = o->lastIterCount; s3[1] = o->lastSelNum; s3[2] = (int)o->frameCount;
}

// ---- mapping ----
void* orc_map_create() { return new LaserMapping(); }
void orc_map_destroy(void* h) { delete (LaserMapping*)h; }
void orc_map_config(void* h, float scanPeriod, int maxIterations, float deltaTAbort, float deltaRAbort, float cornerLeaf,
                    float surfLeaf) {
  auto* m = (LaserMapping*)h;
  m->scanPeriod = scanPeriod; m->maxIterations = maxIterations; m->deltaTAbort = deltaTAbort; m->deltaRAbort = deltaRAbort;
  m->cornerLeaf = cornerLeaf; m->surfLeaf = surfLeaf;
}
// which: 0 cornerLast, 1 surfLast, 2 fullRes
void orc_map_set_cloud(void* h, int which, const float* pts, int n) {
  auto* m = (LaserMapping*)h;
  Cloud* c[3] = {&m->cornerLast, &m->surfLast, &m->fullRes};
  *c[which] = to_cloud(pts, n);
}
void orc_map_update_odometry(void* h, const float* t6) { ((LaserMapping*)h)->update_odometry(t6); }
int orc_map_process(void* h) { return ((LaserMapping*)h)->process() ? 1 : 0; }
void orc_map_update_imu(void* h, double stamp, float roll, float pitch) { ((LaserMapping*)h)->update_imu(stamp, roll, pitch); }
void orc_map_set_time(void* h, double t) { ((LaserMapping*)h)->laserOdometryTime = t; }
// which: 0 aft, 1 bef, 2 tobe, 3 sum
void orc_map_get_transform(void* h, int which, float* t6) {
  auto* m = (LaserMapping*)h;
  const Twist* t[4] = {&m->transformAftMapped, &m->transformBefMapped, &m->transformTobeMapped, &m->transformSum};
  twist_to(*t[which], t6);
}
void orc_map_set_transform(void* h, int which, const float* t6) {
  auto* m = (LaserMapping*)h;
  Twist* t[4] = {&m->transformAftMapped, &m->transformBefMapped, &m->transformTobeMapped, &m->transformSum};
  twist_from(*t[which], t6);
}
// which: 0 fullRes(registered), 1 surroundDS, 2 cornerFromMap, 3 surfFromMap, 4 cornerStackDS, 5 surfStackDS,
//        6 all corner cubes, 7 all surf cubes
int orc_map_get_cloud(void* h, int which, float* out, int cap) {
  auto* m = (LaserMapping*)h;
  if (which >= 6) {
    Cloud all;
    const auto& arr = which == 6 ? m->cornerArray : m->surfArray;
    for (const Cloud& c : arr) all.insert(all.end(), c.begin(), c.end());
    return from_cloud(all, out, cap);
  }
  const Cloud* c[6] = {&m->fullRes, &m->surroundDS, &m->cornerFromMap, &m->surfFromMap, &m->cornerStackDS, &m->surfStackDS};
  return from_cloud(*c[which], out, cap);
}
int orc_map_has_fresh_map(void* h) { return ((LaserMapping*)h)->downsizedMapCreated ? 1 : 0; }
// test hook: drop map-frame points straight into the cube arrays (by coordinate, current window centre)
void orc_map_load_cubes(void* h, const float* corner, int nc, const float* surf, int ns) {
  auto* m = (LaserMapping*)h;
  for (int t = 0; t < 2; t++) {
    const float* p = t == 0 ? corner : surf;
    int n = t == 0 ? nc : ns;
    auto& arr = t == 0 ? m->cornerArray : m->surfArray;
    for (int i = 0; i < n; i++) {
      Pt q{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]};
      int I = LaserMapping::cube_of(q.x, m->cenW), J = LaserMapping::cube_of(q.y, m->cenH), K = LaserMapping::cube_of(q.z, m->cenD);
      if (I >= 0 && I < LaserMapping::W && J >= 0 && J < LaserMapping::H && K >= 0 && K < LaserMapping::D)
        arr[LaserMapping::to_index(I, J, K)].push_back(q);
    }
  }
}
void orc_map_set_frozen(void* h, const float* corner, int nc, const float* surf, int ns) {
  ((LaserMapping*)h)->set_frozen_submap(to_cloud(corner, nc), to_cloud(surf, ns));
}
// transformAssociateToMap with the current sum/bef/aft; returns the predicted transformTobeMapped
void orc_map_associate(void* h, float* tobe6) {
  auto* m = (LaserMapping*)h;
  m->transform_associate_to_map();
  twist_to(m->transformTobeMapped, tobe6);
}
void orc_map_register_frozen(void* h, const float* guess6, float* pose6) {
  auto* m = (LaserMapping*)h;
  m->register_frozen(guess6);
  twist_to(m->transformTobeMapped, pose6);
}
// one residual pass at the given pose against the frozen sub-map, after register_frozen()/set clouds:
// returns rows; ori/coeff sized cap x 4
int orc_map_residual_pass(void* h, const float* pose6, float* ori, float* coeff, int cap) {
  auto* m = (LaserMapping*)h;
  twist_from(m->transformTobeMapped, pose6);
  std::vector<Pt> o, c;
  m->residual_pass(o, c);
  from_cloud(o, ori, cap);
  from_cloud(c, coeff, cap);
  return (int)o.size();
}
// s: iterations, lastSelNum, cornerDS, surfDS, cornerFromMap, surfFromMap, degenerate, optimized
void orc_map_stats(void* h, int* s8) {
  const MappingStats& s = ((LaserMapping*)h)->stats;
  s8[0] = s.iterations; s8[1] = s.lastSelNum; s8[2] = s.cornerDS; s8[3] = s.surfDS; s8[4] = s.cornerFromMap;
  s8[5] = s.surfFromMap; s8[6] = s.degenerate; s8[7] = s.optimized;
}
void orc_map_grid_center(void* h, int* c3) {
  auto* m = (LaserMapping*)h;
  c3[0] = m->cenW; c3[1] = m->cenH; c3[2] = m->cenD;
}

}  // extern "C"
