// C-ABI: the step after the path — loamx_tm_* mirrors loam::BasicTransformMaintenance
// (include/loam_velodyne/BasicTransformMaintenance.h:44-66, src/lib/BasicTransformMaintenance.cpp:46-178) and
// loamx_wire_* the orientation convention the four nodes use on nav_msgs/Odometry (src/lib/LaserOdometry.cpp:300-308,
// src/lib/LaserMapping.cpp:205-213, src/lib/TransformMaintenance.cpp:66-115).  Pure host arithmetic at odometry rate
// (a few hundred flops per message): there is nothing to put on the GPU, so these entry points need no device.
#include "common.h"
#include <cmath>

using namespace loamx;

struct loamx_tm {
  float sum[6] = {0}, incre[6] = {0}, mapped[6] = {0}, bef[6] = {0}, aft[6] = {0};   // float state as in the reference (.h:55-60)
};

namespace {

// Rotation helpers on plain floats; every sin / cos is the float overload, as in the reference file (`using std::sin` ...).
struct SC { float s, c; };
inline SC sc(float a) { return {std::sin(a), std::cos(a)}; }

void associate_to_map(loamx_tm& t) {
  const SC cx = sc(t.sum[0]), cy = sc(t.sum[1]), cz = sc(t.sum[2]);   // "bc": transformSum
  const SC lx = sc(t.bef[0]), ly = sc(t.bef[1]), lz = sc(t.bef[2]);   // "bl": transformBefMapped
  const SC ax = sc(t.aft[0]), ay = sc(t.aft[1]), az = sc(t.aft[2]);   // "al": transformAftMapped
  // increment since the last mapping result, expressed in the odometry frame at transformSum (:86-98)
  const float dx = t.bef[3] - t.sum[3], dy = t.bef[4] - t.sum[4], dz = t.bef[5] - t.sum[5];
  const float x1 = cy.c * dx - cy.s * dz, y1 = dy, z1 = cy.s * dx + cy.c * dz;
  const float x2 = x1, y2 = cx.c * y1 + cx.s * z1, z2 = -cx.s * y1 + cx.c * z1;
  t.incre[3] = cz.c * x2 + cz.s * y2;
  t.incre[4] = -cz.s * x2 + cz.c * y2;
  t.incre[5] = z2;
  // rotation: R_mapped = R_aft * R_bef^-1 * R_sum written out in sines and cosines (:100-160); the grouping of the
  // products follows the reference expression term by term
  const float sbcx = cx.s, cbcx = cx.c, sbcy = cy.s, cbcy = cy.c, sbcz = cz.s, cbcz = cz.c;
  const float sblx = lx.s, cblx = lx.c, sbly = ly.s, cbly = ly.c, sblz = lz.s, cblz = lz.c;
  const float salx = ax.s, calx = ax.c, saly = ay.s, caly = ay.c, salz = az.s, calz = az.c;
  const float srx = -sbcx * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz) -
                    cbcx * sbcy * (calx * calz * (cbly * sblz - cblz * sblx * sbly) - calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) -
                    cbcx * cbcy * (calx * salz * (cblz * sbly - cbly * sblx * sblz) - calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx);
  t.mapped[0] = -std::asin(srx);
  const float srycrx =
      sbcx * (cblx * cblz * (caly * salz - calz * salx * saly) - cblx * sblz * (caly * calz + salx * saly * salz) + calx * saly * sblx) -
      cbcx * cbcy * ((caly * calz + salx * saly * salz) * (cblz * sbly - cbly * sblx * sblz) + (caly * salz - calz * salx * saly) * (sbly * sblz + cbly * cblz * sblx) -
                     calx * cblx * cbly * saly) +
      cbcx * sbcy * ((caly * calz + salx * saly * salz) * (cbly * cblz + sblx * sbly * sblz) + (caly * salz - calz * salx * saly) * (cbly * sblz - cblz * sblx * sbly) +
                     calx * cblx * saly * sbly);
  const float crycrx =
      sbcx * (cblx * sblz * (calz * saly - caly * salx * salz) - cblx * cblz * (saly * salz + caly * calz * salx) + calx * caly * sblx) +
      cbcx * cbcy * ((saly * salz + caly * calz * salx) * (sbly * sblz + cbly * cblz * sblx) + (calz * saly - caly * salx * salz) * (cblz * sbly - cbly * sblx * sblz) +
                     calx * caly * cblx * cbly) -
      cbcx * sbcy * ((saly * salz + caly * calz * salx) * (cbly * sblz - cblz * sblx * sbly) + (calz * saly - caly * salx * salz) * (cbly * cblz + sblx * sbly * sblz) -
                     calx * caly * cblx * sbly);
  const float cm0 = std::cos(t.mapped[0]);
  t.mapped[1] = std::atan2(srycrx / cm0, crycrx / cm0);
  const float srzcrx =
      (cbcz * sbcy - cbcy * sbcx * sbcz) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) - calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx) -
      (cbcy * cbcz + sbcx * sbcy * sbcz) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) - calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) +
      cbcx * sbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
  const float crzcrx =
      (cbcy * sbcz - cbcz * sbcx * sbcy) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) - calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) -
      (sbcy * sbcz + cbcy * cbcz * sbcx) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) - calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx) +
      cbcx * cbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
  t.mapped[2] = std::atan2(srzcrx / cm0, crzcrx / cm0);
  // translation (:162-176)
  const SC m0 = sc(t.mapped[0]), m1 = sc(t.mapped[1]), m2 = sc(t.mapped[2]);
  const float u1 = m2.c * t.incre[3] - m2.s * t.incre[4], v1 = m2.s * t.incre[3] + m2.c * t.incre[4], w1 = t.incre[5];
  const float u2 = u1, v2 = m0.c * v1 - m0.s * w1, w2 = m0.s * v1 + m0.c * w1;
  t.mapped[3] = t.aft[3] - (m1.c * u2 + m1.s * w2);
  t.mapped[4] = t.aft[4] - v2;
  t.mapped[5] = t.aft[5] - (-m1.s * u2 + m1.c * w2);
}

// tf::createQuaternionMsgFromRollPitchYaw (tf/LinearMath/Quaternion.h, setRPY)
void quat_from_rpy(double roll, double pitch, double yaw, double q[4]) {
  const double cy = std::cos(yaw * 0.5), sy = std::sin(yaw * 0.5), cp = std::cos(pitch * 0.5), sp = std::sin(pitch * 0.5),
               cr = std::cos(roll * 0.5), sr = std::sin(roll * 0.5);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}
// tf::Matrix3x3(q).getRPY (setRotation + getEulerYPR, first solution)
void rpy_from_quat(const double q[4], double& roll, double& pitch, double& yaw) {
  const double s = 2.0 / (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs, xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs, yy = q[1] * ys, yz = q[1] * zs,
               zz = q[2] * zs;
  const double r00 = 1.0 - (yy + zz), r10 = xy + wz, r20 = xz - wy, r21 = yz + wx, r22 = 1.0 - (xx + yy);
  if (std::fabs(r20) >= 1) {
    yaw = 0;
    roll = std::atan2(r21, r22);
    pitch = r20 < 0 ? M_PI / 2.0 : -M_PI / 2.0;
  } else {
    pitch = -std::asin(r20);
    const double c = std::cos(pitch);
    roll = std::atan2(r21 / c, r22 / c);
    yaw = std::atan2(r10 / c, r00 / c);
  }
}

}  // namespace

extern "C" {

loamx_tm* loamx_tm_create(void) {
  loamx_tm* h = nullptr;
  guard([&]() { h = new loamx_tm(); return LOAMX_OK; });
  return h;
}
void loamx_tm_destroy(loamx_tm* h) { delete h; }

int loamx_tm_update_odometry(loamx_tm* h, const float transform_sum[6]) {
  return guard([&]() {
    LX_REQUIRE(h && transform_sum, "NULL argument");
    for (int k = 0; k < 6; k++) h->sum[k] = transform_sum[k];
    return LOAMX_OK;
  });
}
int loamx_tm_update_mapping_transform(loamx_tm* h, const float aft_mapped[6], const float bef_mapped[6]) {
  return guard([&]() {
    LX_REQUIRE(h && aft_mapped && bef_mapped, "NULL argument");
    for (int k = 0; k < 6; k++) { h->aft[k] = aft_mapped[k]; h->bef[k] = bef_mapped[k]; }
    return LOAMX_OK;
  });
}
int loamx_tm_associate_to_map(loamx_tm* h) {
  return guard([&]() {
    LX_REQUIRE(h, "NULL handle");
    associate_to_map(*h);
    return LOAMX_OK;
  });
}
int loamx_tm_get_mapped(loamx_tm* h, float transform_mapped[6]) {
  return guard([&]() {
    LX_REQUIRE(h && transform_mapped, "NULL argument");
    for (int k = 0; k < 6; k++) transform_mapped[k] = h->mapped[k];
    return LOAMX_OK;
  });
}

int loamx_wire_pose_to_quat(const float rot_xyz[3], double quat_xyzw[4]) {
  return guard([&]() {
    LX_REQUIRE(rot_xyz && quat_xyzw, "NULL argument");
    double g[4];
    quat_from_rpy(rot_xyz[2], -rot_xyz[0], -rot_xyz[1], g);
    quat_xyzw[0] = -g[1]; quat_xyzw[1] = -g[2]; quat_xyzw[2] = g[0]; quat_xyzw[3] = g[3];
    return LOAMX_OK;
  });
}
int loamx_wire_quat_to_pose(const double quat_xyzw[4], float rot_xyz[3]) {
  return guard([&]() {
    LX_REQUIRE(rot_xyz && quat_xyzw, "NULL argument");
    const double q[4] = {quat_xyzw[2], -quat_xyzw[0], -quat_xyzw[1], quat_xyzw[3]};
    LX_REQUIRE(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] > 0.0, "zero quaternion");
    double roll, pitch, yaw;
    rpy_from_quat(q, roll, pitch, yaw);
    rot_xyz[0] = (float)-pitch; rot_xyz[1] = (float)-yaw; rot_xyz[2] = (float)roll;
    return LOAMX_OK;
  });
}

}  // extern "C"
