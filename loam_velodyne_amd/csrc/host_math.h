// Host-side scalar pose algebra of the registration path (product code; the oracle has its own restatement).
// These are the once-per-sweep closed-form Euler compositions, tens of flops each, so they stay on the host:
//   HAngle / HTwist            <-> reference include/loam_velodyne/Angle.h:16-67, Twist.h:15-27
//   accumulate_rotation        <-> src/lib/BasicLaserOdometry.cpp:155-179
//   plugin_imu_rotation        <-> src/lib/BasicLaserOdometry.cpp:91-151
//   transform_associate_to_map <-> src/lib/BasicLaserMapping.cpp:103-167
#pragma once
#include <cmath>
#include "dev_math.hpp"

namespace loamx {

struct HAngle {
  float r = 0.f, c = 1.f, s = 0.f;
  HAngle() = default;
  HAngle(float rad) : r(rad), c(std::cos(rad)), s(std::sin(rad)) {}
  HAngle operator-() const {   // Angle.h:47-53: only the sine flips
    HAngle o;
    o.r = -r; o.c = c; o.s = -s;
    return o;
  }
  float rad() const { return r; }
  float cos() const { return c; }
  float sin() const { return s; }
};

struct HVec3 {
  float x = 0.f, y = 0.f, z = 0.f;
};

struct HTwist {
  HAngle rot_x, rot_y, rot_z;
  HVec3 pos;
  void get(float* t6) const {
    t6[0] = rot_x.r; t6[1] = rot_y.r; t6[2] = rot_z.r; t6[3] = pos.x; t6[4] = pos.y; t6[5] = pos.z;
  }
  void set(const float* t6) {
    rot_x = HAngle(t6[0]); rot_y = HAngle(t6[1]); rot_z = HAngle(t6[2]);
    pos = {t6[3], t6[4], t6[5]};
  }
  Pose pose() const {
    Pose p;
    p.rx = rot_x.r; p.ry = rot_y.r; p.rz = rot_z.r; p.tx = pos.x; p.ty = pos.y; p.tz = pos.z;
    p.srx = rot_x.s; p.crx = rot_x.c; p.sry = rot_y.s; p.cry = rot_y.c; p.srz = rot_z.s; p.crz = rot_z.c;
    return p;
  }
};

inline void h_rot_zxy(HVec3& v, const HAngle& az, const HAngle& ax, const HAngle& ay) {
  rot_z(v.x, v.y, az.c, az.s);
  rot_x(v.y, v.z, ax.c, ax.s);
  rot_y(v.x, v.z, ay.c, ay.s);
}
inline void h_rot_yxz(HVec3& v, const HAngle& ay, const HAngle& ax, const HAngle& az) {
  rot_y(v.x, v.z, ay.c, ay.s);
  rot_x(v.y, v.z, ax.c, ax.s);
  rot_z(v.x, v.y, az.c, az.s);
}

// BasicLaserOdometry.cpp:155-179
inline void accumulate_rotation(HAngle cx, HAngle cy, HAngle cz, HAngle lx, HAngle ly, HAngle lz, HAngle& ox, HAngle& oy,
                                HAngle& oz) {
  const float srx = lx.c * cx.c * ly.s * cz.s - cx.c * cz.c * lx.s - lx.c * ly.c * cx.s;
  ox = HAngle(-std::asin(srx));
  const float srycrx = lx.s * (cy.c * cz.s - cz.c * cx.s * cy.s) + lx.c * ly.s * (cy.c * cz.c + cx.s * cy.s * cz.s) +
                       lx.c * ly.c * cx.c * cy.s;
  const float crycrx = lx.c * ly.c * cx.c * cy.c - lx.c * ly.s * (cz.c * cy.s - cy.c * cx.s * cz.s) -
                       lx.s * (cy.s * cz.s + cy.c * cz.c * cx.s);
  oy = HAngle(std::atan2(srycrx / ox.c, crycrx / ox.c));
  const float srzcrx = cx.s * (lz.c * ly.s - ly.c * lx.s * lz.s) + cx.c * cz.s * (ly.c * lz.c + lx.s * ly.s * lz.s) +
                       lx.c * cx.c * cz.c * lz.s;
  const float crzcrx = lx.c * lz.c * cx.c * cz.c - cx.c * cz.s * (ly.c * lz.s - lz.c * lx.s * ly.s) -
                       cx.s * (ly.s * lz.s + ly.c * lz.c * lx.s);
  oz = HAngle(std::atan2(srzcrx / ox.c, crzcrx / ox.c));
}

// BasicLaserOdometry.cpp:91-151.  Inputs are read into locals before any output is written (outputs may alias inputs).
inline void plugin_imu_rotation(const HAngle& bcx, const HAngle& bcy, const HAngle& bcz, const HAngle& blx, const HAngle& bly,
                                const HAngle& blz, const HAngle& alx, const HAngle& aly, const HAngle& alz, HAngle& acx,
                                HAngle& acy, HAngle& acz) {
  const float sbcx = bcx.s, cbcx = bcx.c, sbcy = bcy.s, cbcy = bcy.c, sbcz = bcz.s, cbcz = bcz.c;
  const float sblx = blx.s, cblx = blx.c, sbly = bly.s, cbly = bly.c, sblz = blz.s, cblz = blz.c;
  const float salx = alx.s, calx = alx.c, saly = aly.s, caly = aly.c, salz = alz.s, calz = alz.c;

  const float t1 = calx * saly * (cbly * sblz - cblz * sblx * sbly) - calx * caly * (sbly * sblz + cbly * cblz * sblx) + cblx * cblz * salx;
  const float t2 = calx * caly * (cblz * sbly - cbly * sblx * sblz) - calx * saly * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sblz;
  const float t3 = salx * sblx + calx * caly * cblx * cbly + calx * cblx * saly * sbly;
  const float srx = -sbcx * t3 - cbcx * cbcz * t1 - cbcx * sbcz * t2;
  acx = HAngle(-std::asin(srx));

  const float srycrx = (cbcy * sbcz - cbcz * sbcx * sbcy) * t1 - (cbcy * cbcz + sbcx * sbcy * sbcz) * t2 + cbcx * sbcy * t3;
  const float crycrx = (cbcz * sbcy - cbcy * sbcx * sbcz) * t2 - (sbcy * sbcz + cbcy * cbcz * sbcx) * t1 + cbcx * cbcy * t3;
  acy = HAngle(std::atan2(srycrx / acx.c, crycrx / acx.c));

  const float u1 = caly * calz + salx * saly * salz, u2 = calz * saly - caly * salx * salz;
  const float u3 = saly * salz + caly * calz * salx, u4 = caly * salz - calz * salx * saly;
  const float srzcrx = sbcx * (cblx * cbly * u2 - cblx * sbly * u1 + calx * salz * sblx) -
                       cbcx * cbcz * (u1 * (cbly * sblz - cblz * sblx * sbly) + u2 * (sbly * sblz + cbly * cblz * sblx) - calx * cblx * cblz * salz) +
                       cbcx * sbcz * (u1 * (cbly * cblz + sblx * sbly * sblz) + u2 * (cblz * sbly - cbly * sblx * sblz) + calx * cblx * salz * sblz);
  const float crzcrx = sbcx * (cblx * sbly * u4 - cblx * cbly * u3 + calx * calz * sblx) +
                       cbcx * cbcz * (u3 * (sbly * sblz + cbly * cblz * sblx) + u4 * (cbly * sblz - cblz * sblx * sbly) + calx * calz * cblx * cblz) -
                       cbcx * sbcz * (u3 * (cblz * sbly - cbly * sblx * sblz) + u4 * (cbly * cblz + sblx * sbly * sblz) - calx * calz * cblx * sblz);
  acz = HAngle(std::atan2(srzcrx / acx.c, crzcrx / acx.c));
}

// BasicLaserMapping.cpp:103-167: predicts transformTobeMapped from (transformSum, transformBefMapped, transformAftMapped)
inline void transform_associate_to_map(const HTwist& sum, const HTwist& bef, const HTwist& aft, HTwist& incre, HTwist& tobe) {
  incre.pos = {bef.pos.x - sum.pos.x, bef.pos.y - sum.pos.y, bef.pos.z - sum.pos.z};
  h_rot_yxz(incre.pos, -sum.rot_y, -sum.rot_x, -sum.rot_z);

  const float sbcx = sum.rot_x.s, cbcx = sum.rot_x.c, sbcy = sum.rot_y.s, cbcy = sum.rot_y.c, sbcz = sum.rot_z.s, cbcz = sum.rot_z.c;
  const float sblx = bef.rot_x.s, cblx = bef.rot_x.c, sbly = bef.rot_y.s, cbly = bef.rot_y.c, sblz = bef.rot_z.s, cblz = bef.rot_z.c;
  const float salx = aft.rot_x.s, calx = aft.rot_x.c, saly = aft.rot_y.s, caly = aft.rot_y.c, salz = aft.rot_z.s, calz = aft.rot_z.c;

  const float p1 = calx * calz * (cbly * sblz - cblz * sblx * sbly) - calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly;
  const float p2 = calx * salz * (cblz * sbly - cbly * sblx * sblz) - calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx;
  const float p3 = salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz;
  const float srx = -sbcx * p3 - cbcx * sbcy * p1 - cbcx * cbcy * p2;
  tobe.rot_x = HAngle(-std::asin(srx));

  const float q1 = caly * calz + salx * saly * salz, q2 = caly * salz - calz * salx * saly;
  const float q3 = saly * salz + caly * calz * salx, q4 = calz * saly - caly * salx * salz;
  const float srycrx = sbcx * (cblx * cblz * q2 - cblx * sblz * q1 + calx * saly * sblx) -
                       cbcx * cbcy * (q1 * (cblz * sbly - cbly * sblx * sblz) + q2 * (sbly * sblz + cbly * cblz * sblx) - calx * cblx * cbly * saly) +
                       cbcx * sbcy * (q1 * (cbly * cblz + sblx * sbly * sblz) + q2 * (cbly * sblz - cblz * sblx * sbly) + calx * cblx * saly * sbly);
  const float crycrx = sbcx * (cblx * sblz * q4 - cblx * cblz * q3 + calx * caly * sblx) +
                       cbcx * cbcy * (q3 * (sbly * sblz + cbly * cblz * sblx) + q4 * (cblz * sbly - cbly * sblx * sblz) + calx * caly * cblx * cbly) -
                       cbcx * sbcy * (q3 * (cbly * sblz - cblz * sblx * sbly) + q4 * (cbly * cblz + sblx * sbly * sblz) - calx * caly * cblx * sbly);
  tobe.rot_y = HAngle(std::atan2(srycrx / tobe.rot_x.c, crycrx / tobe.rot_x.c));

  const float srzcrx = (cbcz * sbcy - cbcy * sbcx * sbcz) * p2 - (cbcy * cbcz + sbcx * sbcy * sbcz) * p1 + cbcx * sbcz * p3;
  const float crzcrx = (cbcy * sbcz - cbcz * sbcx * sbcy) * p1 - (sbcy * sbcz + cbcy * cbcz * sbcx) * p2 + cbcx * cbcz * p3;
  tobe.rot_z = HAngle(std::atan2(srzcrx / tobe.rot_x.c, crzcrx / tobe.rot_x.c));

  HVec3 v = incre.pos;
  h_rot_zxy(v, tobe.rot_z, tobe.rot_x, tobe.rot_y);
  tobe.pos = {aft.pos.x - v.x, aft.pos.y - v.y, aft.pos.z - v.z};
}

}  // namespace loamx
