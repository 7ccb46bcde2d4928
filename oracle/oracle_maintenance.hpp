// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp for the rules and for what is / is not pinned by the reference's own code).
//
// Pose fusion after the path, restating
//   BasicTransformMaintenance::updateOdometry / updateMappingTransform / transformAssociateToMap
//                                    -> src/lib/BasicTransformMaintenance.cpp:46-178 (state: float[6] arrays, .h:55-60)
//   pose <-> nav_msgs/Odometry orientation as the four nodes exchange it
//                                    -> src/lib/LaserOdometry.cpp:300-308 (publish), src/lib/LaserMapping.cpp:205-213 and
//                                       src/lib/TransformMaintenance.cpp:66-115 (receive)
// tf (ROS geometry, not under /root/reference) supplies createQuaternionMsgFromRollPitchYaw and Matrix3x3::getRPY; their
// published algorithm (tf/LinearMath/Quaternion.h setRPY, Matrix3x3.h setRotation / getEulerYPR solution 1) is restated.
// The arrays are float and the file does `using std::sin` etc., so every sin / cos / asin / atan2 below is the float overload.
#pragma once
#include <cmath>

namespace loam_oracle {

struct TransformMaintenance {
  float transformSum[6] = {0}, transformIncre[6] = {0}, transformMapped[6] = {0}, transformBefMapped[6] = {0}, transformAftMapped[6] = {0};

  void update_odometry(double pitch, double yaw, double roll, double x, double y, double z) {   // :46-54
    transformSum[0] = pitch; transformSum[1] = yaw; transformSum[2] = roll;
    transformSum[3] = x; transformSum[4] = y; transformSum[5] = z;
  }
  void update_mapping_transform(const double aft[6], const double bef[6]) {   // :56-73
    for (int k = 0; k < 6; k++) { transformAftMapped[k] = aft[k]; transformBefMapped[k] = bef[k]; }
  }
  void transform_associate_to_map() {   // :84-178
    using std::sin; using std::cos; using std::asin; using std::atan2;
    const float* S = transformSum; const float* B = transformBefMapped; const float* A = transformAftMapped;
    float x1 = cos(S[1]) * (B[3] - S[3]) - sin(S[1]) * (B[5] - S[5]);
    float y1 = B[4] - S[4];
    float z1 = sin(S[1]) * (B[3] - S[3]) + cos(S[1]) * (B[5] - S[5]);
    float x2 = x1;
    float y2 = cos(S[0]) * y1 + sin(S[0]) * z1;
    float z2 = -sin(S[0]) * y1 + cos(S[0]) * z1;
    transformIncre[3] = cos(S[2]) * x2 + sin(S[2]) * y2;
    transformIncre[4] = -sin(S[2]) * x2 + cos(S[2]) * y2;
    transformIncre[5] = z2;
    float sbcx = sin(S[0]), cbcx = cos(S[0]), sbcy = sin(S[1]), cbcy = cos(S[1]), sbcz = sin(S[2]), cbcz = cos(S[2]);
    float sblx = sin(B[0]), cblx = cos(B[0]), sbly = sin(B[1]), cbly = cos(B[1]), sblz = sin(B[2]), cblz = cos(B[2]);
    float salx = sin(A[0]), calx = cos(A[0]), saly = sin(A[1]), caly = cos(A[1]), salz = sin(A[2]), calz = cos(A[2]);
    float srx = -sbcx * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz)
                - cbcx * sbcy * (calx * calz * (cbly * sblz - cblz * sblx * sbly) - calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly)
                - cbcx * cbcy * (calx * salz * (cblz * sbly - cbly * sblx * sblz) - calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx);
    transformMapped[0] = -asin(srx);
    float srycrx = sbcx * (cblx * cblz * (caly * salz - calz * salx * saly) - cblx * sblz * (caly * calz + salx * saly * salz) + calx * saly * sblx)
                   - cbcx * cbcy * ((caly * calz + salx * saly * salz) * (cblz * sbly - cbly * sblx * sblz) +
                                    (caly * salz - calz * salx * saly) * (sbly * sblz + cbly * cblz * sblx) - calx * cblx * cbly * saly)
                   + cbcx * sbcy * ((caly * calz + salx * saly * salz) * (cbly * cblz + sblx * sbly * sblz) +
                                    (caly * salz - calz * salx * saly) * (cbly * sblz - cblz * sblx * sbly) + calx * cblx * saly * sbly);
    float crycrx = sbcx * (cblx * sblz * (calz * saly - caly * salx * salz) - cblx * cblz * (saly * salz + caly * calz * salx) + calx * caly * sblx)
                   + cbcx * cbcy * ((saly * salz + caly * calz * salx) * (sbly * sblz + cbly * cblz * sblx) +
                                    (calz * saly - caly * salx * salz) * (cblz * sbly - cbly * sblx * sblz) + calx * caly * cblx * cbly)
                   - cbcx * sbcy * ((saly * salz + caly * calz * salx) * (cbly * sblz - cblz * sblx * sbly) +
                                    (calz * saly - caly * salx * salz) * (cbly * cblz + sblx * sbly * sblz) - calx * caly * cblx * sbly);
    transformMapped[1] = atan2(srycrx / cos(transformMapped[0]), crycrx / cos(transformMapped[0]));
    float srzcrx = (cbcz * sbcy - cbcy * sbcx * sbcz) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) - calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx)
                   - (cbcy * cbcz + sbcx * sbcy * sbcz) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) - calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly)
                   + cbcx * sbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    float crzcrx = (cbcy * sbcz - cbcz * sbcx * sbcy) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) - calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly)
                   - (sbcy * sbcz + cbcy * cbcz * sbcx) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) - calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx)
                   + cbcx * cbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    transformMapped[2] = atan2(srzcrx / cos(transformMapped[0]), crzcrx / cos(transformMapped[0]));
    x1 = cos(transformMapped[2]) * transformIncre[3] - sin(transformMapped[2]) * transformIncre[4];
    y1 = sin(transformMapped[2]) * transformIncre[3] + cos(transformMapped[2]) * transformIncre[4];
    z1 = transformIncre[5];
    x2 = x1;
    y2 = cos(transformMapped[0]) * y1 - sin(transformMapped[0]) * z1;
    z2 = sin(transformMapped[0]) * y1 + cos(transformMapped[0]) * z1;
    transformMapped[3] = A[3] - (cos(transformMapped[1]) * x2 + sin(transformMapped[1]) * z2);
    transformMapped[4] = A[4] - y2;
    transformMapped[5] = A[5] - (-sin(transformMapped[1]) * x2 + cos(transformMapped[1]) * z2);
  }
};

// tf::createQuaternionMsgFromRollPitchYaw (Quaternion::setRPY): q = (x, y, z, w)
inline void tf_quat_from_rpy(double roll, double pitch, double yaw, double q[4]) {
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}
// tf::Matrix3x3(q).getRPY (setRotation + getEulerYPR, solution 1)
inline void tf_rpy_from_quat(const double q[4], double& roll, double& pitch, double& yaw) {
  const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double s = 2.0 / d;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs, xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs, yy = q[1] * ys, yz = q[1] * zs,
               zz = q[2] * zs;
  const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  if (std::fabs(m20) >= 1) {   // gimbal lock: yaw = 0, roll from the difference-of-angles formula
    yaw = 0;
    roll = std::atan2(m21, m22);
    pitch = m20 < 0 ? M_PI / 2.0 : -M_PI / 2.0;
  } else {
    pitch = -std::asin(m20);
    roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
    yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
  }
}
// LaserOdometry.cpp:300-308: (rot_x, rot_y, rot_z) -> message orientation (x, y, z, w)
inline void wire_pose_to_quat(const float rot[3], double out[4]) {
  double g[4];
  tf_quat_from_rpy(rot[2], -rot[0], -rot[1], g);
  out[0] = -g[1]; out[1] = -g[2]; out[2] = g[0]; out[3] = g[3];
}
// LaserMapping.cpp:205-213: message orientation -> (rot_x, rot_y, rot_z) = (-pitch, -yaw, roll)
inline void wire_quat_to_pose(const double msg[4], float rot[3]) {
  const double q[4] = {msg[2], -msg[0], -msg[1], msg[3]};
  double roll, pitch, yaw;
  tf_rpy_from_quat(q, roll, pitch, yaw);
  rot[0] = (float)-pitch; rot[1] = (float)-yaw; rot[2] = (float)roll;
}

}  // namespace loam_oracle
