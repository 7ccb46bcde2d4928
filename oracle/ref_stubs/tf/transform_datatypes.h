// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT tf: a quaternion, its roll / pitch / yaw conversions (ZYX fixed-axis convention of
// tf::Matrix3x3::getRPY / createQuaternionMsgFromRollPitchYaw) and a stamped transform record.
#pragma once
#include <cmath>
#include <string>
#include <ros/ros.h>
#include <geometry_msgs/Quaternion.h>
namespace tf {
struct Quaternion {
  double x_ = 0, y_ = 0, z_ = 0, w_ = 1;
  Quaternion() = default;
  Quaternion(double x, double y, double z, double w) : x_(x), y_(y), z_(z), w_(w) {}
};
struct Vector3 {
  double v[3] = {0, 0, 0};
  Vector3() = default;
  Vector3(double x, double y, double z) : v{x, y, z} {}
};
struct Matrix3x3 {
  Quaternion q;
  explicit Matrix3x3(const Quaternion& q_) : q(q_) {}
  void getRPY(double& roll, double& pitch, double& yaw) const {
    const double x = q.x_, y = q.y_, z = q.z_, w = q.w_;
    roll = std::atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
    const double s = 2 * (w * y - z * x);
    pitch = std::fabs(s) >= 1 ? std::copysign(M_PI / 2, s) : std::asin(s);
    yaw = std::atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
  }
};
inline void quaternionMsgToTF(const geometry_msgs::Quaternion& m, Quaternion& q) { q = Quaternion(m.x, m.y, m.z, m.w); }
inline geometry_msgs::Quaternion createQuaternionMsgFromRollPitchYaw(double roll, double pitch, double yaw) {
  const double cr = std::cos(roll / 2), sr = std::sin(roll / 2), cp = std::cos(pitch / 2), sp = std::sin(pitch / 2), cy = std::cos(yaw / 2),
               sy = std::sin(yaw / 2);
  geometry_msgs::Quaternion q;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  q.w = cr * cp * cy + sr * sp * sy;
  return q;
}
struct StampedTransform {
  ros::Time stamp_;
  std::string frame_id_, child_frame_id_;
  Quaternion rotation;
  Vector3 origin;
  void setRotation(const Quaternion& q) { rotation = q; }
  void setOrigin(const Vector3& o) { origin = o; }
};
}  // namespace tf
