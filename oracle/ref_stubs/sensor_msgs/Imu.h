// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: the fields of an IMU message.
#pragma once
#include <geometry_msgs/Quaternion.h>
#include <sensor_msgs/PointCloud2.h>
namespace sensor_msgs {
struct Imu {
  std_msgs::Header header;
  geometry_msgs::Quaternion orientation;
  geometry_msgs::Vector3 linear_acceleration;
  typedef boost::shared_ptr<Imu> Ptr;
  typedef boost::shared_ptr<Imu const> ConstPtr;
};
}  // namespace sensor_msgs
