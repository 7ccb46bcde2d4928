// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: the fields of an IMU message.
#pragma once
#include <sensor_msgs/PointCloud2.h>
namespace sensor_msgs {
struct Imu {
  std_msgs::Header header;
  struct { double x = 0, y = 0, z = 0, w = 1; } orientation;
  struct { double x = 0, y = 0, z = 0; } linear_acceleration;
  typedef boost::shared_ptr<Imu> Ptr;
  typedef boost::shared_ptr<Imu const> ConstPtr;
};
}  // namespace sensor_msgs
