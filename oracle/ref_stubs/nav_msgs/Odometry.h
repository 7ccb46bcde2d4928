// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: the fields of an odometry message.
#pragma once
#include <string>
#include <geometry_msgs/Quaternion.h>
#include <sensor_msgs/PointCloud2.h>
namespace nav_msgs {
struct Odometry {
  std_msgs::Header header;
  std::string child_frame_id;
  geometry_msgs::PoseWithCovariance pose;
  geometry_msgs::TwistWithCovariance twist;
  typedef boost::shared_ptr<Odometry> Ptr;
  typedef boost::shared_ptr<Odometry const> ConstPtr;
};
}  // namespace nav_msgs
