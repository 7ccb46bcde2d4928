// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: a message that carries a time stamp and a flat (x, y, z) payload.
#pragma once
#include <string>
#include <vector>
#include <ros/ros.h>
namespace std_msgs { struct Header { ros::Time stamp; std::string frame_id; }; }
namespace sensor_msgs {
struct PointCloud2 {
  std_msgs::Header header;
  std::vector<float> xyz;
  typedef boost::shared_ptr<PointCloud2> Ptr;
  typedef boost::shared_ptr<PointCloud2 const> ConstPtr;
};
typedef PointCloud2::ConstPtr PointCloud2ConstPtr;
}  // namespace sensor_msgs
