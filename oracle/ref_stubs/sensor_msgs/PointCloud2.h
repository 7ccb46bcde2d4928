// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT ROS: a message that carries a time stamp and a flat float payload of `floats_per_point`
// values per point (3: x y z; 4: x y z intensity) instead of the byte-serialised wire format.
#pragma once
#include <string>
#include <vector>
#include <ros/ros.h>
namespace std_msgs { struct Header { ros::Time stamp; std::string frame_id; }; }
namespace sensor_msgs {
struct PointCloud2 {
  std_msgs::Header header;
  std::vector<float> data;
  int floats_per_point = 3;
  typedef boost::shared_ptr<PointCloud2> Ptr;
  typedef boost::shared_ptr<PointCloud2 const> ConstPtr;
};
typedef PointCloud2::ConstPtr PointCloud2ConstPtr;
}  // namespace sensor_msgs
