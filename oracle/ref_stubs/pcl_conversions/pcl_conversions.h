// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT pcl_conversions: copies the stand-in message's (x, y, z) payload into a cloud.
#pragma once
#include <pcl/point_cloud.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl {
template <class PointT> inline void toROSMsg(const PointCloud<PointT>&, sensor_msgs::PointCloud2&) {}
template <class PointT> inline void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<PointT>& cloud) {
  cloud.clear();
  for (size_t i = 0; i + 2 < msg.xyz.size(); i += 3) {
    PointT p;
    p.x = msg.xyz[i]; p.y = msg.xyz[i + 1]; p.z = msg.xyz[i + 2];
    cloud.push_back(p);
  }
}
}  // namespace pcl
