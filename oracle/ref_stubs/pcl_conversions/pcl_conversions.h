// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT pcl_conversions: copies points between a cloud and the stand-in message's float payload
// (x y z [intensity]); a field the message lacks keeps the point type's default, as pcl::fromROSMsg leaves it.
#pragma once
#include <pcl/point_cloud.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl {
namespace detail {
inline void put_intensity(const PointXYZI& p, std::vector<float>& d) { d.push_back(p.intensity); }
inline void put_intensity(const PointXYZ&, std::vector<float>&) {}
inline void get_intensity(PointXYZI& p, const float* f, int n) { if (n >= 4) p.intensity = f[3]; }
inline void get_intensity(PointXYZ&, const float*, int) {}
template <class P> struct Floats { static const int n = 3; };
template <> struct Floats<PointXYZI> { static const int n = 4; };
}  // namespace detail
template <class PointT> inline void toROSMsg(const PointCloud<PointT>& cloud, sensor_msgs::PointCloud2& msg) {
  msg.floats_per_point = detail::Floats<PointT>::n;
  msg.data.clear();
  for (const PointT& p : cloud.points) {
    msg.data.push_back(p.x); msg.data.push_back(p.y); msg.data.push_back(p.z);
    detail::put_intensity(p, msg.data);
  }
}
template <class PointT> inline void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<PointT>& cloud) {
  cloud.clear();
  const int n = msg.floats_per_point;
  for (size_t i = 0; i + n <= msg.data.size(); i += n) {
    PointT p;
    p.x = msg.data[i]; p.y = msg.data[i + 1]; p.z = msg.data[i + 2];
    detail::get_intensity(p, &msg.data[i], n);
    cloud.push_back(p);
  }
}
}  // namespace pcl
