// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT PCL: removeNaNFromPointCloud as the reference's odometry calls it (in place, indices
// returned) — points with a non-finite x, y or z are dropped, order kept.
#pragma once
#include <cmath>
#include <vector>
#include <pcl/point_cloud.h>
namespace pcl {
template <class PointT> inline void removeNaNFromPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, std::vector<int>& index) {
  std::vector<PointT> kept;
  index.clear();
  for (size_t i = 0; i < in.points.size(); i++) {
    const PointT& p = in.points[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    kept.push_back(p);
    index.push_back((int)i);
  }
  out.points.swap(kept);
  out.width = (uint32_t)out.points.size();
  out.height = 1;
}
}  // namespace pcl
