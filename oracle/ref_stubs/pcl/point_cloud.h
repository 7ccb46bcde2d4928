// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT PCL: a std::vector with the handful of members the reference's
// sources touch (push_back, size, clear, operator[], operator+=, the (width, height) constructor, Ptr),
// so that the reference's feature extraction and IMU bookkeeping compile where they lie (oracle/Makefile target `ref`).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
#include <pcl/point_types.h>

namespace pcl {
template <class PointT> class PointCloud {
 public:
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  uint32_t width = 0, height = 0;
  PointCloud() = default;
  PointCloud(uint32_t w, uint32_t h, const PointT& value = PointT()) : points((size_t)w * h, value), width(w), height(h) {}
  void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = height = 0; }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  PointCloud& operator+=(const PointCloud& o) {
    points.insert(points.end(), o.points.begin(), o.points.end());
    width = (uint32_t)points.size(); height = 1;
    return *this;
  }
  typename std::vector<PointT>::iterator begin() { return points.begin(); }
  typename std::vector<PointT>::iterator end() { return points.end(); }
  typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
  typename std::vector<PointT>::const_iterator end() const { return points.end(); }
};
}  // namespace pcl
