// ORACLE — TEST INFRASTRUCTURE ONLY.  NOT PCL: pcl::VoxelGrid's interface over the ORACLE'S OWN restatement of its published
// algorithm (oracle_cloud.hpp voxel_grid).  When the reference's BasicScanRegistration.cpp is compiled against this header,
// everything in it is the reference's code EXCEPT the down-sampling of the less-flat cloud, which is the oracle's — tests must
// not read agreement on that step as pinning.
#pragma once
#include <pcl/point_cloud.h>
#include "../../../oracle_cloud.hpp"

namespace pcl {
template <class PointT> class VoxelGrid {
 public:
  void setInputCloud(const typename PointCloud<PointT>::Ptr& c) { in_ = c; }
  void setLeafSize(float lx, float, float) { leaf_ = lx; }   // the reference always passes three equal sizes
  void filter(PointCloud<PointT>& out) {
    loam_oracle::Cloud a, b;
    for (const PointT& p : in_->points) a.push_back({p.x, p.y, p.z, p.intensity});
    loam_oracle::voxel_grid(a, leaf_, b);
    out.clear();
    for (const loam_oracle::Pt& q : b) { PointT p; p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.i; out.push_back(p); }
  }
 private:
  typename PointCloud<PointT>::Ptr in_;
  float leaf_ = 0.2f;
};
}  // namespace pcl
