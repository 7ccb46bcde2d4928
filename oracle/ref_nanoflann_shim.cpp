// ORACLE — TEST INFRASTRUCTURE ONLY.
// Thin C shim around the REFERENCE'S OWN vendored nanoflann.hpp (compiled from where it lies under
// /root/reference/include/loam_velodyne, never copied).  It builds the index the same way
// nanoflann_pcl.h:97-152 does (KDTreeSingleIndexAdaptor<SO3_Adaptor<float,...>, ..., 3, int>, default
// leaf size, KNNResultSet, default SearchParams) so the oracle's KdTree can be pinned against it.
#include <nanoflann.hpp>
#include <vector>

namespace {
struct Adaptor {
  const float* pts;   // n x 4 floats
  size_t n;
  inline size_t kdtree_get_point_count() const { return n; }
  inline float kdtree_get_pt(const size_t idx, int dim) const { return dim < 3 ? pts[4 * idx + dim] : 0.f; }
  template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};
typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::SO3_Adaptor<float, Adaptor>, Adaptor, 3, int> Tree;
}  // namespace

extern "C" void ref_knn(const float* pts, int n, const float* q, int nq, int k, int* idx, float* d2) {
  Adaptor a{pts, (size_t)n};
  Tree tree(3, a);
  tree.buildIndex();
  for (int i = 0; i < nq; i++) {
    nanoflann::KNNResultSet<float, int> rs(k);
    for (int j = 0; j < k; j++) { idx[(size_t)i * k + j] = 0; d2[(size_t)i * k + j] = 0.f; }
    rs.init(idx + (size_t)i * k, d2 + (size_t)i * k);
    tree.findNeighbors(rs, q + 4 * (size_t)i, nanoflann::SearchParams());
  }
}
