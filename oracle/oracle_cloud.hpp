// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp for the rules and for what is / is not pinned by the reference's own code).
//
// Point-cloud primitives whose arithmetic lives OUTSIDE /root/reference:
//   voxel_grid  -> pcl::VoxelGrid<PointXYZI>::filter as called at BasicLaserMapping.cpp:261-262,
//                  520-521, 525-526, 583-588 and BasicScanRegistration.cpp:246-250.  PCL is an un-vendored,
//                  un-pinned dependency (CMakeLists.txt:15; README.md:6 => PCL 1.7.x/1.8).  Restated from the
//                  published algorithm of filters/impl/voxel_grid.hpp (PCL 1.7/1.8):
//                    inverse_leaf = 1/leaf (float); bounding box of the finite points;
//                    min_b = floor(min*inverse_leaf), max_b likewise, div_b = max_b-min_b+1;
//                    if dx*dy*dz > INT_MAX: warn and copy the input through;
//                    voxel id = ix + iy*div_x + iz*div_x*div_y with i* = floor(p*inverse_leaf) - min_b;
//                    sort by id; one output per occupied voxel = float mean of x,y,z AND intensity
//                    (downsample_all_data = true), emitted in ascending id.
//                  PCL sorts with the unstable std::sort, so the summation order inside a voxel is
//                  implementation-defined there; this restatement fixes it to input order (stable sort).
//   KdTree      -> the kNN contract of nanoflann_pcl.h:131-152 over nanoflann.hpp (vendored 1.2.3):
//                  full rebuild per setInputCloud, leaf <= 10 points, split on the widest bounding-box
//                  dimension at the clamped box middle (nanoflann.hpp:916-1043), exact search with
//                  per-dimension lower bounds (:1354-1412), float L2^2 accumulated x->y->z (:372-379),
//                  k results ascending, strict '>' insertion so ties keep the first point visited (:115-139),
//                  dists[k-1] = FLT_MAX when fewer than k points exist (:96-97).
//                  Own array-based layout (no pooled pointer nodes).  tests/ pins it against the real
//                  nanoflann.hpp via oracle/_ref.
#pragma once
#include "oracle_math.hpp"
#include <numeric>

namespace loam_oracle {

inline bool pt_finite(const Pt& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); }

inline void voxel_grid(const Cloud& in, float leaf, Cloud& out) {
  out.clear();
  if (in.empty()) return;
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  size_t nfinite = 0;
  for (const Pt& p : in) {
    if (!pt_finite(p)) continue;
    nfinite++;
    mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
    mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
    mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
  }
  if (nfinite == 0) return;
  int64_t d[3];
  int minb[3], maxb[3];
  for (int a = 0; a < 3; a++) {
    minb[a] = (int)std::floor(mn[a] * inv);
    maxb[a] = (int)std::floor(mx[a] * inv);
    d[a] = (int64_t)std::floor(mx[a] * inv) - (int64_t)std::floor(mn[a] * inv) + 1;
  }
  if ((__int128)d[0] * d[1] * d[2] > (__int128)std::numeric_limits<int32_t>::max()) {   // (128-bit: PCL's own 64-bit product can wrap)
    out = in;  // "Leaf size is too small for the input dataset" -> input copied through
    return;
  }
  const int divx = maxb[0] - minb[0] + 1, divy = maxb[1] - minb[1] + 1;
  const int mul1 = divx, mul2 = divx * divy;
  std::vector<std::pair<int, uint32_t>> keys;
  keys.reserve(in.size());
  for (uint32_t k = 0; k < in.size(); k++) {
    const Pt& p = in[k];
    if (!pt_finite(p)) continue;
    int i0 = (int)(std::floor(p.x * inv) - (float)minb[0]);
    int i1 = (int)(std::floor(p.y * inv) - (float)minb[1]);
    int i2 = (int)(std::floor(p.z * inv) - (float)minb[2]);
    keys.emplace_back(i0 + i1 * mul1 + i2 * mul2, k);
  }
  std::stable_sort(keys.begin(), keys.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  size_t a = 0;
  while (a < keys.size()) {
    size_t b = a;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    while (b < keys.size() && keys[b].first == keys[a].first) {
      const Pt& p = in[keys[b].second];
      sx += p.x; sy += p.y; sz += p.z; si += p.i;
      b++;
    }
    float n = (float)(b - a);
    out.push_back({sx / n, sy / n, sz / n, si / n});
    a = b;
  }
}

class KdTree {
 public:
  void build(const Cloud* cloud) {
    pts_ = cloud;
    nodes_.clear();
    vind_.resize(cloud->size());
    std::iota(vind_.begin(), vind_.end(), 0);
    if (cloud->empty()) return;
    for (int a = 0; a < 3; a++) { root_lo_[a] = FLT_MAX; root_hi_[a] = -FLT_MAX; }
    for (const Pt& p : *cloud) {
      const float v[3] = {p.x, p.y, p.z};
      for (int a = 0; a < 3; a++) { root_lo_[a] = std::min(root_lo_[a], v[a]); root_hi_[a] = std::max(root_hi_[a], v[a]); }
    }
    nodes_.reserve(cloud->size() / 4 + 16);
    float lo[3] = {root_lo_[0], root_lo_[1], root_lo_[2]}, hi[3] = {root_hi_[0], root_hi_[1], root_hi_[2]};
    divide(0, (int)cloud->size(), lo, hi);
  }
  size_t size() const { return pts_ ? pts_->size() : 0; }

  // k nearest; idx/d2 sized k.  Returns number found.  d2[k-1]==FLT_MAX if fewer than k points.
  int knn(const Pt& q, int k, int* idx, float* d2) const {
    for (int j = 0; j < k; j++) { idx[j] = 0; d2[j] = 0.f; }
    d2[k - 1] = FLT_MAX;
    if (nodes_.empty()) return 0;
    const float qv[3] = {q.x, q.y, q.z};
    float dists[3] = {0.f, 0.f, 0.f};
    float distsq = 0.f;
    for (int a = 0; a < 3; a++) {
      if (qv[a] < root_lo_[a]) { dists[a] = (qv[a] - root_lo_[a]) * (qv[a] - root_lo_[a]); distsq += dists[a]; }
      if (qv[a] > root_hi_[a]) { dists[a] = (qv[a] - root_hi_[a]) * (qv[a] - root_hi_[a]); distsq += dists[a]; }
    }
    int count = 0;
    search(0, qv, distsq, dists, k, idx, d2, count);
    return count;
  }

 private:
  struct Node {
    int left, right;   // leaf: point range in vind_; inner: children indices
    int feat;          // -1 for leaf
    float divlow, divhigh;
  };
  const Cloud* pts_ = nullptr;
  std::vector<Node> nodes_;
  std::vector<int> vind_;
  float root_lo_[3], root_hi_[3];

  float coord(int i, int a) const {
    const Pt& p = (*pts_)[i];
    return a == 0 ? p.x : (a == 1 ? p.y : p.z);
  }
  int divide(int left, int right, float* lo, float* hi) {
    int me = (int)nodes_.size();
    nodes_.push_back(Node());
    if (right - left <= 10) {
      nodes_[me].feat = -1;
      nodes_[me].left = left;
      nodes_[me].right = right;
      for (int a = 0; a < 3; a++) lo[a] = hi[a] = coord(vind_[left], a);
      // NB nanoflann.hpp:931 iterates k < right, i.e. it covers every leaf point (right is exclusive)
      for (int k = left + 1; k < right; k++)
        for (int a = 0; a < 3; a++) {
          float v = coord(vind_[k], a);
          if (lo[a] > v) lo[a] = v;
          if (hi[a] < v) hi[a] = v;
        }
      return me;
    }
    const int count = right - left;
    int* ind = vind_.data() + left;
    // choose the cut dimension: among the (nearly) widest box spans, the one with the widest data spread
    float max_span = hi[0] - lo[0];
    for (int a = 1; a < 3; a++) max_span = std::max(max_span, hi[a] - lo[a]);
    int cut = 0;
    float max_spread = -1.f;
    for (int a = 0; a < 3; a++) {
      if (hi[a] - lo[a] > (1.f - 0.00001f) * max_span) {
        float mn, mx;
        minmax(ind, count, a, mn, mx);
        if (mx - mn > max_spread) { cut = a; max_spread = mx - mn; }
      }
    }
    float split = (lo[cut] + hi[cut]) / 2;
    float mn, mx;
    minmax(ind, count, cut, mn, mx);
    float cutval = split < mn ? mn : (split > mx ? mx : split);
    int lim1, lim2;
    plane_split(ind, count, cut, cutval, lim1, lim2);
    int idx = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);

    float llo[3] = {lo[0], lo[1], lo[2]}, lhi[3] = {hi[0], hi[1], hi[2]};
    float rlo[3] = {lo[0], lo[1], lo[2]}, rhi[3] = {hi[0], hi[1], hi[2]};
    lhi[cut] = cutval;
    rlo[cut] = cutval;
    int c1 = divide(left, left + idx, llo, lhi);
    int c2 = divide(left + idx, right, rlo, rhi);
    nodes_[me].feat = cut;
    nodes_[me].left = c1;
    nodes_[me].right = c2;
    nodes_[me].divlow = lhi[cut];
    nodes_[me].divhigh = rlo[cut];
    for (int a = 0; a < 3; a++) { lo[a] = std::min(llo[a], rlo[a]); hi[a] = std::max(lhi[a], rhi[a]); }
    return me;
  }
  void minmax(const int* ind, int count, int a, float& mn, float& mx) const {
    mn = mx = coord(ind[0], a);
    for (int k = 1; k < count; k++) {
      float v = coord(ind[k], a);
      if (v < mn) mn = v;
      if (v > mx) mx = v;
    }
  }
  // three-way partition: [< cutval | == cutval | > cutval]
  void plane_split(int* ind, int count, int a, float cutval, int& lim1, int& lim2) const {
    int left = 0, right = count - 1;
    for (;;) {
      while (left <= right && coord(ind[left], a) < cutval) ++left;
      while (right && left <= right && coord(ind[right], a) >= cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left; --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;) {
      while (left <= right && coord(ind[left], a) <= cutval) ++left;
      while (right && left <= right && coord(ind[right], a) > cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left; --right;
    }
    lim2 = left;
  }
  static void add_point(float dist, int index, int k, int* idx, float* d2, int& count) {
    int i;
    for (i = count; i > 0; --i) {
      if (d2[i - 1] > dist) {
        if (i < k) { d2[i] = d2[i - 1]; idx[i] = idx[i - 1]; }
      } else
        break;
    }
    if (i < k) { d2[i] = dist; idx[i] = index; }
    if (count < k) count++;
  }
  void search(int n, const float* q, float mindistsq, float* dists, int k, int* idx, float* d2, int& count) const {
    const Node& node = nodes_[n];
    if (node.feat < 0) {
      const float worst = d2[k - 1];   // sampled once per leaf, as nanoflann.hpp:1361 does
      for (int i = node.left; i < node.right; i++) {
        const int pi = vind_[i];
        const Pt& p = (*pts_)[pi];
        float dx = q[0] - p.x, dy = q[1] - p.y, dz = q[2] - p.z;
        float dist = dx * dx + dy * dy + dz * dz;   // x -> y -> z accumulation, nanoflann.hpp:372-379
        if (dist < worst) add_point(dist, pi, k, idx, d2, count);
      }
      return;
    }
    const int a = node.feat;
    const float val = q[a];
    const float diff1 = val - node.divlow, diff2 = val - node.divhigh;
    int best, other;
    float cut_dist;
    if (diff1 + diff2 < 0) { best = node.left; other = node.right; cut_dist = diff2 * diff2; }
    else { best = node.right; other = node.left; cut_dist = diff1 * diff1; }
    search(best, q, mindistsq, dists, k, idx, d2, count);
    float dst = dists[a];
    mindistsq = mindistsq + cut_dist - dst;
    dists[a] = cut_dist;
    if (mindistsq <= d2[k - 1]) search(other, q, mindistsq, dists, k, idx, d2, count);
    dists[a] = dst;
  }
};

// Brute-force exact kNN used by tests as ground truth (ties: lowest index first).
inline int knn_brute(const Cloud& c, const Pt& q, int k, int* idx, float* d2) {
  std::vector<std::pair<float, int>> all(c.size());
  for (size_t i = 0; i < c.size(); i++) {
    float dx = q.x - c[i].x, dy = q.y - c[i].y, dz = q.z - c[i].z;
    all[i] = {dx * dx + dy * dy + dz * dz, (int)i};
  }
  int kk = std::min<int>(k, (int)c.size());
  std::partial_sort(all.begin(), all.begin() + kk, all.end());
  for (int j = 0; j < k; j++) { idx[j] = 0; d2[j] = 0.f; }
  d2[k - 1] = FLT_MAX;
  for (int j = 0; j < kk; j++) { idx[j] = all[j].second; d2[j] = all[j].first; }
  return kk;
}

}  // namespace loam_oracle
