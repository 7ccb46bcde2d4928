// Drives the three adapter classes the way the reference's node loop does
// (MultiScanRegistration::process -> LaserOdometry::process/publishResult -> LaserMapping::process), on sweeps read from
// a binary file written by tests/test_gpu_adapter.py:
//   int32 n_sweeps; per sweep: int32 n_rings, int32 ring_size[n_rings], float32 xyzi[sum][4]
// Prints one line per sweep: "k transformSum[6] transformAftMapped[6]".
#include <cmath>
#include <cstdio>
#include <chrono>
#include "loam_velodyne/loamx_adapter.h"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: adapter_test sweeps.bin\n"); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  int32_t n_sweeps = 0;
  if (std::fread(&n_sweeps, 4, 1, f) != 1) return 2;
  try {
    loam::BasicScanRegistration scanReg;
    scanReg.configure();
    loam::BasicLaserOdometry odom;
    loam::BasicLaserMapping mapping;
    loam::BasicTransformMaintenance maintenance;   // fed like TransformMaintenance.cpp:66-115 feeds the original
    for (int k = 0; k < n_sweeps; k++) {
      int32_t n_rings = 0;
      if (std::fread(&n_rings, 4, 1, f) != 1) return 2;
      std::vector<int32_t> sizes(n_rings);
      if (std::fread(sizes.data(), 4, n_rings, f) != (size_t)n_rings) return 2;
      std::vector<loam::CloudXYZI> rings(n_rings);
      for (int r = 0; r < n_rings; r++) {
        std::vector<float> buf((size_t)sizes[r] * 4);
        if (sizes[r] && std::fread(buf.data(), 16, sizes[r], f) != (size_t)sizes[r]) return 2;
        rings[r].points.resize(sizes[r]);
        for (int i = 0; i < sizes[r]; i++) {
          auto& p = rings[r].points[i];
          p.x = buf[4 * i]; p.y = buf[4 * i + 1]; p.z = buf[4 * i + 2]; p.intensity = buf[4 * i + 3];
        }
      }
      scanReg.processScanlines(std::chrono::system_clock::now(), rings);
      if (k == 0) {
        // the raw entry point (MultiScanRegistration::process): the same sweep as a driver delivers it — sensor axes, firing
        // order — must give the same ring sizes and the same picks as the pre-binned call above (VLP-16 mapper)
        bool equal_rings = true;
        for (int r = 1; r < n_rings; r++) equal_rings = equal_rings && sizes[r] == sizes[0];
        if (equal_rings && n_rings == 16) {
          loamx_pcl::PointCloud<loamx_pcl::PointXYZ> raw;
          raw.points.resize((size_t)n_rings * sizes[0]);
          for (int a = 0; a < sizes[0]; a++)
            for (int r = 0; r < n_rings; r++) {
              const auto& p = rings[r].points[a];
              auto& q = raw.points[(size_t)a * n_rings + r];
              q.x = p.z; q.y = p.x; q.z = p.y;   // inverse of the axis remap at MultiScanRegistration.cpp:184-186
            }
          loam::BasicScanRegistration rawReg;
          rawReg.configure();
          rawReg.processRawSweep(std::chrono::system_clock::now(), raw, -15.f, 15.f, 16);
          bool ok = rawReg.laserCloud().size() == scanReg.laserCloud().size() &&
                    rawReg.cornerPointsSharp().size() == scanReg.cornerPointsSharp().size() &&
                    rawReg.surfacePointsFlat().size() == scanReg.surfacePointsFlat().size();
          for (int r = 0; ok && r < n_rings; r++) ok = rawReg.ringSizes()[r] == (uint32_t)sizes[r];
          for (size_t i = 0; ok && i < rawReg.cornerPointsSharp().size(); i++)
            ok = rawReg.cornerPointsSharp().points[i].x == scanReg.cornerPointsSharp().points[i].x &&
                 rawReg.cornerPointsSharp().points[i].z == scanReg.cornerPointsSharp().points[i].z;
          if (!ok) { std::fprintf(stderr, "error: raw-sweep ingestion disagrees with the pre-binned path\n"); return 3; }
        }
      }
      *odom.cornerPointsSharp() = scanReg.cornerPointsSharp();
      *odom.cornerPointsLessSharp() = scanReg.cornerPointsLessSharp();
      *odom.surfPointsFlat() = scanReg.surfacePointsFlat();
      *odom.surfPointsLessFlat() = scanReg.surfacePointsLessFlat();
      *odom.laserCloud() = scanReg.laserCloud();
      odom.process();
      odom.transformToEnd(odom.laserCloud());
      mapping.laserCloudCornerLast() = *odom.lastCornerCloud();
      mapping.laserCloudSurfLast() = *odom.lastSurfaceCloud();
      mapping.laserCloud() = *odom.laserCloud();
      mapping.updateOdometry(odom.transformSum());
      mapping.process(std::chrono::system_clock::now());
      float s[6], a[6];
      loam::detail::twist_to(odom.transformSum(), s);
      loam::detail::twist_to(mapping.transformAftMapped(), a);
      maintenance.updateOdometry(s[0], s[1], s[2], s[3], s[4], s[5]);
      maintenance.updateMappingTransform(mapping.transformAftMapped(), mapping.transformBefMapped());
      maintenance.transformAssociateToMap();
      // right after a mapping update transformBefMapped == transformSum, so the fused pose is the mapping result
      for (int c = 0; c < 6; c++)
        if (std::fabs(maintenance.transformMapped()[c] - a[c]) > 1e-5f) { std::fprintf(stderr, "error: transform maintenance disagrees with the mapping pose\n"); return 4; }
      std::printf("%d %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", k, s[0], s[1], s[2], s[3], s[4], s[5], a[0], a[1], a[2],
                  a[3], a[4], a[5]);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
