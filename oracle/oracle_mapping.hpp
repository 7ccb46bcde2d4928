// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp for the rules and for what is / is not pinned by the reference's own code).
//
// Scan-to-map registration, restating BasicLaserMapping:
//   transform_associate_to_map -> src/lib/BasicLaserMapping.cpp:103-167
//   transform_update           -> :171-203 (IMU blend omitted: empty IMU history)
//   point_associate_to_map     -> :207-219        point_associate_tobe_mapped -> :223-231
//   process                    -> :266-599        create_downsized_map        -> :242-264
//   optimize                   -> :626-926
// The rolling 21x11x21 grid of 50 m cubes is kept as a window over cube contents; shifting the window moves the
// contents by one cube along the axis and clears the vacated layer, which is what the pointer-swap loops at
// :311-441 amount to.
// Extra, not in the reference: set_frozen_submap()/register_frozen() expose the correspondence + Gauss-Newton part
// (stack round trip, voxel down-sampling, optimize) against a caller-provided sub-map, which is the unit the batched
// multi-GPU mode shards.
#pragma once
#include <deque>
#include "oracle_odometry.hpp"

namespace loam_oracle {

struct MappingStats {
  int iterations = 0;        // iterations entered
  int lastSelNum = 0;        // rows selected in the last iteration entered
  int cornerDS = 0, surfDS = 0;
  int cornerFromMap = 0, surfFromMap = 0;
  bool degenerate = false;
  bool optimized = false;    // false when the early-return guard (:628) fired
};

class LaserMapping {
 public:
  float scanPeriod = 0.1f;
  size_t maxIterations = 10;
  float deltaTAbort = 0.05f, deltaRAbort = 0.05f;
  float cornerLeaf = 0.2f, surfLeaf = 0.4f, mapLeaf = 0.f;
  static constexpr int W = 21, H = 11, D = 21;
  int cenW = 10, cenH = 5, cenD = 10;
  long frameCount = 0, mapFrameCount = 4;   // stackFrameNum-1 = 0, mapFrameNum-1 = 4 (:80-81)
  bool downsizedMapCreated = false;

  Cloud cornerLast, surfLast, fullRes;                 // inputs
  Cloud surround, surroundDS, cornerFromMap, surfFromMap, cornerStackDS, surfStackDS;
  std::vector<Cloud> cornerArray, surfArray;
  std::vector<size_t> validInd, surroundInd;
  Twist transformSum, transformIncre, transformTobeMapped, transformBefMapped, transformAftMapped;
  MappingStats stats;

  LaserMapping() : cornerArray(W * H * D), surfArray(W * H * D) {}

  void update_odometry(const float* t6) {
    transformSum.rot_x = t6[0]; transformSum.rot_y = t6[1]; transformSum.rot_z = t6[2];
    transformSum.pos = {t6[3], t6[4], t6[5]};
  }

  void transform_associate_to_map() {
    transformIncre.pos = transformBefMapped.pos - transformSum.pos;
    rotateYXZ(transformIncre.pos, -(transformSum.rot_y), -(transformSum.rot_x), -(transformSum.rot_z));

    float sbcx = transformSum.rot_x.sin(), cbcx = transformSum.rot_x.cos();
    float sbcy = transformSum.rot_y.sin(), cbcy = transformSum.rot_y.cos();
    float sbcz = transformSum.rot_z.sin(), cbcz = transformSum.rot_z.cos();
    float sblx = transformBefMapped.rot_x.sin(), cblx = transformBefMapped.rot_x.cos();
    float sbly = transformBefMapped.rot_y.sin(), cbly = transformBefMapped.rot_y.cos();
    float sblz = transformBefMapped.rot_z.sin(), cblz = transformBefMapped.rot_z.cos();
    float salx = transformAftMapped.rot_x.sin(), calx = transformAftMapped.rot_x.cos();
    float saly = transformAftMapped.rot_y.sin(), caly = transformAftMapped.rot_y.cos();
    float salz = transformAftMapped.rot_z.sin(), calz = transformAftMapped.rot_z.cos();

    float srx = -sbcx * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz) -
                cbcx * sbcy * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                               calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) -
                cbcx * cbcy * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                               calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx);
    transformTobeMapped.rot_x = -std::asin(srx);

    float srycrx = sbcx * (cblx * cblz * (caly * salz - calz * salx * saly) - cblx * sblz * (caly * calz + salx * saly * salz) +
                           calx * saly * sblx) -
                   cbcx * cbcy * ((caly * calz + salx * saly * salz) * (cblz * sbly - cbly * sblx * sblz) +
                                  (caly * salz - calz * salx * saly) * (sbly * sblz + cbly * cblz * sblx) - calx * cblx * cbly * saly) +
                   cbcx * sbcy * ((caly * calz + salx * saly * salz) * (cbly * cblz + sblx * sbly * sblz) +
                                  (caly * salz - calz * salx * saly) * (cbly * sblz - cblz * sblx * sbly) + calx * cblx * saly * sbly);
    float crycrx = sbcx * (cblx * sblz * (calz * saly - caly * salx * salz) - cblx * cblz * (saly * salz + caly * calz * salx) +
                           calx * caly * sblx) +
                   cbcx * cbcy * ((saly * salz + caly * calz * salx) * (sbly * sblz + cbly * cblz * sblx) +
                                  (calz * saly - caly * salx * salz) * (cblz * sbly - cbly * sblx * sblz) + calx * caly * cblx * cbly) -
                   cbcx * sbcy * ((saly * salz + caly * calz * salx) * (cbly * sblz - cblz * sblx * sbly) +
                                  (calz * saly - caly * salx * salz) * (cbly * cblz + sblx * sbly * sblz) - calx * caly * cblx * sbly);
    transformTobeMapped.rot_y = std::atan2(srycrx / transformTobeMapped.rot_x.cos(), crycrx / transformTobeMapped.rot_x.cos());

    float srzcrx = (cbcz * sbcy - cbcy * sbcx * sbcz) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                                                         calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx) -
                   (cbcy * cbcz + sbcx * sbcy * sbcz) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                                                         calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) +
                   cbcx * sbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    float crzcrx = (cbcy * sbcz - cbcz * sbcx * sbcy) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                                                         calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) -
                   (sbcy * sbcz + cbcy * cbcz * sbcx) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                                                         calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx) +
                   cbcx * cbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    transformTobeMapped.rot_z = std::atan2(srzcrx / transformTobeMapped.rot_x.cos(), crzcrx / transformTobeMapped.rot_x.cos());

    Vec3 v = transformIncre.pos;
    rotateZXY(v, transformTobeMapped.rot_z, transformTobeMapped.rot_x, transformTobeMapped.rot_y);
    transformTobeMapped.pos = transformAftMapped.pos - v;
  }

  // IMUState2 (BasicLaserMapping.h:47-75) history; stamps and laserOdometryTime in seconds (the reference's Time /
  // toSec arithmetic is double seconds, time_utils.h)
  struct ImuState2 {
    double stamp = 0;
    Angle roll, pitch;
  };
  std::deque<ImuState2> imuHistory;   // CircularBuffer<IMUState2>, capacity 200 (BasicLaserMapping.cpp:56)
  size_t imuCapacity = 200;
  double laserOdometryTime = 0;
  void update_imu(double stamp, float roll, float pitch) {   // :602-605 (+ CircularBuffer::push, CircularBuffer.h:111-119)
    if (imuHistory.size() >= imuCapacity) imuHistory.pop_front();
    ImuState2 st;
    st.stamp = stamp; st.roll = Angle(roll); st.pitch = Angle(pitch);
    imuHistory.push_back(st);
  }

  void transform_update() {   // :171-203
    if (0 < imuHistory.size()) {
      size_t imuIdx = 0;
      while (imuIdx < imuHistory.size() - 1 && (laserOdometryTime - imuHistory[imuIdx].stamp) + scanPeriod > 0) imuIdx++;
      ImuState2 imuCur;
      if (imuIdx == 0 || (laserOdometryTime - imuHistory[imuIdx].stamp) + scanPeriod > 0) {
        imuCur = imuHistory[imuIdx];   // scan time newer than the newest or older than the oldest IMU message
      } else {
        float ratio = ((imuHistory[imuIdx].stamp - laserOdometryTime) - scanPeriod) / (imuHistory[imuIdx].stamp - imuHistory[imuIdx - 1].stamp);
        float invRatio = 1 - ratio;   // IMUState2::interpolate(hist[idx], hist[idx-1], ratio, imuCur), .h:65-74
        imuCur.roll = Angle(imuHistory[imuIdx].roll.rad() * invRatio + imuHistory[imuIdx - 1].roll.rad() * ratio);
        imuCur.pitch = Angle(imuHistory[imuIdx].pitch.rad() * invRatio + imuHistory[imuIdx - 1].pitch.rad() * ratio);
      }
      transformTobeMapped.rot_x = Angle((float)(0.998 * transformTobeMapped.rot_x.rad() + 0.002 * imuCur.pitch.rad()));
      transformTobeMapped.rot_z = Angle((float)(0.998 * transformTobeMapped.rot_z.rad() + 0.002 * imuCur.roll.rad()));
    }
    transformBefMapped = transformSum;
    transformAftMapped = transformTobeMapped;
  }

  void point_associate_to_map(const Pt& pi, Pt& po) const {
    po = pi;
    rotateZXY(po, transformTobeMapped.rot_z, transformTobeMapped.rot_x, transformTobeMapped.rot_y);
    po.x += transformTobeMapped.pos.x;
    po.y += transformTobeMapped.pos.y;
    po.z += transformTobeMapped.pos.z;
  }
  void point_associate_tobe_mapped(const Pt& pi, Pt& po) const {
    po.x = pi.x - transformTobeMapped.pos.x;
    po.y = pi.y - transformTobeMapped.pos.y;
    po.z = pi.z - transformTobeMapped.pos.z;
    po.i = pi.i;
    rotateYXZ(po, -transformTobeMapped.rot_y, -transformTobeMapped.rot_x, -transformTobeMapped.rot_z);
  }

  static size_t to_index(int i, int j, int k) { return (size_t)i + (size_t)W * j + (size_t)W * H * k; }

  // cube index of a map-frame coordinate (:303-309, :540-546): double arithmetic, truncation + negative fix-up
  static int cube_of(float v, int cen) {
    const double CUBE_SIZE = 50.0, CUBE_HALF = CUBE_SIZE / 2;
    int c = int((v + CUBE_HALF) / CUBE_SIZE) + cen;
    if (v + CUBE_HALF < 0) c--;
    return c;
  }

  bool process() {
    frameCount++;
    if (frameCount < 1) return false;
    frameCount = 0;

    Pt pointSel;
    transform_associate_to_map();

    Cloud cornerStack, surfStack;
    for (const Pt& pt : cornerLast) { point_associate_to_map(pt, pointSel); cornerStack.push_back(pointSel); }
    for (const Pt& pt : surfLast) { point_associate_to_map(pt, pointSel); surfStack.push_back(pointSel); }

    Pt pointOnYAxis{0.f, 10.f, 0.f, 0.f};
    point_associate_to_map(pointOnYAxis, pointOnYAxis);

    int ci = cube_of(transformTobeMapped.pos.x, cenW);
    int cj = cube_of(transformTobeMapped.pos.y, cenH);
    int ck = cube_of(transformTobeMapped.pos.z, cenD);

    while (ci < 3) { shift(0, +1); ci++; cenW++; }
    while (ci >= W - 3) { shift(0, -1); ci--; cenW--; }
    while (cj < 3) { shift(1, +1); cj++; cenH++; }
    while (cj >= H - 3) { shift(1, -1); cj--; cenH--; }
    while (ck < 3) { shift(2, +1); ck++; cenD++; }
    while (ck >= D - 3) { shift(2, -1); ck--; cenD--; }

    select_cubes(ci, cj, ck, pointOnYAxis);

    cornerFromMap.clear();
    surfFromMap.clear();
    for (size_t ind : validInd) {
      cornerFromMap.insert(cornerFromMap.end(), cornerArray[ind].begin(), cornerArray[ind].end());
      surfFromMap.insert(surfFromMap.end(), surfArray[ind].begin(), surfArray[ind].end());
    }

    for (Pt& pt : cornerStack) point_associate_tobe_mapped(pt, pt);
    for (Pt& pt : surfStack) point_associate_tobe_mapped(pt, pt);

    voxel_grid(cornerStack, cornerLeaf, cornerStackDS);
    voxel_grid(surfStack, surfLeaf, surfStackDS);

    optimize();

    for (const Pt& p : cornerStackDS) {
      point_associate_to_map(p, pointSel);
      int I = cube_of(pointSel.x, cenW), J = cube_of(pointSel.y, cenH), K = cube_of(pointSel.z, cenD);
      if (I >= 0 && I < W && J >= 0 && J < H && K >= 0 && K < D) cornerArray[to_index(I, J, K)].push_back(pointSel);
    }
    for (const Pt& p : surfStackDS) {
      point_associate_to_map(p, pointSel);
      int I = cube_of(pointSel.x, cenW), J = cube_of(pointSel.y, cenH), K = cube_of(pointSel.z, cenD);
      if (I >= 0 && I < W && J >= 0 && J < H && K >= 0 && K < D) surfArray[to_index(I, J, K)].push_back(pointSel);
    }

    Cloud tmp;
    for (size_t ind : validInd) {
      voxel_grid(cornerArray[ind], cornerLeaf, tmp);
      cornerArray[ind].swap(tmp);
      voxel_grid(surfArray[ind], surfLeaf, tmp);
      surfArray[ind].swap(tmp);
    }

    for (Pt& pt : fullRes) point_associate_to_map(pt, pt);
    downsizedMapCreated = create_downsized_map();
    return true;
  }

  bool create_downsized_map() {
    mapFrameCount++;
    if (mapFrameCount < 5) return false;
    mapFrameCount = 0;
    surround.clear();
    for (size_t ind : surroundInd) {
      surround.insert(surround.end(), cornerArray[ind].begin(), cornerArray[ind].end());
      surround.insert(surround.end(), surfArray[ind].begin(), surfArray[ind].end());
    }
    voxel_grid(surround, cornerLeaf, surroundDS);   // the corner filter, not the map filter (:261)
    return true;
  }

  // ---- batched-mode unit: register one sweep's features against a caller-provided (frozen) sub-map.
  void set_frozen_submap(const Cloud& corner, const Cloud& surf) {
    cornerFromMap = corner;
    surfFromMap = surf;
    kdCorner_.build(&cornerFromMap);
    kdSurf_.build(&surfFromMap);
    frozen_ = true;
  }
  // guess6 = initial transformTobeMapped (rx, ry, rz, x, y, z); result in transformTobeMapped / transformAftMapped.
  void register_frozen(const float* guess6) {
    transformTobeMapped.rot_x = guess6[0]; transformTobeMapped.rot_y = guess6[1]; transformTobeMapped.rot_z = guess6[2];
    transformTobeMapped.pos = {guess6[3], guess6[4], guess6[5]};
    Pt pointSel;
    Cloud cornerStack, surfStack;
    for (const Pt& pt : cornerLast) { point_associate_to_map(pt, pointSel); cornerStack.push_back(pointSel); }
    for (const Pt& pt : surfLast) { point_associate_to_map(pt, pointSel); surfStack.push_back(pointSel); }
    for (Pt& pt : cornerStack) point_associate_tobe_mapped(pt, pt);
    for (Pt& pt : surfStack) point_associate_tobe_mapped(pt, pt);
    voxel_grid(cornerStack, cornerLeaf, cornerStackDS);
    voxel_grid(surfStack, surfLeaf, surfStackDS);
    optimize();
  }

  // Residual rows of the current transformTobeMapped (one pass of :665-817); exposed for tests.
  void residual_pass(std::vector<Pt>& ori, std::vector<Pt>& coeffs) {
    ori.clear();
    coeffs.clear();
    Pt pointSel, coeff;
    int sInd[5];
    float sDis[5];
    for (const Pt& pointOri : cornerStackDS) {
      point_associate_to_map(pointOri, pointSel);
      kdCorner_.knn(pointSel, 5, sInd, sDis);
      if (sDis[4] < 1.0) {
        float cx = 0.f, cy = 0.f, cz = 0.f;
        for (int j = 0; j < 5; j++) { cx += cornerFromMap[sInd[j]].x; cy += cornerFromMap[sInd[j]].y; cz += cornerFromMap[sInd[j]].z; }
        cx /= 5.0f; cy /= 5.0f; cz /= 5.0f;
        float m[9] = {0};
        for (int j = 0; j < 5; j++) {
          float ax = cornerFromMap[sInd[j]].x - cx, ay = cornerFromMap[sInd[j]].y - cy, az = cornerFromMap[sInd[j]].z - cz;
          m[0] += ax * ax; m[3] += ax * ay; m[6] += ax * az; m[4] += ay * ay; m[7] += ay * az; m[8] += az * az;
        }
        for (int k = 0; k < 9; k++) m[k] = m[k] / 5.0f;
        float w[3], V[9];
        eig_sym_jacobi<3>(m, w, V);
        if (w[2] > 3 * w[1]) {
          float x0 = pointSel.x, y0 = pointSel.y, z0 = pointSel.z;
          float x1 = (float)(cx + 0.1 * V[0 * 3 + 2]), y1 = (float)(cy + 0.1 * V[1 * 3 + 2]), z1 = (float)(cz + 0.1 * V[2 * 3 + 2]);
          float x2 = (float)(cx - 0.1 * V[0 * 3 + 2]), y2 = (float)(cy - 0.1 * V[1 * 3 + 2]), z2 = (float)(cz - 0.1 * V[2 * 3 + 2]);
          float a012 = std::sqrt(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                                 ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                                 ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
          float l12 = std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
          float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                      (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) / a012 / l12;
          float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) -
                       (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
          float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                       (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
          float ld2 = a012 / l12;
          float s = 1 - 0.9f * std::fabs(ld2);
          coeff.x = s * la; coeff.y = s * lb; coeff.z = s * lc; coeff.i = s * ld2;
          if (s > 0.1) { ori.push_back(pointOri); coeffs.push_back(coeff); }
        }
      }
    }
    for (const Pt& pointOri : surfStackDS) {
      point_associate_to_map(pointOri, pointSel);
      kdSurf_.knn(pointSel, 5, sInd, sDis);
      if (sDis[4] < 1.0) {
        float A0[15], B0[5] = {-1.f, -1.f, -1.f, -1.f, -1.f}, X0[3];
        for (int j = 0; j < 5; j++) {
          A0[j * 3 + 0] = surfFromMap[sInd[j]].x;
          A0[j * 3 + 1] = surfFromMap[sInd[j]].y;
          A0[j * 3 + 2] = surfFromMap[sInd[j]].z;
        }
        colpiv_qr_solve<5, 3>(A0, B0, X0);
        float pa = X0[0], pb = X0[1], pc = X0[2], pd = 1;
        float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
        pa /= ps; pb /= ps; pc /= ps; pd /= ps;
        bool planeValid = true;
        for (int j = 0; j < 5; j++) {
          if (std::fabs(pa * surfFromMap[sInd[j]].x + pb * surfFromMap[sInd[j]].y + pc * surfFromMap[sInd[j]].z + pd) > 0.2) {
            planeValid = false;
            break;
          }
        }
        if (planeValid) {
          float pd2 = pa * pointSel.x + pb * pointSel.y + pc * pointSel.z + pd;
          float s = 1 - 0.9f * std::fabs(pd2) / std::sqrt(pt_dist(pointSel));
          coeff.x = s * pa; coeff.y = s * pb; coeff.z = s * pc; coeff.i = s * pd2;
          if (s > 0.1) { ori.push_back(pointOri); coeffs.push_back(coeff); }
        }
      }
    }
  }

  void optimize() {
    stats = MappingStats();
    stats.cornerDS = (int)cornerStackDS.size();
    stats.surfDS = (int)surfStackDS.size();
    stats.cornerFromMap = (int)cornerFromMap.size();
    stats.surfFromMap = (int)surfFromMap.size();
    if (cornerFromMap.size() <= 10 || surfFromMap.size() <= 100) return;   // NB: skips transform_update (:628)
    stats.optimized = true;
    if (!frozen_) {
      kdCorner_.build(&cornerFromMap);
      kdSurf_.build(&surfFromMap);
    }
    bool isDegenerate = false;
    float matP[36];
    std::vector<Pt> ori, coeffs;
    std::vector<float> A, B;
    for (size_t iter = 0; iter < maxIterations; iter++) {
      stats.iterations = (int)iter + 1;
      residual_pass(ori, coeffs);
      float srx = transformTobeMapped.rot_x.sin(), crx = transformTobeMapped.rot_x.cos();
      float sry = transformTobeMapped.rot_y.sin(), cry = transformTobeMapped.rot_y.cos();
      float srz = transformTobeMapped.rot_z.sin(), crz = transformTobeMapped.rot_z.cos();
      const size_t selNum = ori.size();
      stats.lastSelNum = (int)selNum;
      if (selNum < 50) continue;
      A.resize(selNum * 6);
      B.resize(selNum);
      for (size_t i = 0; i < selNum; i++) {
        const Pt& po = ori[i];
        const Pt& co = coeffs[i];
        float arx = (crx * sry * srz * po.x + crx * crz * sry * po.y - srx * sry * po.z) * co.x +
                    (-srx * srz * po.x - crz * srx * po.y - crx * po.z) * co.y +
                    (crx * cry * srz * po.x + crx * cry * crz * po.y - cry * srx * po.z) * co.z;
        float ary = ((cry * srx * srz - crz * sry) * po.x + (sry * srz + cry * crz * srx) * po.y + crx * cry * po.z) * co.x +
                    ((-cry * crz - srx * sry * srz) * po.x + (cry * srz - crz * srx * sry) * po.y - crx * sry * po.z) * co.z;
        float arz = ((crz * srx * sry - cry * srz) * po.x + (-cry * crz - srx * sry * srz) * po.y) * co.x +
                    (crx * crz * po.x - crx * srz * po.y) * co.y +
                    ((sry * srz + cry * crz * srx) * po.x + (crz * sry - cry * srx * srz) * po.y) * co.z;
        float* a = &A[i * 6];
        a[0] = arx; a[1] = ary; a[2] = arz; a[3] = co.x; a[4] = co.y; a[5] = co.z;
        B[i] = -co.i;
      }
      NormalEq ne;
      accumulate_normal_eq(A, B, ne);
      float X[6];
      colpiv_qr_solve<6, 6>(ne.AtA, ne.AtB, X);
      if (iter == 0) isDegenerate = degeneracy_projector(ne.AtA, 100.f, matP);
      stats.degenerate = isDegenerate;
      if (isDegenerate) {
        float X2[6];
        std::memcpy(X2, X, sizeof(X));
        for (int r = 0; r < 6; r++) {
          float s = 0.f;
          for (int c = 0; c < 6; c++) s += matP[r * 6 + c] * X2[c];
          X[r] = s;
        }
      }
      transformTobeMapped.rot_x += X[0];
      transformTobeMapped.rot_y += X[1];
      transformTobeMapped.rot_z += X[2];
      transformTobeMapped.pos.x += X[3];
      transformTobeMapped.pos.y += X[4];
      transformTobeMapped.pos.z += X[5];
      float deltaR = std::sqrt(std::pow(rad2deg_f(X[0]), 2) + std::pow(rad2deg_f(X[1]), 2) + std::pow(rad2deg_f(X[2]), 2));
      float deltaT = std::sqrt(std::pow(X[3] * 100, 2) + std::pow(X[4] * 100, 2) + std::pow(X[5] * 100, 2));
      if (deltaR < deltaRAbort && deltaT < deltaTAbort) break;
    }
    transform_update();
  }

 private:
  KdTree kdCorner_, kdSurf_;
  bool frozen_ = false;

  // move cube contents by one along axis (0=i,1=j,2=k); dir=+1 moves content towards higher indices and clears index 0.
  void shift(int axis, int dir) {
    const int n[3] = {W, H, D};
    auto idx = [&](int a, int b, int c) {   // a along axis, (b,c) the other two in i,j,k order
      int ijk[3];
      int o = 0;
      for (int t = 0; t < 3; t++) ijk[t] = (t == axis) ? a : (o++ == 0 ? b : c);
      return to_index(ijk[0], ijk[1], ijk[2]);
    };
    int o1 = axis == 0 ? 1 : 0, o2 = axis == 2 ? 1 : 2;
    for (int b = 0; b < n[o1]; b++)
      for (int c = 0; c < n[o2]; c++) {
        if (dir > 0) {
          for (int a = n[axis] - 1; a >= 1; a--) {
            cornerArray[idx(a, b, c)].swap(cornerArray[idx(a - 1, b, c)]);
            surfArray[idx(a, b, c)].swap(surfArray[idx(a - 1, b, c)]);
          }
          cornerArray[idx(0, b, c)].clear();
          surfArray[idx(0, b, c)].clear();
        } else {
          for (int a = 0; a < n[axis] - 1; a++) {
            cornerArray[idx(a, b, c)].swap(cornerArray[idx(a + 1, b, c)]);
            surfArray[idx(a, b, c)].swap(surfArray[idx(a + 1, b, c)]);
          }
          cornerArray[idx(n[axis] - 1, b, c)].clear();
          surfArray[idx(n[axis] - 1, b, c)].clear();
        }
      }
  }

  // 5x5x5 neighbourhood + field-of-view test on the 8 cube corners (:443-500)
  void select_cubes(int ci, int cj, int ck, const Pt& pointOnYAxis) {
    validInd.clear();
    surroundInd.clear();
    Pt tpos{transformTobeMapped.pos.x, transformTobeMapped.pos.y, transformTobeMapped.pos.z, 0.f};
    for (int i = ci - 2; i <= ci + 2; i++)
      for (int j = cj - 2; j <= cj + 2; j++)
        for (int k = ck - 2; k <= ck + 2; k++) {
          if (i < 0 || i >= W || j < 0 || j >= H || k < 0 || k >= D) continue;
          float centerX = 50.0f * (i - cenW), centerY = 50.0f * (j - cenH), centerZ = 50.0f * (k - cenD);
          bool inFOV = false;
          for (int ii = -1; ii <= 1; ii += 2)
            for (int jj = -1; jj <= 1; jj += 2)
              for (int kk = -1; kk <= 1; kk += 2) {
                Pt corner{centerX + 25.0f * ii, centerY + 25.0f * jj, centerZ + 25.0f * kk, 0.f};
                float s1 = sq_diff(tpos, corner), s2 = sq_diff(pointOnYAxis, corner);
                float check1 = 100.0f + s1 - s2 - 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
                float check2 = 100.0f + s1 - s2 + 10.0f * std::sqrt(3.0f) * std::sqrt(s1);
                if (check1 < 0 && check2 > 0) inFOV = true;
              }
          size_t cubeIdx = to_index(i, j, k);
          if (inFOV) validInd.push_back(cubeIdx);
          surroundInd.push_back(cubeIdx);
        }
  }
};

}  // namespace loam_oracle
