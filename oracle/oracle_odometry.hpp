// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp for the rules and for what is / is not pinned by the reference's own code).
//
// Sweep-to-sweep registration, restating BasicLaserOdometry:
//   transform_to_start   -> src/lib/BasicLaserOdometry.cpp:40-53
//   transform_to_end     -> :57-87
//   plugin_imu_rotation  -> :91-151
//   accumulate_rotation  -> :155-179
//   update_imu           -> :181-194
//   process              -> :196-666
// Quirks kept on purpose (SURVEY.md §8c checklist 8-11): the forward ring scans are bounded by the CURRENT
// feature counts (:262, :378); the raw, not de-skewed, point goes into laserCloudOri (:358, :478); x1.05 on the
// accumulated yaw increment and z translation (:631, :637); transformToEnd truncates intensity (:70).
#pragma once
#include "oracle_cloud.hpp"

namespace loam_oracle {

// 6 unknowns solve shared with mapping: AtA (float, row-order accumulation), colpiv QR.
struct NormalEq {
  float AtA[36];
  float AtB[6];
};
inline void accumulate_normal_eq(const std::vector<float>& A, const std::vector<float>& B, NormalEq& ne) {
  // matAtA = matAt * matA, matAtB = matAt * matB in float; plain row-order accumulation (Eigen's blocked
  // product order is not reproducible without Eigen; see DESIGN.md "float divergence").
  const size_t n = B.size();
  for (int k = 0; k < 36; k++) ne.AtA[k] = 0.f;
  for (int k = 0; k < 6; k++) ne.AtB[k] = 0.f;
  for (size_t r = 0; r < n; r++) {
    const float* a = &A[r * 6];
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) ne.AtA[i * 6 + j] += a[i] * a[j];
      ne.AtB[i] += a[i] * B[r];
    }
  }
}

class LaserOdometry {
 public:
  float scanPeriod = 0.1f;
  size_t maxIterations = 25;
  float deltaTAbort = 0.1f, deltaRAbort = 0.1f;
  long frameCount = 0;
  bool systemInited = false;

  Cloud cornerSharp, cornerLessSharp, surfFlat, surfLessFlat, laserCloud;   // inputs (per sweep)
  Cloud lastCorner, lastSurf;
  Twist transform, transformSum;
  Angle imuRollStart, imuPitchStart, imuYawStart, imuRollEnd, imuPitchEnd, imuYawEnd;
  Vec3 imuShiftFromStart, imuVeloFromStart;
  int lastIterCount = 0;   // diagnostics: iterations executed in the last process()
  int lastSelNum = 0;

  void update_imu(const float* t12) {
    imuPitchStart = t12[0]; imuYawStart = t12[1]; imuRollStart = t12[2];
    imuPitchEnd = t12[3]; imuYawEnd = t12[4]; imuRollEnd = t12[5];
    imuShiftFromStart = {t12[6], t12[7], t12[8]};
    imuVeloFromStart = {t12[9], t12[10], t12[11]};
  }

  void transform_to_start(const Pt& pi, Pt& po) const {
    float s = (1.f / scanPeriod) * (pi.i - int(pi.i));
    po.x = pi.x - s * transform.pos.x;
    po.y = pi.y - s * transform.pos.y;
    po.z = pi.z - s * transform.pos.z;
    po.i = pi.i;
    Angle rx = -s * transform.rot_x.rad();
    Angle ry = -s * transform.rot_y.rad();
    Angle rz = -s * transform.rot_z.rad();
    rotateZXY(po, rz, rx, ry);
  }

  size_t transform_to_end(Cloud& cloud) const {
    for (Pt& p : cloud) {
      float s = (1.f / scanPeriod) * (p.i - int(p.i));
      p.x -= s * transform.pos.x;
      p.y -= s * transform.pos.y;
      p.z -= s * transform.pos.z;
      p.i = int(p.i);
      Angle rx = -s * transform.rot_x.rad();
      Angle ry = -s * transform.rot_y.rad();
      Angle rz = -s * transform.rot_z.rad();
      rotateZXY(p, rz, rx, ry);
      rotateYXZ(p, transform.rot_y, transform.rot_x, transform.rot_z);
      p.x += transform.pos.x - imuShiftFromStart.x;
      p.y += transform.pos.y - imuShiftFromStart.y;
      p.z += transform.pos.z - imuShiftFromStart.z;
      rotateZXY(p, imuRollStart, imuPitchStart, imuYawStart);
      rotateYXZ(p, -imuYawEnd, -imuPitchEnd, -imuRollEnd);
    }
    return cloud.size();
  }

  static void plugin_imu_rotation(const Angle& bcx, const Angle& bcy, const Angle& bcz, const Angle& blx,
                                  const Angle& bly, const Angle& blz, const Angle& alx, const Angle& aly,
                                  const Angle& alz, Angle& acx, Angle& acy, Angle& acz) {
    float sbcx = bcx.sin(), cbcx = bcx.cos(), sbcy = bcy.sin(), cbcy = bcy.cos(), sbcz = bcz.sin(), cbcz = bcz.cos();
    float sblx = blx.sin(), cblx = blx.cos(), sbly = bly.sin(), cbly = bly.cos(), sblz = blz.sin(), cblz = blz.cos();
    float salx = alx.sin(), calx = alx.cos(), saly = aly.sin(), caly = aly.cos(), salz = alz.sin(), calz = alz.cos();

    float srx = -sbcx * (salx * sblx + calx * caly * cblx * cbly + calx * cblx * saly * sbly) -
                cbcx * cbcz * (calx * saly * (cbly * sblz - cblz * sblx * sbly) -
                               calx * caly * (sbly * sblz + cbly * cblz * sblx) + cblx * cblz * salx) -
                cbcx * sbcz * (calx * caly * (cblz * sbly - cbly * sblx * sblz) -
                               calx * saly * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sblz);
    acx = -std::asin(srx);

    float srycrx = (cbcy * sbcz - cbcz * sbcx * sbcy) * (calx * saly * (cbly * sblz - cblz * sblx * sbly) -
                                                         calx * caly * (sbly * sblz + cbly * cblz * sblx) + cblx * cblz * salx) -
                   (cbcy * cbcz + sbcx * sbcy * sbcz) * (calx * caly * (cblz * sbly - cbly * sblx * sblz) -
                                                         calx * saly * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sblz) +
                   cbcx * sbcy * (salx * sblx + calx * caly * cblx * cbly + calx * cblx * saly * sbly);
    float crycrx = (cbcz * sbcy - cbcy * sbcx * sbcz) * (calx * caly * (cblz * sbly - cbly * sblx * sblz) -
                                                         calx * saly * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sblz) -
                   (sbcy * sbcz + cbcy * cbcz * sbcx) * (calx * saly * (cbly * sblz - cblz * sblx * sbly) -
                                                         calx * caly * (sbly * sblz + cbly * cblz * sblx) + cblx * cblz * salx) +
                   cbcx * cbcy * (salx * sblx + calx * caly * cblx * cbly + calx * cblx * saly * sbly);
    acy = std::atan2(srycrx / acx.cos(), crycrx / acx.cos());

    float srzcrx = sbcx * (cblx * cbly * (calz * saly - caly * salx * salz) - cblx * sbly * (caly * calz + salx * saly * salz) +
                           calx * salz * sblx) -
                   cbcx * cbcz * ((caly * calz + salx * saly * salz) * (cbly * sblz - cblz * sblx * sbly) +
                                  (calz * saly - caly * salx * salz) * (sbly * sblz + cbly * cblz * sblx) - calx * cblx * cblz * salz) +
                   cbcx * sbcz * ((caly * calz + salx * saly * salz) * (cbly * cblz + sblx * sbly * sblz) +
                                  (calz * saly - caly * salx * salz) * (cblz * sbly - cbly * sblx * sblz) + calx * cblx * salz * sblz);
    float crzcrx = sbcx * (cblx * sbly * (caly * salz - calz * salx * saly) - cblx * cbly * (saly * salz + caly * calz * salx) +
                           calx * calz * sblx) +
                   cbcx * cbcz * ((saly * salz + caly * calz * salx) * (sbly * sblz + cbly * cblz * sblx) +
                                  (caly * salz - calz * salx * saly) * (cbly * sblz - cblz * sblx * sbly) + calx * calz * cblx * cblz) -
                   cbcx * sbcz * ((saly * salz + caly * calz * salx) * (cblz * sbly - cbly * sblx * sblz) +
                                  (caly * salz - calz * salx * saly) * (cbly * cblz + sblx * sbly * sblz) - calx * calz * cblx * sblz);
    acz = std::atan2(srzcrx / acx.cos(), crzcrx / acx.cos());
  }

  static void accumulate_rotation(Angle cx, Angle cy, Angle cz, Angle lx, Angle ly, Angle lz, Angle& ox, Angle& oy,
                                  Angle& oz) {
    float srx = lx.cos() * cx.cos() * ly.sin() * cz.sin() - cx.cos() * cz.cos() * lx.sin() - lx.cos() * ly.cos() * cx.sin();
    ox = -std::asin(srx);
    float srycrx = lx.sin() * (cy.cos() * cz.sin() - cz.cos() * cx.sin() * cy.sin()) +
                   lx.cos() * ly.sin() * (cy.cos() * cz.cos() + cx.sin() * cy.sin() * cz.sin()) +
                   lx.cos() * ly.cos() * cx.cos() * cy.sin();
    float crycrx = lx.cos() * ly.cos() * cx.cos() * cy.cos() -
                   lx.cos() * ly.sin() * (cz.cos() * cy.sin() - cy.cos() * cx.sin() * cz.sin()) -
                   lx.sin() * (cy.sin() * cz.sin() + cy.cos() * cz.cos() * cx.sin());
    oy = std::atan2(srycrx / ox.cos(), crycrx / ox.cos());
    float srzcrx = cx.sin() * (lz.cos() * ly.sin() - ly.cos() * lx.sin() * lz.sin()) +
                   cx.cos() * cz.sin() * (ly.cos() * lz.cos() + lx.sin() * ly.sin() * lz.sin()) +
                   lx.cos() * cx.cos() * cz.cos() * lz.sin();
    float crzcrx = lx.cos() * lz.cos() * cx.cos() * cz.cos() -
                   cx.cos() * cz.sin() * (ly.cos() * lz.sin() - lz.cos() * lx.sin() * ly.sin()) -
                   cx.sin() * (ly.sin() * lz.sin() + ly.cos() * lz.cos() * lx.sin());
    oz = std::atan2(srzcrx / ox.cos(), crzcrx / ox.cos());
  }

  void process() {
    if (!systemInited) {
      cornerLessSharp.swap(lastCorner);
      surfLessFlat.swap(lastSurf);
      kdCorner_.build(&lastCorner);
      kdSurf_.build(&lastSurf);
      transformSum.rot_x += imuPitchStart.rad();
      transformSum.rot_z += imuRollStart.rad();
      systemInited = true;
      return;
    }
    Pt coeff;
    bool isDegenerate = false;
    float matP[36];
    frameCount++;
    transform.pos.x -= imuVeloFromStart.x * scanPeriod;
    transform.pos.y -= imuVeloFromStart.y * scanPeriod;
    transform.pos.z -= imuVeloFromStart.z * scanPeriod;
    lastIterCount = 0;
    lastSelNum = 0;

    size_t lastCornerN = lastCorner.size(), lastSurfN = lastSurf.size();
    if (lastCornerN > 10 && lastSurfN > 100) {
      int sInd[1];
      float sDis[1];
      const size_t nSharp = cornerSharp.size(), nFlat = surfFlat.size();
      ind1c_.resize(nSharp); ind2c_.resize(nSharp);
      ind1s_.resize(nFlat); ind2s_.resize(nFlat); ind3s_.resize(nFlat);
      std::vector<Pt> ori, coeffs;
      std::vector<float> A, B;

      for (size_t iter = 0; iter < maxIterations; iter++) {
        lastIterCount = (int)iter + 1;
        Pt pointSel, tripod1, tripod2, tripod3;
        ori.clear();
        coeffs.clear();
        for (size_t i = 0; i < nSharp; i++) {
          transform_to_start(cornerSharp[i], pointSel);
          if (iter % 5 == 0) {
            kdCorner_.knn(pointSel, 1, sInd, sDis);
            int closest = -1, minInd2 = -1;
            if (sDis[0] < 25) {
              closest = sInd[0];
              int closestScan = int(lastCorner[closest].i);
              float d, minD2 = 25;
              for (int j = closest + 1; j < (int)std::min(nSharp, lastCornerN); j++) {   // bound = CURRENT sharp count (:262); min() only guards the reference's out-of-range read
                if (int(lastCorner[j].i) > closestScan + 2.5) break;
                d = sq_diff(lastCorner[j], pointSel);
                if (int(lastCorner[j].i) > closestScan) {
                  if (d < minD2) { minD2 = d; minInd2 = j; }
                }
              }
              for (int j = closest - 1; j >= 0; j--) {
                if (int(lastCorner[j].i) < closestScan - 2.5) break;
                d = sq_diff(lastCorner[j], pointSel);
                if (int(lastCorner[j].i) < closestScan) {
                  if (d < minD2) { minD2 = d; minInd2 = j; }
                }
              }
            }
            ind1c_[i] = closest;
            ind2c_[i] = minInd2;
          }
          if (ind2c_[i] >= 0) {
            tripod1 = lastCorner[ind1c_[i]];
            tripod2 = lastCorner[ind2c_[i]];
            float x0 = pointSel.x, y0 = pointSel.y, z0 = pointSel.z;
            float x1 = tripod1.x, y1 = tripod1.y, z1 = tripod1.z;
            float x2 = tripod2.x, y2 = tripod2.y, z2 = tripod2.z;
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
            float s = 1;
            if (iter >= 5) s = 1 - 1.8f * std::fabs(ld2);
            coeff.x = s * la; coeff.y = s * lb; coeff.z = s * lc; coeff.i = s * ld2;
            if (s > 0.1 && ld2 != 0) {
              ori.push_back(cornerSharp[i]);
              coeffs.push_back(coeff);
            }
          }
        }
        for (size_t i = 0; i < nFlat; i++) {
          transform_to_start(surfFlat[i], pointSel);
          if (iter % 5 == 0) {
            kdSurf_.knn(pointSel, 1, sInd, sDis);
            int closest = -1, minInd2 = -1, minInd3 = -1;
            if (sDis[0] < 25) {
              closest = sInd[0];
              int closestScan = int(lastSurf[closest].i);
              float d, minD2 = 25, minD3 = 25;
              for (int j = closest + 1; j < (int)std::min(nFlat, lastSurfN); j++) {   // bound = CURRENT flat count (:378)
                if (int(lastSurf[j].i) > closestScan + 2.5) break;
                d = sq_diff(lastSurf[j], pointSel);
                if (int(lastSurf[j].i) <= closestScan) {
                  if (d < minD2) { minD2 = d; minInd2 = j; }
                } else {
                  if (d < minD3) { minD3 = d; minInd3 = j; }
                }
              }
              for (int j = closest - 1; j >= 0; j--) {
                if (int(lastSurf[j].i) < closestScan - 2.5) break;
                d = sq_diff(lastSurf[j], pointSel);
                if (int(lastSurf[j].i) >= closestScan) {
                  if (d < minD2) { minD2 = d; minInd2 = j; }
                } else {
                  if (d < minD3) { minD3 = d; minInd3 = j; }
                }
              }
            }
            ind1s_[i] = closest; ind2s_[i] = minInd2; ind3s_[i] = minInd3;
          }
          if (ind2s_[i] >= 0 && ind3s_[i] >= 0) {
            tripod1 = lastSurf[ind1s_[i]];
            tripod2 = lastSurf[ind2s_[i]];
            tripod3 = lastSurf[ind3s_[i]];
            float pa = (tripod2.y - tripod1.y) * (tripod3.z - tripod1.z) - (tripod3.y - tripod1.y) * (tripod2.z - tripod1.z);
            float pb = (tripod2.z - tripod1.z) * (tripod3.x - tripod1.x) - (tripod3.z - tripod1.z) * (tripod2.x - tripod1.x);
            float pc = (tripod2.x - tripod1.x) * (tripod3.y - tripod1.y) - (tripod3.x - tripod1.x) * (tripod2.y - tripod1.y);
            float pd = -(pa * tripod1.x + pb * tripod1.y + pc * tripod1.z);
            float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
            pa /= ps; pb /= ps; pc /= ps; pd /= ps;
            float pd2 = pa * pointSel.x + pb * pointSel.y + pc * pointSel.z + pd;
            float s = 1;
            if (iter >= 5) s = 1 - 1.8f * std::fabs(pd2) / std::sqrt(pt_dist(pointSel));
            coeff.x = s * pa; coeff.y = s * pb; coeff.z = s * pc; coeff.i = s * pd2;
            if (s > 0.1 && pd2 != 0) {
              ori.push_back(surfFlat[i]);
              coeffs.push_back(coeff);
            }
          }
        }
        const int selNum = (int)ori.size();
        lastSelNum = selNum;
        if (selNum < 10) continue;

        A.resize((size_t)selNum * 6);
        B.resize(selNum);
        for (int i = 0; i < selNum; i++) {
          const Pt& po = ori[i];
          coeff = coeffs[i];
          float s = 1;
          float srx = std::sin(s * transform.rot_x.rad()), crx = std::cos(s * transform.rot_x.rad());
          float sry = std::sin(s * transform.rot_y.rad()), cry = std::cos(s * transform.rot_y.rad());
          float srz = std::sin(s * transform.rot_z.rad()), crz = std::cos(s * transform.rot_z.rad());
          float tx = s * transform.pos.x, ty = s * transform.pos.y, tz = s * transform.pos.z;

          float arx = (-s * crx * sry * srz * po.x + s * crx * crz * sry * po.y + s * srx * sry * po.z + s * tx * crx * sry * srz -
                       s * ty * crx * crz * sry - s * tz * srx * sry) * coeff.x +
                      (s * srx * srz * po.x - s * crz * srx * po.y + s * crx * po.z + s * ty * crz * srx - s * tz * crx -
                       s * tx * srx * srz) * coeff.y +
                      (s * crx * cry * srz * po.x - s * crx * cry * crz * po.y - s * cry * srx * po.z + s * tz * cry * srx +
                       s * ty * crx * cry * crz - s * tx * crx * cry * srz) * coeff.z;
          float ary = ((-s * crz * sry - s * cry * srx * srz) * po.x + (s * cry * crz * srx - s * sry * srz) * po.y -
                       s * crx * cry * po.z + tx * (s * crz * sry + s * cry * srx * srz) + ty * (s * sry * srz - s * cry * crz * srx) +
                       s * tz * crx * cry) * coeff.x +
                      ((s * cry * crz - s * srx * sry * srz) * po.x + (s * cry * srz + s * crz * srx * sry) * po.y -
                       s * crx * sry * po.z + s * tz * crx * sry - ty * (s * cry * srz + s * crz * srx * sry) -
                       tx * (s * cry * crz - s * srx * sry * srz)) * coeff.z;
          float arz = ((-s * cry * srz - s * crz * srx * sry) * po.x + (s * cry * crz - s * srx * sry * srz) * po.y +
                       tx * (s * cry * srz + s * crz * srx * sry) - ty * (s * cry * crz - s * srx * sry * srz)) * coeff.x +
                      (-s * crx * crz * po.x - s * crx * srz * po.y + s * ty * crx * srz + s * tx * crx * crz) * coeff.y +
                      ((s * cry * crz * srx - s * sry * srz) * po.x + (s * crz * sry + s * cry * srx * srz) * po.y +
                       tx * (s * sry * srz - s * cry * crz * srx) - ty * (s * crz * sry + s * cry * srx * srz)) * coeff.z;
          float atx = -s * (cry * crz - srx * sry * srz) * coeff.x + s * crx * srz * coeff.y - s * (crz * sry + cry * srx * srz) * coeff.z;
          float aty = -s * (cry * srz + crz * srx * sry) * coeff.x - s * crx * crz * coeff.y - s * (sry * srz - cry * crz * srx) * coeff.z;
          float atz = s * crx * sry * coeff.x - s * srx * coeff.y - s * crx * cry * coeff.z;
          float d2 = coeff.i;
          float* a = &A[(size_t)i * 6];
          a[0] = arx; a[1] = ary; a[2] = arz; a[3] = atx; a[4] = aty; a[5] = atz;
          B[i] = (float)(-0.05 * d2);
        }
        NormalEq ne;
        accumulate_normal_eq(A, B, ne);
        float X[6];
        colpiv_qr_solve<6, 6>(ne.AtA, ne.AtB, X);
        if (iter == 0) isDegenerate = degeneracy_projector(ne.AtA, 10.f, matP);
        if (isDegenerate) {
          float X2[6];
          std::memcpy(X2, X, sizeof(X));
          for (int r = 0; r < 6; r++) {
            float s = 0.f;
            for (int c = 0; c < 6; c++) s += matP[r * 6 + c] * X2[c];
            X[r] = s;
          }
        }
        transform.rot_x = transform.rot_x.rad() + X[0];
        transform.rot_y = transform.rot_y.rad() + X[1];
        transform.rot_z = transform.rot_z.rad() + X[2];
        transform.pos.x += X[3];
        transform.pos.y += X[4];
        transform.pos.z += X[5];
        if (!std::isfinite(transform.rot_x.rad())) transform.rot_x = Angle();
        if (!std::isfinite(transform.rot_y.rad())) transform.rot_y = Angle();
        if (!std::isfinite(transform.rot_z.rad())) transform.rot_z = Angle();
        if (!std::isfinite(transform.pos.x)) transform.pos.x = 0.f;
        if (!std::isfinite(transform.pos.y)) transform.pos.y = 0.f;
        if (!std::isfinite(transform.pos.z)) transform.pos.z = 0.f;

        float deltaR = std::sqrt(std::pow(rad2deg_f(X[0]), 2) + std::pow(rad2deg_f(X[1]), 2) + std::pow(rad2deg_f(X[2]), 2));
        float deltaT = std::sqrt(std::pow(X[3] * 100, 2) + std::pow(X[4] * 100, 2) + std::pow(X[5] * 100, 2));
        if (deltaR < deltaRAbort && deltaT < deltaTAbort) break;
      }
    }

    Angle rx, ry, rz;
    accumulate_rotation(transformSum.rot_x, transformSum.rot_y, transformSum.rot_z, -transform.rot_x,
                        Angle((float)(-transform.rot_y.rad() * 1.05)), -transform.rot_z, rx, ry, rz);
    Vec3 v{transform.pos.x - imuShiftFromStart.x, transform.pos.y - imuShiftFromStart.y,
           (float)(transform.pos.z * 1.05 - imuShiftFromStart.z)};
    rotateZXY(v, rz, rx, ry);
    Vec3 trans = transformSum.pos - v;
    plugin_imu_rotation(rx, ry, rz, imuPitchStart, imuYawStart, imuRollStart, imuPitchEnd, imuYawEnd, imuRollEnd, rx, ry, rz);
    transformSum.rot_x = rx;
    transformSum.rot_y = ry;
    transformSum.rot_z = rz;
    transformSum.pos = trans;

    transform_to_end(cornerLessSharp);
    transform_to_end(surfLessFlat);
    cornerLessSharp.swap(lastCorner);
    surfLessFlat.swap(lastSurf);
    lastCornerN = lastCorner.size();
    lastSurfN = lastSurf.size();
    if (lastCornerN > 10 && lastSurfN > 100) {
      kdCorner_.build(&lastCorner);
      kdSurf_.build(&lastSurf);
    }
  }

 private:
  KdTree kdCorner_, kdSurf_;
  std::vector<int> ind1c_, ind2c_, ind1s_, ind2s_, ind3s_;
};

}  // namespace loam_oracle
