// sos_devmath.h -- small fp64 device functions shared by the device-resident loops (sos_gn_resident.inc, sos_tracker_lm.inc)
#pragma once
#include <hip/hip_runtime.h>

// thirdparty/Sophus/sophus/se3.hpp:407-428 + so3.hpp:343-368 as restated in csrc/host/sos_math.hpp (SE3::exp)
__device__ static inline void sos_dev_se3_exp(const double *a, double *R, double *t) {
  const double eps = 1e-10;
  const double *om = a + 3;
  const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  const double theta = sqrt(theta_sq);
  double imag, real;
  if (theta < eps) {
    const double po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * po4;
    real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * po4;
  } else {
    double sh, ch;
    sincos(0.5 * theta, &sh, &ch);  // one argument reduction for the pair
    imag = sh / theta;
    real = ch;
  }
  double qw = real, qx = imag * om[0], qy = imag * om[1], qz = imag * om[2];
  const double nrm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy,
               tzz = tz * qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
  const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double V[9];
  if (theta < eps) {
    for (int i = 0; i < 9; i++) V[i] = R[i];
  } else {
    double Om2[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Om2[3 * i + j] = Om[3 * i] * Om[j] + Om[3 * i + 1] * Om[3 + j] + Om[3 * i + 2] * Om[6 + j];
    double st_, ct_;
    sincos(theta, &st_, &ct_);
    const double c1 = (1.0 - ct_) / theta_sq, c2 = (theta - st_) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
  }
  for (int i = 0; i < 3; i++) t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
}

