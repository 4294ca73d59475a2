/*
 * oracle/orc_math.h -- TEST INFRASTRUCTURE (CPU oracle).  Not part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load anything under oracle/.
 *
 * Plain-C fp64 math floor of the oracle: SE3 (restating thirdparty/Sophus/sophus/se3.hpp:131-139
 * Adj, :168-173 inverse, :407-428 exp, :560-586 log; so3.hpp:343-368 expAndTheta, :491-526
 * logAndTheta), a pivoted LDL^T solve standing in for Eigen's `.ldlt().solve`
 * (OB/EnergyFunctional.cpp:1148, FS/CoarseTracker.cpp:423) and a dense inverse
 * (OB/EnergyFunctional.cpp:841).  Eigen is not vendored by the reference (CMakeLists.txt:8), so the
 * solve is pinned by residual norm, not bitwise.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SOPHUS_EPS 1e-10 /* SophusConstants<double>::epsilon() */

typedef struct orc_se3 {
  double R[9]; /* row-major rotation */
  double t[3];
} orc_se3;

static inline void orc_mat3_mul(const double *A, const double *B, double *C) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
static inline void orc_mat3_vec(const double *A, const double *v, double *o) {
  double T[3];
  for (int i = 0; i < 3; i++) T[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
  o[0] = T[0]; o[1] = T[1]; o[2] = T[2];
}
static inline void orc_hat(const double *w, double *O) {
  O[0] = 0; O[1] = -w[2]; O[2] = w[1];
  O[3] = w[2]; O[4] = 0; O[5] = -w[0];
  O[6] = -w[1]; O[7] = w[0]; O[8] = 0;
}
static inline void orc_quat_to_R(double qw, double qx, double qy, double qz, double *R) {
  /* Eigen::Quaternion::toRotationMatrix */
  double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
/* so3.hpp:343-368 */
static inline void orc_so3_exp(const double *omega, double *R, double *theta_out) {
  double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  double theta = sqrt(theta_sq);
  double half = 0.5 * theta, imag, real;
  if (theta < ORC_SOPHUS_EPS) {
    double po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * po4;
    real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * po4;
  } else {
    imag = sin(half) / theta;
    real = cos(half);
  }
  double qw = real, qx = imag * omega[0], qy = imag * omega[1], qz = imag * omega[2];
  /* SO3Group(Quaternion) normalises */
  double nrm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw /= nrm; qx /= nrm; qy /= nrm; qz /= nrm;
  orc_quat_to_R(qw, qx, qy, qz, R);
  if (theta_out) *theta_out = theta;
}
/* se3.hpp:407-428; tangent = [upsilon(3) omega(3)] */
static inline orc_se3 orc_se3_exp(const double *a) {
  orc_se3 T;
  const double *omega = a + 3;
  double theta;
  orc_so3_exp(omega, T.R, &theta);
  double Om[9], Om2[9], V[9];
  orc_hat(omega, Om);
  orc_mat3_mul(Om, Om, Om2);
  if (theta < ORC_SOPHUS_EPS) {
    memcpy(V, T.R, sizeof(V));
  } else {
    double tsq = theta * theta;
    double c1 = (1.0 - cos(theta)) / tsq, c2 = (theta - sin(theta)) / (tsq * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
  }
  orc_mat3_vec(V, a, T.t);
  return T;
}
static inline orc_se3 orc_se3_identity(void) {
  orc_se3 T;
  memset(&T, 0, sizeof(T));
  T.R[0] = T.R[4] = T.R[8] = 1;
  return T;
}
static inline orc_se3 orc_se3_mul(const orc_se3 *A, const orc_se3 *B) {
  orc_se3 C;
  double Rt[3];
  orc_mat3_vec(A->R, B->t, Rt);
  orc_mat3_mul(A->R, B->R, C.R);
  for (int i = 0; i < 3; i++) C.t[i] = A->t[i] + Rt[i];
  return C;
}
static inline orc_se3 orc_se3_inverse(const orc_se3 *A) {
  orc_se3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.R[3 * i + j] = A->R[3 * j + i];
  double nt[3] = {-A->t[0], -A->t[1], -A->t[2]};
  orc_mat3_vec(C.R, nt, C.t);
  return C;
}
/* se3.hpp:131-139: [R, hat(t)R; 0, R], 6x6 row-major */
static inline void orc_se3_adj(const orc_se3 *A, double *Ad) {
  double H[9], HR[9];
  orc_hat(A->t, H);
  orc_mat3_mul(H, A->R, HR);
  memset(Ad, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ad[6 * i + j] = A->R[3 * i + j];
      Ad[6 * (i + 3) + (j + 3)] = A->R[3 * i + j];
      Ad[6 * i + (j + 3)] = HR[3 * i + j];
    }
}
/* rotation matrix -> unit quaternion (Eigen::Quaternion(Matrix3)) */
static inline void orc_R_to_quat(const double *R, double *q /* w x y z */) {
  double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    q[0] = 0.5 * s;
    s = 0.5 / s;
    q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[1 + i] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * s;
    q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * s;
    q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * s;
  }
}
/* so3.hpp:491-526 + se3.hpp:560-586 */
static inline void orc_se3_log(const orc_se3 *A, double *out) {
  double q[4];
  orc_R_to_quat(A->R, q);
  double sq = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double n = sqrt(sq), w = q[0], f;
  if (n < ORC_SOPHUS_EPS) {
    f = 2.0 / w - 2.0 * sq / (w * w * w);
  } else if (fabs(w) < ORC_SOPHUS_EPS) {
    f = (w > 0 ? M_PI : -M_PI) / n;
  } else {
    f = 2.0 * atan(n / w) / n;
  }
  double theta = f * n;
  double om[3] = {f * q[1], f * q[2], f * q[3]};
  double Om[9], Om2[9], Vi[9];
  orc_hat(om, Om);
  orc_mat3_mul(Om, Om, Om2);
  double c;
  if (fabs(theta) < ORC_SOPHUS_EPS) c = 1.0 / 12.0;
  else c = (1.0 - theta / (2.0 * tan(theta / 2.0))) / (theta * theta);
  for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
  orc_mat3_vec(Vi, A->t, out);
  out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}

/* Pivoted LDL^T solve of a symmetric (possibly indefinite / semi-definite) system, n x n row-major.
 * Stand-in for Eigen::LDLT::solve: symmetric pivoting on the largest |diagonal|; a pivot that is
 * exactly zero (|d| <= DBL_MIN, Eigen's solve tolerance) contributes nothing. Returns 0. */
static inline int orc_ldlt_solve(const double *A, const double *b, double *x, int n) {
  size_t N = (size_t)n;
  double *M = (double *)malloc(sizeof(double) * N * N);
  double *L = (double *)calloc(N * N, sizeof(double));
  double *D = (double *)malloc(sizeof(double) * N);
  double *y = (double *)malloc(sizeof(double) * N);
  int *perm = (int *)malloc(sizeof(int) * N);
  if (!M || !L || !D || !y || !perm) return -1;
  /* Eigen's LDLT<MatrixXd, Lower> (what `.ldlt()` is, OB/EnergyFunctional.cpp:1148) references ONLY the lower triangle of its input.
   * The matrices it is handed are not exactly symmetric -- AccumulatedSCHessian.cpp:128-139 fills block (j, k) and block (k, j) from
   * float accumulators whose entries were rounded as (w L_p) R_q and (w R_q) L_p -- so the elimination below, which carries both
   * triangles along, starts from the lower triangle mirrored: on a symmetric matrix it is Eigen's arithmetic, pivot for pivot. */
  {
    static int both = -1; /* ORC_LDLT_BOTH_TRIANGLES=1: the elimination of rounds 1-2 (both triangles carried along), for A/B runs */
    if (both < 0) both = getenv("ORC_LDLT_BOTH_TRIANGLES") != NULL;
    if (both) memcpy(M, A, sizeof(double) * N * N);
    else
      for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) M[(size_t)i * N + j] = M[(size_t)j * N + i] = A[(size_t)i * N + j];
  }
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = fabs(M[k * N + k]);
    for (int i = k + 1; i < n; i++)
      if (fabs(M[i * N + i]) > best) { best = fabs(M[i * N + i]); p = i; }
    if (p != k) {
      for (int j = 0; j < n; j++) { double t = M[k * N + j]; M[k * N + j] = M[p * N + j]; M[p * N + j] = t; }
      for (int j = 0; j < n; j++) { double t = M[j * N + k]; M[j * N + k] = M[j * N + p]; M[j * N + p] = t; }
      for (int j = 0; j < k; j++) { double t = L[k * N + j]; L[k * N + j] = L[p * N + j]; L[p * N + j] = t; }
      int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
    }
    double d = M[k * N + k];
    D[k] = d;
    L[k * N + k] = 1.0;
    if (!(fabs(d) > 2.2250738585072014e-308)) continue;
    for (int i = k + 1; i < n; i++) L[i * N + k] = M[i * N + k] / d;
    for (int i = k + 1; i < n; i++) {
      double lik = L[i * N + k];
      for (int j = k + 1; j < n; j++) M[i * N + j] -= lik * M[k * N + j];
    }
  }
  for (int i = 0; i < n; i++) y[i] = b[perm[i]];
  for (int i = 0; i < n; i++) {
    double s = y[i];
    for (int j = 0; j < i; j++) s -= L[i * N + j] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < n; i++) y[i] = (fabs(D[i]) > 2.2250738585072014e-308) ? y[i] / D[i] : 0.0;
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int j = i + 1; j < n; j++) s -= L[j * N + i] * y[j];
    y[i] = s;
  }
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
  free(M); free(L); free(D); free(y); free(perm);
  return 0;
}

/* dense inverse by Gauss-Jordan with partial pivoting (stand-in for Eigen `.inverse()`,
 * OB/EnergyFunctional.cpp:841). Returns 0 on success. */
static inline int orc_mat_inverse(const double *A, double *Ainv, int n) {
  size_t N = (size_t)n;
  double *M = (double *)malloc(sizeof(double) * N * 2 * N);
  if (!M) return -1;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      M[i * 2 * N + j] = A[i * N + j];
      M[i * 2 * N + N + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int i = k + 1; i < n; i++)
      if (fabs(M[i * 2 * N + k]) > fabs(M[p * 2 * N + k])) p = i;
    if (p != k)
      for (int j = 0; j < 2 * n; j++) { double t = M[k * 2 * N + j]; M[k * 2 * N + j] = M[p * 2 * N + j]; M[p * 2 * N + j] = t; }
    double d = M[k * 2 * N + k];
    for (int j = 0; j < 2 * n; j++) M[k * 2 * N + j] /= d;
    for (int i = 0; i < n; i++)
      if (i != k) {
        double f = M[i * 2 * N + k];
        if (f != 0.0)
          for (int j = 0; j < 2 * n; j++) M[i * 2 * N + j] -= f * M[k * 2 * N + j];
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ainv[i * N + j] = M[i * 2 * N + N + j];
  free(M);
  return 0;
}

#endif /* ORC_MATH_H */
