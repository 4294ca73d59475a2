/* oracle/orc_imu.c -- TEST INFRASTRUCTURE (CPU oracle, see oracle.h; PARITY UNPINNED).
 *
 * Restatement of the IMU / spline factor assembly that sits inside EnergyFunctional::solveSystemF:
 *   FrameHessian::getImuHi, spline accessors          FS/HessianBlocks.cpp:178-225, FS/HessianBlocks.h:352-412
 *   getImuHessianCurrentFrame / getImuHessian         OB/EnergyFunctional.cpp:288-494
 *   expandHbtoFitImu                                  OB/EnergyFunctional.cpp:256-286
 *   the IMU branch of solveSystemF                    OB/EnergyFunctional.cpp:1053-1171
 * All fp64, dense row-major arrays in place of Eigen blocks, written in the order of the reference's statements. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sos_slam_host.h"
#include "oracle.h"
#include "orc_math.h"

#define CP 4
#define SC_SCALE 200.0
#define SC_SL_ROT 100.0
#define SC_SQ_TRANS 1000.0
#define SC_SQ_ROT 1000.0
#define SC_SC_TRANS 1000.0
#define SC_SC_ROT 1000.0
#define SC_BA 100.0
#define SC_BG 1.0
#define SC_XI_ROT 1.0
#define SC_XI_TRANS 0.5

/* setImuState: state_imu -> state_imu_scaled (FS/HessianBlocks.h:352-361) */
static void scaled_state(const double *s, double *o) {
  for (int i = 0; i < 3; i++) {
    o[i] = SC_BA * s[i];
    o[3 + i] = SC_BG * s[3 + i];
    o[6 + i] = SC_SL_ROT * s[6 + i];
    o[9 + i] = SC_SQ_TRANS * s[9 + i];
    o[12 + i] = SC_SQ_ROT * s[12 + i];
    o[15 + i] = SC_SC_TRANS * s[15 + i];
    o[18 + i] = SC_SC_ROT * s[18 + i];
  }
}
/* imu_bias = scaled[0:6], spline_l_rot = scaled[6:9], spline_q = scaled[9:15], spline_c = scaled[15:21] (:277-280) */
static void spline_acc(const sosf_imu_frame *f, double t, int zero, double *acc) { /* getSplineAcc, :379-388 */
  double sc[21];
  scaled_state(f->state_imu, sc);
  for (int i = 0; i < 3; i++)
    acc[i] = zero ? 2 * SC_SQ_TRANS * f->state_imu_zero[9 + i] + 6 * t * SC_SC_TRANS * f->state_imu_zero[15 + i]
                  : 2 * sc[9 + i] + 6 * t * sc[15 + i];
}
static void spline_gyro(const sosf_imu_frame *f, double t, double *g) { /* getSplineGryo, :390-392 */
  double sc[21];
  scaled_state(f->state_imu, sc);
  for (int i = 0; i < 3; i++) g[i] = sc[6 + i] + (2 * t * sc[12 + i] + 3 * t * t * sc[18 + i]);
}
static void spline_R_c_t(const sosf_imu_frame *f, double t, int zero, double *R) { /* getSplineR_c_t, :399-410 */
  double t2 = t * t, so3[3], sc[21];
  scaled_state(f->state_imu, sc);
  for (int i = 0; i < 3; i++)
    so3[i] = zero ? t * SC_SL_ROT * f->state_imu_zero[6 + i] + t2 * SC_SQ_ROT * f->state_imu_zero[12 + i] +
                        t * t2 * SC_SC_ROT * f->state_imu_zero[18 + i]
                  : t * sc[6 + i] + (t2 * sc[12 + i] + t * t2 * sc[18 + i]);
  orc_so3_exp(so3, R, 0);
}
static void mat3_T(const double *A, double *T) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * j + i];
}
static void so3_log(const double *R, double *w) {
  orc_se3 T;
  memcpy(T.R, R, sizeof(T.R));
  T.t[0] = T.t[1] = T.t[2] = 0;
  double l[6];
  orc_se3_log(&T, l);
  w[0] = l[3]; w[1] = l[4]; w[2] = l[5];
}

void orc_imu_get_Hi(const sosf_imu_settings *S, const sosf_imu_calib *C, const sosf_imu_frame *f, double tt, double *JsTW,
                    double *JfTW, double *Hss, double *Hff, double *Hfs) { /* FS/HessianBlocks.cpp:178-225 */
  double tt2 = tt * tt;
  int trapped = C->scale_trapped;
  double scale_scaled = trapped ? C->scale_zero * SC_SCALE : C->scale * SC_SCALE;
  double acc[3], acc_w[3];
  spline_acc(f, tt, trapped, acc);
  for (int i = 0; i < 3; i++) acc_w[i] = scale_scaled * acc[i] + S->gravity[i];
  double Rct[9], RctT[9], RevT[9], rot_t_w[9], rot_i_w[9];
  spline_R_c_t(f, tt, trapped, Rct);
  mat3_T(Rct, RctT);
  mat3_T(f->evalPT_R, RevT);
  orc_mat3_mul(RctT, RevT, rot_t_w);
  orc_mat3_mul(S->rot_imu_cam, rot_t_w, rot_i_w);
  double v[3], hatv[9], R_acc_t_hat[9], hat_accw[9], riw_hat[9];
  orc_mat3_vec(rot_t_w, acc_w, v);
  orc_hat(v, hatv);
  orc_mat3_mul(S->rot_imu_cam, hatv, R_acc_t_hat);
  double Js[6] = {0, 0, 0, 0, 0, 0}, Jf[6 * 29];
  memset(Jf, 0, sizeof(Jf));
  double ra[3];
  orc_mat3_vec(rot_i_w, acc, ra);
  for (int i = 0; i < 3; i++) Js[i] = SC_SCALE * ra[i];
  if (trapped) {
    orc_hat(acc_w, hat_accw);
    orc_mat3_mul(rot_i_w, hat_accw, riw_hat);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Jf[29 * i + 3 + j] = SC_XI_ROT * riw_hat[3 * i + j];
  }
  for (int i = 0; i < 3; i++) {
    Jf[29 * i + 8 + i] = SC_BA;
    for (int j = 0; j < 3; j++) {
      Jf[29 * i + 14 + j] = SC_SL_ROT * R_acc_t_hat[3 * i + j] * tt;
      Jf[29 * i + 20 + j] = SC_SQ_ROT * R_acc_t_hat[3 * i + j] * tt2;
      Jf[29 * i + 26 + j] = SC_SC_ROT * R_acc_t_hat[3 * i + j] * tt * tt2;
      Jf[29 * i + 17 + j] = SC_SQ_TRANS * rot_i_w[3 * i + j] * 2 * scale_scaled;
      Jf[29 * i + 23 + j] = SC_SC_TRANS * rot_i_w[3 * i + j] * 6 * tt * scale_scaled;
    }
    Jf[29 * (3 + i) + 11 + i] = SC_BG;
    for (int j = 0; j < 3; j++) {
      Jf[29 * (3 + i) + 14 + j] = SC_SL_ROT * S->rot_imu_cam[3 * i + j];
      Jf[29 * (3 + i) + 20 + j] = SC_SQ_ROT * S->rot_imu_cam[3 * i + j] * 2 * tt;
      Jf[29 * (3 + i) + 26 + j] = SC_SC_ROT * S->rot_imu_cam[3 * i + j] * 3 * tt2;
    }
  }
  /* JsTW = Js^T W, JfTW = Jf^T W, Hss = JsTW Js, Hff = JfTW Jf, Hfs = JfTW Js */
  for (int c = 0; c < 6; c++) {
    double a = 0;
    for (int k = 0; k < 6; k++) a += Js[k] * S->weight_imu[6 * k + c];
    JsTW[c] = a;
  }
  for (int r = 0; r < 29; r++)
    for (int c = 0; c < 6; c++) {
      double a = 0;
      for (int k = 0; k < 6; k++) a += Jf[29 * k + r] * S->weight_imu[6 * k + c];
      JfTW[6 * r + c] = a;
    }
  double hs = 0;
  for (int k = 0; k < 6; k++) hs += JsTW[k] * Js[k];
  *Hss = hs;
  for (int r = 0; r < 29; r++) {
    for (int c = 0; c < 29; c++) {
      double a = 0;
      for (int k = 0; k < 6; k++) a += JfTW[6 * r + k] * Jf[29 * k + c];
      Hff[29 * r + c] = a;
    }
    double a = 0;
    for (int k = 0; k < 6; k++) a += JfTW[6 * r + k] * Js[k];
    Hfs[r] = a;
  }
}

/* getImuHessianCurrentFrame, OB/EnergyFunctional.cpp:288-441 (print branch omitted).  Returns the number of constraint
 * rows it appended (0 when the spline of this frame is not valid). */
static int imu_hessian_frame(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, int fi, double *H,
                             double *b, double *Jc, double *rc, int32_t *spline_valid) {
  const int dim = SOSF_IMU_DIM(n);
  const sosf_imu_frame *cur = &F[fi], *prv = &F[fi - 1];
  double tpf = prv->timestamp - cur->timestamp, tpf2 = tpf * tpf;
  int cur_idx = CP + 1 + 29 * fi, prv_idx = CP + 1 + 29 * (fi - 1);
  /* bias random walk, :304-318 */
  double tmpH[36];
  for (int i = 0; i < 36; i++) tmpH[i] = S->weight_imu_bias[i] / -tpf;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      tmpH[6 * i + j] *= (SC_BA * SC_BA);
      tmpH[6 * (3 + i) + 3 + j] *= (SC_BG * SC_BG);
    }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      H[(size_t)(prv_idx + 8 + i) * dim + prv_idx + 8 + j] += tmpH[6 * i + j];
      H[(size_t)(cur_idx + 8 + i) * dim + cur_idx + 8 + j] += tmpH[6 * i + j];
      H[(size_t)(prv_idx + 8 + i) * dim + cur_idx + 8 + j] += -tmpH[6 * i + j];
      H[(size_t)(cur_idx + 8 + i) * dim + prv_idx + 8 + j] += -tmpH[6 * i + j];
    }
  double sc_c[21], sc_p[21], r_bias[6], tmpb[6];
  scaled_state(cur->state_imu, sc_c);
  scaled_state(prv->state_imu, sc_p);
  for (int i = 0; i < 6; i++) r_bias[i] = sc_c[i] - sc_p[i];
  for (int i = 0; i < 6; i++) {
    double a = 0;
    for (int k = 0; k < 6; k++) a += (S->weight_imu_bias[6 * i + k] / -tpf) * r_bias[k];
    tmpb[i] = a;
  }
  for (int i = 0; i < 3; i++) { tmpb[i] *= SC_BA; tmpb[3 + i] *= SC_BG; }
  for (int i = 0; i < 6; i++) { b[prv_idx + 8 + i] += -tmpb[i]; b[cur_idx + 8 + i] += tmpb[i]; }

  int sv = cur->trackingRefIsPrev && (-tpf < S->maxImuInterval);
  spline_valid[fi] = sv;
  int vel_valid = fi < (n - 1);
  int rows = 0;
  if (!sv) return 0;
  rows = vel_valid ? 6 : 3;
  memset(Jc, 0, sizeof(double) * (size_t)rows * dim);
  memset(rc, 0, sizeof(double) * rows);
  /* rotation constraint, :335-347 */
  double Rpred[9], Rc[9], RcT[9], Rmeas[9], RmT[9], M[9], w[3], RpevT[9];
  spline_R_c_t(cur, tpf, 0, Rpred);
  memcpy(Rc, cur->camToWorld, sizeof(Rc));
  mat3_T(Rc, RcT);
  orc_mat3_mul(RcT, prv->camToWorld, Rmeas);
  mat3_T(Rmeas, RmT);
  orc_mat3_mul(RmT, Rpred, M);
  so3_log(M, w);
  for (int i = 0; i < 3; i++) rc[i] = w[i];
  mat3_T(prv->evalPT_R, RpevT);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      Jc[(size_t)i * dim + prv_idx + 3 + j] = -SC_XI_ROT * RpevT[3 * i + j];
      Jc[(size_t)i * dim + cur_idx + 3 + j] = SC_XI_ROT * RpevT[3 * i + j];
    }
    Jc[(size_t)i * dim + cur_idx + 14 + i] = SC_SL_ROT * tpf;
    Jc[(size_t)i * dim + cur_idx + 20 + i] = SC_SQ_ROT * tpf2;
    Jc[(size_t)i * dim + cur_idx + 26 + i] = SC_SC_ROT * tpf * tpf2;
  }
  if (vel_valid) { /* :350-376 */
    const sosf_imu_frame *nxt = &F[fi + 1];
    double tnf = cur->timestamp - nxt->timestamp;
    if (nxt->trackingRefIsPrev && (-tnf < S->maxImuInterval)) {
      int nxt_idx = CP + 1 + 29 * (fi + 1);
      double tnf2 = tnf * tnf, sc_n[21];
      scaled_state(nxt->state_imu, sc_n);
      for (int i = 0; i < 3; i++) {
        double d_vel_dso = (1 / tpf) * (prv->camToWorld[9 + i] - cur->camToWorld[9 + i]) -
                           (1 / tnf) * (cur->camToWorld[9 + i] - nxt->camToWorld[9 + i]);
        double d_vel_imu = tpf * sc_c[9 + i] + tpf2 * sc_c[15 + i] + tnf * sc_n[9 + i] + 2 * tnf2 * sc_n[15 + i];
        rc[3 + i] = d_vel_imu - d_vel_dso;
        Jc[(size_t)(3 + i) * dim + prv_idx + i] = -SC_XI_TRANS / tpf;
        Jc[(size_t)(3 + i) * dim + cur_idx + i] = SC_XI_TRANS * (1 / tpf + 1 / tnf);
        Jc[(size_t)(3 + i) * dim + nxt_idx + i] = -SC_XI_TRANS / tnf;
        Jc[(size_t)(3 + i) * dim + cur_idx + 17 + i] = SC_SQ_TRANS * tpf;
        Jc[(size_t)(3 + i) * dim + cur_idx + 23 + i] = SC_SC_TRANS * tpf2;
        Jc[(size_t)(3 + i) * dim + nxt_idx + 17 + i] = SC_SQ_TRANS * tnf;
        Jc[(size_t)(3 + i) * dim + nxt_idx + 23 + i] = SC_SC_TRANS * 2 * tnf2;
      }
    }
  }
  /* IMU dynamics, :379-441 */
  double JsTW[6], JfTW[29 * 6], Hss, Hff[29 * 29], Hfs[29];
  if (C->scale_trapped) { /* the FEJ sums of setImuStateZero (FS/HessianBlocks.cpp:227-251), added in one go */
    double sHss = 0, sHff[29 * 29], sHfs[29];
    memset(sHff, 0, sizeof(sHff));
    memset(sHfs, 0, sizeof(sHfs));
    for (int j = 0; j < cur->n_imu; j++) {
      orc_imu_get_Hi(S, C, cur, cur->imu[7 * j] - cur->timestamp, JsTW, JfTW, &Hss, Hff, Hfs);
      sHss += Hss;
      for (int k = 0; k < 29 * 29; k++) sHff[k] += Hff[k];
      for (int k = 0; k < 29; k++) sHfs[k] += Hfs[k];
    }
    H[(size_t)CP * dim + CP] += sHss;
    for (int r = 0; r < 29; r++) {
      H[(size_t)(cur_idx + r) * dim + CP] += sHfs[r];
      H[(size_t)CP * dim + cur_idx + r] += sHfs[r];
      for (int c = 0; c < 29; c++) H[(size_t)(cur_idx + r) * dim + cur_idx + c] += sHff[29 * r + c];
    }
  }
  double RwcR[9];
  mat3_T(cur->camToWorld, RwcR); /* PRE_worldToCam.rotationMatrix() */
  double scale_scaled = C->scale * SC_SCALE;
  for (int j = 0; j < cur->n_imu; j++) {
    double tt = cur->imu[7 * j] - cur->timestamp;
    double Rct[9], RctT[9], A1[9], A2[9], acc[3], aw[3], pred[6], g[3];
    spline_R_c_t(cur, tt, 0, Rct);
    mat3_T(Rct, RctT);
    orc_mat3_mul(S->rot_imu_cam, RctT, A1);
    orc_mat3_mul(A1, RwcR, A2);
    spline_acc(cur, tt, 0, acc);
    for (int i = 0; i < 3; i++) aw[i] = scale_scaled * acc[i] + S->gravity[i];
    orc_mat3_vec(A2, aw, pred);
    spline_gyro(cur, tt, g);
    orc_mat3_vec(S->rot_imu_cam, g, pred + 3);
    double r_imu[6];
    for (int i = 0; i < 6; i++) r_imu[i] = (pred[i] + sc_c[i]) - cur->imu[7 * j + 1 + i];
    orc_imu_get_Hi(S, C, cur, tt, JsTW, JfTW, &Hss, Hff, Hfs);
    if (!C->scale_trapped) { /* no FEJ while initialising: Hessian parts per sample */
      H[(size_t)CP * dim + CP] += Hss;
      for (int r = 0; r < 29; r++) {
        H[(size_t)(cur_idx + r) * dim + CP] += Hfs[r];
        H[(size_t)CP * dim + cur_idx + r] += Hfs[r];
        for (int c = 0; c < 29; c++) H[(size_t)(cur_idx + r) * dim + cur_idx + c] += Hff[29 * r + c];
      }
    }
    double a = 0;
    for (int k = 0; k < 6; k++) a += JsTW[k] * r_imu[k];
    b[CP] += a;
    for (int r = 0; r < 29; r++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += JfTW[6 * r + k] * r_imu[k];
      b[cur_idx + r] += s;
    }
  }
  return rows;
}

int orc_imu_hessian(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, double *H, double *b,
                    double *J_cst, double *r_cst, int32_t *n_cst, int32_t *spline_valid) { /* getImuHessian, :443-481 */
  const int dim = SOSF_IMU_DIM(n);
  memset(H, 0, sizeof(double) * (size_t)dim * dim);
  memset(b, 0, sizeof(double) * dim);
  *n_cst = 0;
  spline_valid[0] = 0;
  if (n == 1) return 0;
  int rows = 0;
  for (int i = 1; i < n; i++) rows += imu_hessian_frame(S, C, n, F, i, H, b, J_cst + (size_t)rows * dim, r_cst + rows, spline_valid);
  *n_cst = rows;
  return 0;
}

void orc_imu_expand(int n, const double *H, const double *b, double *He, double *be) { /* expandHbtoFitImu, :256-286 */
  const int d0 = CP + 8 * n, dim = SOSF_IMU_DIM(n);
  memset(He, 0, sizeof(double) * (size_t)dim * dim);
  memset(be, 0, sizeof(double) * dim);
  for (int i = 0; i < CP; i++) {
    for (int j = 0; j < CP; j++) He[(size_t)i * dim + j] = H[(size_t)i * d0 + j];
    be[i] = b[i];
  }
  for (int i = 0; i < n; i++) {
    int fi = CP + 8 * i, fie = CP + 1 + 29 * i;
    for (int r = 0; r < CP; r++)
      for (int c = 0; c < 8; c++) {
        He[(size_t)r * dim + fie + c] += H[(size_t)r * d0 + fi + c];
        He[(size_t)(fie + c) * dim + r] += H[(size_t)(fi + c) * d0 + r];
      }
    for (int j = i; j < n; j++) {
      int fj = CP + 8 * j, fje = CP + 1 + 29 * j;
      for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
          He[(size_t)(fie + r) * dim + fje + c] += H[(size_t)(fi + r) * d0 + fj + c];
          if (j > i) He[(size_t)(fje + r) * dim + fie + c] += H[(size_t)(fj + r) * d0 + fi + c];
        }
    }
    for (int r = 0; r < 8; r++) be[fie + r] += b[fi + r];
  }
}

/* the IMU branch of solveSystemF, OB/EnergyFunctional.cpp:1053-1171 */
int orc_imu_solve(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, const double *H_top,
                  const double *b_top, const double *H_sc, const double *b_sc, const double *HM, const double *bM, const double *delta,
                  double lambda, double *x_out, double *scale_step, double *step_imu) {
  int dim = SOSF_IMU_DIM(n);
  const int dimI = dim, maxc = 6 * n, cap = dimI + maxc;
  double *Himu = (double *)malloc(sizeof(double) * (size_t)dimI * dimI), *bimu = (double *)malloc(sizeof(double) * dimI);
  double *Jc = (double *)calloc((size_t)(maxc ? maxc : 1) * dimI, sizeof(double)), *rc = (double *)calloc(maxc ? maxc : 1, sizeof(double));
  int32_t ncst = 0, *sv = (int32_t *)calloc(n, sizeof(int32_t));
  orc_imu_hessian(S, C, n, F, Himu, bimu, Jc, rc, &ncst, sv);
  double *Hf = (double *)calloc((size_t)cap * cap, sizeof(double)), *bf = (double *)calloc(cap, sizeof(double));
  double *He = (double *)malloc(sizeof(double) * (size_t)dimI * dimI), *be = (double *)malloc(sizeof(double) * dimI);
  /* expanded dso system + imu, :1064-1066 (kept with leading dimension dimI for now) */
  orc_imu_expand(n, H_top, b_top, He, be);
  for (size_t k = 0; k < (size_t)dimI * dimI; k++) He[k] += Himu[k];
  for (int k = 0; k < dimI; k++) be[k] += bimu[k];
  /* marginalisation prior, :1070-1090 */
  double *d2 = (double *)calloc(dimI, sizeof(double));
  for (int i = 0; i < CP; i++) d2[i] = delta[i];
  if (C->scale_trapped) d2[CP] = C->scale - C->scale_zero;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 8; k++) d2[CP + 1 + 29 * i + k] = delta[CP + 8 * i + k];
    if (C->scale_trapped)
      for (int k = 0; k < 21; k++) d2[CP + 1 + 29 * i + 8 + k] = F[i].state_imu[k] - F[i].state_imu_zero[k];
  }
  for (int r = 0; r < dimI; r++) {
    double a = bM[r];
    for (int c = 0; c < dimI; c++) a += HM[(size_t)r * dimI + c] * d2[c];
    be[r] += a;
    for (int c = 0; c < dimI; c++) He[(size_t)r * dimI + c] += HM[(size_t)r * dimI + c];
  }
  /* Schur complement, :1093-1099 */
  double *Hs = (double *)malloc(sizeof(double) * (size_t)dimI * dimI), *bs = (double *)malloc(sizeof(double) * dimI);
  orc_imu_expand(n, H_sc, b_sc, Hs, bs);
  for (int i = 0; i < dimI; i++) He[(size_t)i * dimI + i] *= (1 + lambda);
  const double f = 1.0f / (1 + lambda); /* float literal over a double sum: a double quotient */
  for (size_t k = 0; k < (size_t)dimI * dimI; k++) He[k] -= Hs[k] * f;
  for (int k = 0; k < dimI; k++) be[k] -= bs[k];
  /* constraint rows, :1103-1110 */
  const int cdim = ncst;
  int full = dimI + cdim;
  for (int r = 0; r < dimI; r++) {
    for (int c = 0; c < dimI; c++) Hf[(size_t)r * cap + c] = He[(size_t)r * dimI + c];
    bf[r] = be[r];
  }
  for (int r = 0; r < cdim; r++) {
    for (int c = 0; c < dimI; c++) {
      Hf[(size_t)c * cap + dimI + r] = Jc[(size_t)r * dimI + c];
      Hf[(size_t)(dimI + r) * cap + c] = Jc[(size_t)r * dimI + c];
    }
    bf[dimI + r] = rc[r];
  }
  dim = full;
  /* remove the unconstrained states, :1113-1135: block moves in the reference's order (columns, then rows) */
  int vi = CP + (S->enable_scale_opt ? 0 : 1);
  double *tmp = (double *)malloc(sizeof(double) * (size_t)cap * (maxc > 29 ? maxc : 29));
  for (int i = 0; i <= n; i++) {
    int fi = (i < n) ? CP + 1 + 29 * i : CP + 1 + 29 * n;
    int vs = (i < n) ? (sv[i] ? 29 : 14) : cdim;
    for (int r = 0; r < dim; r++)
      for (int c = 0; c < vs; c++) tmp[(size_t)r * vs + c] = Hf[(size_t)r * cap + fi + c];
    for (int r = 0; r < dim; r++)
      for (int c = 0; c < vs; c++) Hf[(size_t)r * cap + vi + c] = tmp[(size_t)r * vs + c];
    for (int r = 0; r < vs; r++)
      for (int c = 0; c < dim; c++) tmp[(size_t)r * dim + c] = Hf[(size_t)(fi + r) * cap + c];
    for (int r = 0; r < vs; r++)
      for (int c = 0; c < dim; c++) Hf[(size_t)(vi + r) * cap + c] = tmp[(size_t)r * dim + c];
    for (int r = 0; r < vs; r++) tmp[r] = bf[fi + r];
    for (int r = 0; r < vs; r++) bf[vi + r] = tmp[r];
    vi += vs;
  }
  const int m = vi;
  /* Jacobi scaling + LDLT, :1144-1148 */
  double *A = (double *)malloc(sizeof(double) * (size_t)m * m), *rhs = (double *)malloc(sizeof(double) * m), *sol = (double *)malloc(sizeof(double) * m),
         *sI = (double *)malloc(sizeof(double) * m);
  for (int i = 0; i < m; i++) sI[i] = 1.0 / sqrt(Hf[(size_t)i * cap + i] + 10);
  for (int r = 0; r < m; r++) {
    for (int c = 0; c < m; c++) A[(size_t)r * m + c] = sI[r] * Hf[(size_t)r * cap + c] * sI[c];
    rhs[r] = sI[r] * bf[r];
  }
  orc_ldlt_solve(A, rhs, sol, m);
  for (int i = 0; i < m; i++) sol[i] *= sI[i];
  /* split, :1150-1167 */
  memset(x_out, 0, sizeof(double) * (CP + 8 * n));
  for (int i = 0; i < CP; i++) x_out[i] = sol[i];
  vi = CP;
  *scale_step = 0;
  if (!S->enable_scale_opt) { *scale_step = -sol[vi]; vi += 1; }
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 8; k++) x_out[CP + 8 * i + k] = sol[vi + k];
    vi += 8;
    memset(step_imu + 21 * i, 0, sizeof(double) * 21);
    for (int k = 0; k < 6; k++) step_imu[21 * i + k] = -sol[vi + k];
    vi += 6;
    if (sv[i]) {
      for (int k = 0; k < 15; k++) step_imu[21 * i + 6 + k] = -sol[vi + k];
      vi += 15;
    }
  }
  free(Himu); free(bimu); free(Jc); free(rc); free(sv); free(Hf); free(bf); free(He); free(be); free(d2); free(Hs); free(bs); free(tmp);
  free(A); free(rhs); free(sol); free(sI);
  return 0;
}

/* general n x n inverse by Gauss-Jordan elimination with partial pivoting (stand-in for Eigen's MatrixXd::inverse) */
static void dense_inverse(double *A, int n) {
  double *M = (double *)malloc(sizeof(double) * (size_t)n * 2 * n);
  for (int r = 0; r < n; r++)
    for (int c = 0; c < 2 * n; c++) M[(size_t)r * 2 * n + c] = c < n ? A[(size_t)r * n + c] : (c - n == r ? 1.0 : 0.0);
  for (int k = 0; k < n; k++) {
    int p = k;
    for (int r = k + 1; r < n; r++)
      if (fabs(M[(size_t)r * 2 * n + k]) > fabs(M[(size_t)p * 2 * n + k])) p = r;
    if (p != k)
      for (int c = 0; c < 2 * n; c++) { double t = M[(size_t)k * 2 * n + c]; M[(size_t)k * 2 * n + c] = M[(size_t)p * 2 * n + c]; M[(size_t)p * 2 * n + c] = t; }
    double d = M[(size_t)k * 2 * n + k];
    for (int c = 0; c < 2 * n; c++) M[(size_t)k * 2 * n + c] /= d;
    for (int r = 0; r < n; r++) {
      if (r == k) continue;
      double f = M[(size_t)r * 2 * n + k];
      if (f == 0) continue;
      for (int c = 0; c < 2 * n; c++) M[(size_t)r * 2 * n + c] -= f * M[(size_t)k * 2 * n + c];
    }
  }
  for (int r = 0; r < n; r++)
    for (int c = 0; c < n; c++) A[(size_t)r * n + c] = M[(size_t)r * 2 * n + n + c];
  free(M);
}

/* marginalizeFrame with IMU enabled, OB/EnergyFunctional.cpp:730-889 */
int orc_imu_marginalize_frame(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, int idx,
                              const double *delta, const double *prior8, const double *delta_prior8, double margWeightFac,
                              const double *HM_in, const double *bM_in, double *HM_out, double *bM_out) {
  const int dim = SOSF_IMU_DIM(n);
  double *HM = (double *)malloc(sizeof(double) * (size_t)dim * dim), *bM = (double *)malloc(sizeof(double) * dim);
  memcpy(HM, HM_in, sizeof(double) * (size_t)dim * dim);
  memcpy(bM, bM_in, sizeof(double) * dim);
  /* :748-785 */
  double *Hc = (double *)calloc((size_t)dim * dim, sizeof(double)), *bc = (double *)calloc(dim, sizeof(double)), *d2 = (double *)calloc(dim, sizeof(double));
  double *Jc = (double *)malloc(sizeof(double) * 6 * (size_t)dim), rc[6];
  int32_t *sv = (int32_t *)calloc(n, sizeof(int32_t));
  for (int i = 0; i < CP; i++) d2[i] = delta[i];
  if (C->scale_trapped) d2[CP] = C->scale - C->scale_zero;
  imu_hessian_frame(S, C, n, F, idx + 1, Hc, bc, Jc, rc, sv);
  for (int k = 0; k < 8; k++) d2[CP + 1 + 29 * (idx + 1) + k] = delta[CP + 8 * (idx + 1) + k];
  if (C->scale_trapped)
    for (int k = 0; k < 21; k++) d2[CP + 1 + 29 * (idx + 1) + 8 + k] = F[idx + 1].state_imu[k] - F[idx + 1].state_imu_zero[k];
  int spline_valid_idx = 0;
  if (idx > 0) {
    imu_hessian_frame(S, C, n, F, idx, Hc, bc, Jc, rc, sv);
    spline_valid_idx = sv[idx];
    for (int k = 0; k < 8; k++) d2[CP + 1 + 29 * (idx - 1) + k] = delta[CP + 8 * (idx - 1) + k];
    if (C->scale_trapped)
      for (int k = 0; k < 21; k++) d2[CP + 1 + 29 * (idx - 1) + 8 + k] = F[idx - 1].state_imu[k] - F[idx - 1].state_imu_zero[k];
  }
  for (int r = 0; r < dim; r++) {
    double a = 0;
    for (int c = 0; c < dim; c++) a += Hc[(size_t)r * dim + c] * d2[c];
    bc[r] -= a;
  }
  for (size_t k = 0; k < (size_t)dim * dim; k++) HM[k] += margWeightFac * Hc[k];
  for (int k = 0; k < dim; k++) bM[k] += margWeightFac * bc[k];
  /* move the keyframe's block to the end, :787-809 */
  const int args = CP + 1;
  int step = 29;
  int odim = args + n * step, ndim = odim - step;
  const int io = args + idx * step, ntail = step * (n - idx - 1);
  int *perm = (int *)malloc(sizeof(int) * odim);
  for (int k = 0; k < odim; k++) perm[k] = k < io ? k : (k < io + ntail ? k + step : k - ntail); /* new position k holds old perm[k] */
  double *H2 = (double *)malloc(sizeof(double) * (size_t)odim * odim), *b2 = (double *)malloc(sizeof(double) * odim);
  for (int r = 0; r < odim; r++) {
    b2[r] = bM[perm[r]];
    for (int c = 0; c < odim; c++) H2[(size_t)r * odim + c] = HM[(size_t)perm[r] * odim + perm[c]];
  }
  for (int k = 0; k < 8; k++) { /* :812-813 */
    H2[(size_t)(io + ntail + k) * odim + io + ntail + k] += prior8[k];
    b2[io + ntail + k] += prior8[k] * delta_prior8[k];
  }
  /* discard the spline part when nothing constrains it, :818-822 */
  int cur = odim;
  if (!((idx > 0) && spline_valid_idx)) { cur = odim - 15; step = 14; }
  double *Hs = (double *)malloc(sizeof(double) * (size_t)cur * cur), *bs = (double *)malloc(sizeof(double) * cur), *sv_ = (double *)malloc(sizeof(double) * cur);
  for (int r = 0; r < cur; r++) {
    sv_[r] = sqrt(fabs(H2[(size_t)r * odim + r]) + 10);
    bs[r] = b2[r];
  }
  for (int r = 0; r < cur; r++) {
    for (int c = 0; c < cur; c++) Hs[(size_t)r * cur + c] = (1.0 / sv_[r]) * H2[(size_t)r * odim + c] * (1.0 / sv_[c]);
    bs[r] = (1.0 / sv_[r]) * bs[r];
  }
  /* invert the bottom block, :837-841 (the two 0.5f * (hpi + hpi) lines are identities) */
  double *hpi = (double *)malloc(sizeof(double) * (size_t)step * step);
  for (int r = 0; r < step; r++)
    for (int c = 0; c < step; c++) hpi[(size_t)r * step + c] = Hs[(size_t)(ndim + r) * cur + ndim + c];
  dense_inverse(hpi, step);
  /* Schur complement, :844-848 */
  double *bli = (double *)malloc(sizeof(double) * (size_t)ndim * step);
  for (int r = 0; r < ndim; r++)
    for (int c = 0; c < step; c++) {
      double a = 0;
      for (int k = 0; k < step; k++) a += Hs[(size_t)(ndim + k) * cur + r] * hpi[(size_t)k * step + c];
      bli[(size_t)r * step + c] = a;
    }
  for (int r = 0; r < ndim; r++) {
    for (int c = 0; c < ndim; c++) {
      double a = 0;
      for (int k = 0; k < step; k++) a += bli[(size_t)r * step + k] * Hs[(size_t)(ndim + k) * cur + c];
      Hs[(size_t)r * cur + c] -= a;
    }
    double a = 0;
    for (int k = 0; k < step; k++) a += bli[(size_t)r * step + k] * bs[ndim + k];
    bs[r] -= a;
  }
  /* unscale and symmetrise, :851-857 */
  for (int r = 0; r < ndim; r++) {
    for (int c = 0; c < ndim; c++)
      HM_out[(size_t)r * ndim + c] = 0.5 * (sv_[r] * Hs[(size_t)r * cur + c] * sv_[c] + sv_[c] * Hs[(size_t)c * cur + r] * sv_[r]);
    bM_out[r] = sv_[r] * bs[r];
  }
  free(HM); free(bM); free(Hc); free(bc); free(d2); free(Jc); free(sv); free(perm); free(H2); free(b2); free(Hs); free(bs); free(sv_); free(hpi); free(bli);
  return 0;
}
