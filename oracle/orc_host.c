/*
 * oracle/orc_host.c -- TEST INFRASTRUCTURE (CPU oracle, "parity unpinned", see oracle.h).
 *
 * Host-level restatement of the Gauss-Newton loop around the backend:
 *   FullSystem::optimize / linearizeAll / setNewFrameEnergyTH / doStepFromBackup / backupState
 *                            FS/FullSystemOptimize.cpp:305-489, 125-182, 84-124, 185-257, 260-269
 *   FullSystem::setPrecalcValues -> FrameFramePrecalc::set    FS/FullSystem.cpp:1099-1107,
 *                                                             FS/HessianBlocks.cpp:431-461
 *   EnergyFunctional::setAdjointsF / setDeltaF / solveSystemF (visual part, IMU off)
 *                            OB/EnergyFunctional.cpp:42-103, 163-194, 1029-1184
 *   FrameHessian::setState / setEvalPT / getPrior             FS/HessianBlocks.h:196-260, 280-304
 *   CalibHessian::setValue                                    FS/HessianBlocks.h:476-491
 *   AffLight::fromToVecExposure                               util/NumType.h:149-171
 */
#include "orc_internal.h"
#include "../include/sos_slam_host.h"

/* orc_imu.c */
typedef void (*orc_imu_solve_tap_t)(const sosf_imu_settings *, const sosf_imu_calib *, int, const sosf_imu_frame *, const double *, const double *,
                                   const double *, const double *, const double *, const double *, const double *, double, const double *, double,
                                   const double *);
static orc_imu_solve_tap_t orc_imu_solve_tap_fn = 0;
/* ... and of the visual solve: (n, H_top + L (priors in), b, H_sc, b_sc, HM, bM, delta, lambda, x) */
typedef void (*orc_solve_tap_t)(int, const double *, const double *, const double *, const double *, const double *, const double *, const double *,
                               double, const double *);
static orc_solve_tap_t orc_solve_tap_fn = 0;
void orc_set_solve_tap(orc_solve_tap_t fn) { orc_solve_tap_fn = fn; }
/* test hook: called with the inputs and outputs of every IMU solve of orc_solve_system (inputs as the solve saw them) */
void orc_set_imu_solve_tap(orc_imu_solve_tap_t fn) { orc_imu_solve_tap_fn = fn; }
int orc_imu_solve(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *F, const double *H_top,
                  const double *b_top, const double *H_sc, const double *b_sc, const double *HM, const double *bM, const double *delta,
                  double lambda, double *x_out, double *scale_step, double *step_imu);

#include <stdio.h>
#include <stdlib.h>

/* reference defaults, util/settings.cpp:47-77 */
#define SETTING_initialRotPrior 1e11f
#define SETTING_initialTransPrior 1e10f
#define SETTING_initialAffBPrior 1e14f
#define SETTING_initialAffAPrior 1e14f
#define SETTING_thOptIterations 1.2f
#define SETTING_minOptIterations 1

static void to12(const orc_se3 *T, double *o) { memcpy(o, T->R, 9 * sizeof(double)); memcpy(o + 9, T->t, 3 * sizeof(double)); }
static orc_se3 from12(const double *i) { orc_se3 T; memcpy(T.R, i, 9 * sizeof(double)); memcpy(T.t, i + 9, 3 * sizeof(double)); return T; }

void orc_se3_exp12(const double *a, double *T12) { orc_se3 T = orc_se3_exp(a); to12(&T, T12); }
void orc_se3_log12(const double *T12, double *a) { orc_se3 T = from12(T12); orc_se3_log(&T, a); }
void orc_se3_adj12(const double *T12, double *Ad) { orc_se3 T = from12(T12); orc_se3_adj(&T, Ad); }
void orc_se3_mul12(const double *A, const double *B, double *C) { orc_se3 a = from12(A), b = from12(B); orc_se3 c = orc_se3_mul(&a, &b); to12(&c, C); }
void orc_se3_inv12(const double *A, double *C) { orc_se3 a = from12(A); orc_se3 c = orc_se3_inverse(&a); to12(&c, C); }
int orc_solve_ldlt(const double *A, const double *b, double *x, int n) { return orc_ldlt_solve(A, b, x, n); }

/* util/NumType.h:156-168 */
static void from_to_vec_exposure(float exposureF, float exposureT, double g2F_a, double g2F_b, double g2T_a,
                                 double g2T_b, double *out) {
  if (exposureF == 0 || exposureT == 0) exposureT = exposureF = 1;
  double a = exp(g2T_a - g2F_a) * exposureT / exposureF;
  double b = g2T_b - a * g2F_b;
  out[0] = a; out[1] = b;
}

static void calib_set_value(orc_window *W, const double *value) { /* FS/HessianBlocks.h:476-491 */
  for (int i = 0; i < 4; i++) W->c_value[i] = value[i];
  W->c_value_scaled[0] = SOS_SCALE_F * value[0];
  W->c_value_scaled[1] = SOS_SCALE_F * value[1];
  W->c_value_scaled[2] = SOS_SCALE_C * value[2];
  W->c_value_scaled[3] = SOS_SCALE_C * value[3];
  sos_calib *C = &W->calib;
  C->fxl = (float)W->c_value_scaled[0]; C->fyl = (float)W->c_value_scaled[1];
  C->cxl = (float)W->c_value_scaled[2]; C->cyl = (float)W->c_value_scaled[3];
  C->fxli = 1.0f / C->fxl; C->fyli = 1.0f / C->fyl;
  C->cxli = -C->cxl / C->fxl; C->cyli = -C->cyl / C->fyl;
  for (int i = 0; i < 4; i++) W->c_value_minus_value_zero[i] = W->c_value[i] - W->c_value_zero[i];
}

static void frame_set_state(orc_hframe *f, const double *state) { /* FS/HessianBlocks.h:217-230 */
  for (int i = 0; i < 10; i++) f->state[i] = state[i];
  for (int i = 0; i < 3; i++) f->state_scaled[i] = SOS_SCALE_XI_TRANS * state[i];
  for (int i = 3; i < 6; i++) f->state_scaled[i] = SOS_SCALE_XI_ROT * state[i];
  f->state_scaled[6] = SOS_SCALE_A * state[6];
  f->state_scaled[7] = SOS_SCALE_B * state[7];
  f->state_scaled[8] = SOS_SCALE_A * state[8];
  f->state_scaled[9] = SOS_SCALE_B * state[9];
  orc_se3 E = orc_se3_exp(f->state_scaled);
  f->PRE_camToWorld = orc_se3_mul(&E, &f->camToWorld_evalPT);
  f->PRE_worldToCam = orc_se3_inverse(&f->PRE_camToWorld);
}

static void frame_take_data(orc_window *W, orc_hframe *f) { /* getPrior FS/HessianBlocks.h:280-302 + EFFrame::takeData */
  double p[10];
  memset(p, 0, sizeof(p));
  if (f->frameID == 0) {
    p[0] = p[1] = p[2] = SETTING_initialTransPrior;
    p[3] = p[4] = p[5] = SETTING_initialRotPrior;
    p[6] = SETTING_initialAffAPrior;
    p[7] = SETTING_initialAffBPrior;
  } else {
    p[6] = W->prm.affineOptModeA < 0 ? SETTING_initialAffAPrior : W->prm.affineOptModeA;
    p[7] = W->prm.affineOptModeB < 0 ? SETTING_initialAffBPrior : W->prm.affineOptModeB;
  }
  for (int i = 0; i < 8; i++) {
    f->prior[i] = p[i];
    f->delta[i] = f->state[i] - f->state_zero[i];
    f->delta_prior[i] = f->state[i];
  }
}

void orc_precalc_pair(const double *hostEval12, const double *targetEval12, const double *hostPRE12,
                      const double *targetPRE12, const sos_calib *C, float host_ab, float target_ab,
                      const double *host_aff, const double *target_aff, double host_b0, sos_precalc *out,
                      float *distanceLL) { /* FS/HessianBlocks.cpp:431-461 */
  orc_se3 hE = from12(hostEval12), tE = from12(targetEval12), hP = from12(hostPRE12), tP = from12(targetPRE12);
  orc_se3 tEi = orc_se3_inverse(&tE);
  orc_se3 l0 = orc_se3_mul(&tEi, &hE);
  orc_se3 tPi = orc_se3_inverse(&tP); /* == PRE_worldToCam */
  orc_se3 l = orc_se3_mul(&tPi, &hP);
  float R[9], t[3];
  for (int i = 0; i < 9; i++) { out->PRE_RTll_0[i] = (float)l0.R[i]; R[i] = (float)l.R[i]; }
  for (int i = 0; i < 3; i++) { out->PRE_tTll_0[i] = (float)l0.t[i]; t[i] = (float)l.t[i]; }
  if (distanceLL) *distanceLL = (float)sqrt(l.t[0] * l.t[0] + l.t[1] * l.t[1] + l.t[2] * l.t[2]);
  /* K * R * K^-1 in float; K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1] */
  float K[9] = {C->fxl, 0, C->cxl, 0, C->fyl, C->cyl, 0, 0, 1};
  float Ki[9] = {C->fxli, 0, C->cxli, 0, C->fyli, C->cyli, 0, 0, 1};
  float KR[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) KR[3 * i + j] = K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j] + K[3 * i + 2] * R[6 + j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      out->PRE_KRKiTll[3 * i + j] = KR[3 * i] * Ki[j] + KR[3 * i + 1] * Ki[3 + j] + KR[3 * i + 2] * Ki[6 + j];
  for (int i = 0; i < 3; i++) out->PRE_KtTll[i] = K[3 * i] * t[0] + K[3 * i + 1] * t[1] + K[3 * i + 2] * t[2];
  double aff[2];
  from_to_vec_exposure(host_ab, target_ab, host_aff[0], host_aff[1], target_aff[0], target_aff[1], aff);
  out->PRE_aff_mode[0] = (float)aff[0];
  out->PRE_aff_mode[1] = (float)aff[1];
  out->PRE_b0_mode = (float)host_b0;
  out->pad = 0;
}

static void set_adjoints(orc_window *W) { /* OB/EnergyFunctional.cpp:42-103 */
  int n = W->n;
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) {
      orc_hframe *host = &W->hf[h], *target = &W->hf[t];
      orc_se3 w2t = orc_se3_inverse(&target->camToWorld_evalPT);
      double Ad[36];
      orc_se3_adj(&w2t, Ad);
      double AH[64], AT[64];
      memset(AH, 0, sizeof(AH)); memset(AT, 0, sizeof(AT));
      for (int i = 0; i < 8; i++) AH[9 * i] = AT[9 * i] = 1;
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { AH[8 * i + j] = Ad[6 * j + i]; AT[8 * i + j] = -Ad[6 * j + i]; }
      double aff[2];
      from_to_vec_exposure(host->ab_exposure, target->ab_exposure, host->state_zero[6] * SOS_SCALE_A,
                           host->state_zero[7] * SOS_SCALE_B, target->state_zero[6] * SOS_SCALE_A,
                           target->state_zero[7] * SOS_SCALE_B, aff);
      float a0 = (float)aff[0];
      AT[8 * 6 + 6] = -a0; AH[8 * 6 + 6] = a0; AT[8 * 7 + 7] = -1; AH[8 * 7 + 7] = a0;
      for (int j = 0; j < 8; j++) {
        for (int i = 0; i < 3; i++) { AH[8 * i + j] *= SOS_SCALE_XI_TRANS; AT[8 * i + j] *= SOS_SCALE_XI_TRANS; }
        for (int i = 3; i < 6; i++) { AH[8 * i + j] *= SOS_SCALE_XI_ROT; AT[8 * i + j] *= SOS_SCALE_XI_ROT; }
        AH[8 * 6 + j] *= SOS_SCALE_A; AT[8 * 6 + j] *= SOS_SCALE_A;
        AH[8 * 7 + j] *= SOS_SCALE_B; AT[8 * 7 + j] *= SOS_SCALE_B;
      }
      size_t idx = (size_t)(h + t * n);
      memcpy(&W->adHost[64 * idx], AH, sizeof(AH));
      memcpy(&W->adTarget[64 * idx], AT, sizeof(AT));
      for (int i = 0; i < 64; i++) { W->adHostF[64 * idx + i] = (float)AH[i]; W->adTargetF[64 * idx + i] = (float)AT[i]; }
    }
}

static void set_delta(orc_window *W) { /* OB/EnergyFunctional.cpp:163-194 */
  int n = W->n;
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) {
      size_t idx = (size_t)(h + t * n);
      float dh[8], dt[8];
      for (int i = 0; i < 8; i++) {
        dh[i] = (float)(W->hf[h].state[i] - W->hf[h].state_zero[i]);
        dt[i] = (float)(W->hf[t].state[i] - W->hf[t].state_zero[i]);
      }
      for (int j = 0; j < 8; j++) {
        float s1 = 0, s2 = 0;
        for (int i = 0; i < 8; i++) { s1 += dh[i] * W->adHostF[64 * idx + 8 * i + j]; s2 += dt[i] * W->adTargetF[64 * idx + 8 * i + j]; }
        W->adHTdeltaF[8 * idx + j] = s1 + s2;
      }
    }
  for (int i = 0; i < 4; i++) W->cDeltaF[i] = (float)W->c_value_minus_value_zero[i];
  for (int f = 0; f < n; f++)
    for (int i = 0; i < 8; i++) {
      W->hf[f].delta[i] = W->hf[f].state[i] - W->hf[f].state_zero[i];
      W->hf[f].delta_prior[i] = W->hf[f].state[i];
    }
  /* p->deltaF = idepth - idepth_zero (SCALE_IDEPTH = 1) */
  for (int p = 0; p < W->P; p++) W->pts[p].deltaF = W->pts[p].idepth_scaled - W->pts[p].idepth_zero_scaled;
}

void orc_host_precalc(orc_window *W) { /* FS/FullSystem.cpp:1099-1107 */
  int n = W->n;
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) {
      double hE[12], tE[12], hP[12], tP[12];
      to12(&W->hf[h].camToWorld_evalPT, hE); to12(&W->hf[t].camToWorld_evalPT, tE);
      to12(&W->hf[h].PRE_camToWorld, hP); to12(&W->hf[t].PRE_camToWorld, tP);
      double ha[2] = {W->hf[h].state_scaled[6], W->hf[h].state_scaled[7]};
      double ta[2] = {W->hf[t].state_scaled[6], W->hf[t].state_scaled[7]};
      orc_precalc_pair(hE, tE, hP, tP, &W->calib, W->hf[h].ab_exposure, W->hf[t].ab_exposure, ha, ta,
                       W->hf[h].state_zero[7] * SOS_SCALE_B, &W->precalc[h + n * t], 0);
    }
  set_delta(W);
}

void orc_host_init(orc_window *W, const orc_frame_init *frames, const double *calib_value_scaled,
                   const double *HM, const double *bM) {
  int n = W->n, dim = 4 + 8 * n;
  /* CalibHessian(): setValueScaled(initial); value_zero = value */
  double v[4] = {calib_value_scaled[0] / SOS_SCALE_F, calib_value_scaled[1] / SOS_SCALE_F,
                 calib_value_scaled[2] / SOS_SCALE_C, calib_value_scaled[3] / SOS_SCALE_C};
  v[0] = (1.0f / SOS_SCALE_F) * calib_value_scaled[0];
  v[1] = (1.0f / SOS_SCALE_F) * calib_value_scaled[1];
  v[2] = (1.0f / SOS_SCALE_C) * calib_value_scaled[2];
  v[3] = (1.0f / SOS_SCALE_C) * calib_value_scaled[3];
  for (int i = 0; i < 4; i++) W->c_value_zero[i] = v[i];
  calib_set_value(W, v);
  for (int f = 0; f < n; f++) {
    orc_hframe *F = &W->hf[f];
    F->camToWorld_evalPT = from12(frames[f].camToWorld);
    F->ab_exposure = frames[f].ab_exposure;
    F->frameID = frames[f].frameID;
    for (int i = 0; i < 10; i++) F->state_zero[i] = (i < 6) ? 0.0 : frames[f].state[i];
    frame_set_state(F, frames[f].state);
    memset(F->step, 0, sizeof(F->step));
    frame_take_data(W, F);
    W->frameEnergyTH[f] = frames[f].frameEnergyTH;
  }
  if (HM) memcpy(W->HM, HM, sizeof(double) * (size_t)dim * dim);
  if (bM) memcpy(W->bM, bM, sizeof(double) * (size_t)dim);
  set_adjoints(W);
  orc_host_precalc(W);
}

/* Rolling-window entry: a window that has lived through earlier keyframes.  Every frame brings its own FEJ point
 * (evalPT pose and state_zero differ from the current state), the calibration brings value and value_zero
 * (FS/HessianBlocks.h:453-475), and the prior HM / bM is the one earlier marginalisations left. */
void orc_host_init_ex(orc_window *W, const orc_frame_init_ex *frames, const double *calib_value, const double *calib_value_zero,
                      const double *HM, const double *bM) {
  int n = W->n, dim = 4 + 8 * n;
  for (int i = 0; i < 4; i++) W->c_value_zero[i] = calib_value_zero[i];
  calib_set_value(W, calib_value);
  for (int f = 0; f < n; f++) {
    orc_hframe *F = &W->hf[f];
    F->camToWorld_evalPT = from12(frames[f].camToWorld);
    F->ab_exposure = frames[f].ab_exposure;
    F->frameID = frames[f].frameID;
    for (int i = 0; i < 10; i++) F->state_zero[i] = frames[f].state_zero[i];
    frame_set_state(F, frames[f].state);
    memset(F->step, 0, sizeof(F->step));
    frame_take_data(W, F);
    W->frameEnergyTH[f] = frames[f].frameEnergyTH;
  }
  if (HM) memcpy(W->HM, HM, sizeof(double) * (size_t)dim * dim);
  if (bM) memcpy(W->bM, bM, sizeof(double) * (size_t)dim);
  set_adjoints(W);
  orc_host_precalc(W);
}
void orc_host_get_calib_value(orc_window *W, double *value4, double *value_zero4) {
  if (value4) memcpy(value4, W->c_value, sizeof(double) * 4);
  if (value_zero4) memcpy(value_zero4, W->c_value_zero, sizeof(double) * 4);
}
void orc_host_get_evalpt(orc_window *W, int f, double *c2w_evalPT12) { to12(&W->hf[f].camToWorld_evalPT, c2w_evalPT12); }

void orc_host_get_frame(orc_window *W, int f, double *c2w, double *state, double *state_zero, float *th) {
  if (c2w) to12(&W->hf[f].PRE_camToWorld, c2w);
  if (state) memcpy(state, W->hf[f].state, sizeof(double) * 10);
  if (state_zero) memcpy(state_zero, W->hf[f].state_zero, sizeof(double) * 10);
  if (th) *th = W->frameEnergyTH[f];
}
void orc_host_set_truth_mode(orc_window *W, int on) { W->truth_mode = on; }
void orc_host_get_calib(orc_window *W, double *vs) { memcpy(vs, W->c_value_scaled, sizeof(double) * 4); }
const sos_precalc *orc_host_get_precalc(orc_window *W) { return W->precalc; }
const float *orc_host_get_adHTdeltaF(orc_window *W) { return W->adHTdeltaF; }
const double *orc_host_get_adHost(orc_window *W) { return W->adHost; }
const double *orc_host_get_adTarget(orc_window *W) { return W->adTarget; }
const double *orc_host_get_lastX(orc_window *W) { return W->lastX; }
void orc_host_get_frame_prior(orc_window *W, int frame, double *prior8, double *delta_prior8) {
  memcpy(prior8, W->hf[frame].prior, sizeof(double) * 8);
  memcpy(delta_prior8, W->hf[frame].delta_prior, sizeof(double) * 8);
}
void orc_host_get_HM(orc_window *W, double *HM, double *bM) {
  int dim = 4 + 8 * W->n;
  memcpy(HM, W->HM, sizeof(double) * (size_t)dim * dim);
  memcpy(bM, W->bM, sizeof(double) * (size_t)dim);
}

static int cmp_float(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

static void set_new_frame_energy_th(orc_window *W) { /* FS/FullSystemOptimize.cpp:84-124 */
  int newest = W->n - 1;
  float *v = (float *)malloc(sizeof(float) * (size_t)(W->R > 0 ? W->R : 1));
  int cnt = 0;
  for (int r = 0; r < W->R; r++) {
    if (W->res[r].flags & (SOS_RF_LINEARIZED | ORC_RF_REMOVED)) continue;
    if (W->newEnergyWO[r] >= 0 && W->res[r].target == newest) v[cnt++] = W->newEnergyWO[r];
  }
  if (cnt == 0) {
    W->frameEnergyTH[newest] = 12 * 12 * 8;
    free(v);
    return;
  }
  int nthIdx = (int)(W->prm.frameEnergyTHN * cnt);
  qsort(v, (size_t)cnt, sizeof(float), cmp_float); /* nth_element: value at sorted position */
  float nthElement = sqrtf(v[nthIdx]);
  float th = nthElement * W->prm.frameEnergyTHFacMedian;
  th = 26.0f * W->prm.frameEnergyTHConstWeight + th * (1 - W->prm.frameEnergyTHConstWeight);
  th = th * th;
  th *= W->prm.overallEnergyTHWeight * W->prm.overallEnergyTHWeight;
  W->frameEnergyTH[newest] = th;
  free(v);
}

/* linearizeAll, FS/FullSystemOptimize.cpp:125-182 */
static double host_linearize_all(orc_window *W, int fix, int nthreads) {
  double E = orc_linearize_all(W, W->frameEnergyTH, nthreads);
  if (fix) { /* linearizeAll_Reductor with fixLinearization, :51-75 */
    for (int r = 0; r < W->R; r++) {
      sos_resid *res = &W->res[r];
      if (res->flags & (SOS_RF_LINEARIZED | ORC_RF_REMOVED)) continue;
      orc_apply_res_one(W, r);
      if (res->flags & SOS_RF_ACTIVE) {
        if (res->flags & SOS_RF_ISNEW) {
          const sos_point *p = &W->pts[res->point];
          const sos_precalc *pc = &W->precalc[res->host + W->n * res->target];
          const float *K = pc->PRE_KRKiTll, *Kt = pc->PRE_KtTll;
          float inf0 = K[0] * p->u + K[1] * p->v + K[2], inf1 = K[3] * p->u + K[4] * p->v + K[5],
                inf2 = K[6] * p->u + K[7] * p->v + K[8];
          float q0 = inf0 + Kt[0] * p->idepth_scaled, q1 = inf1 + Kt[1] * p->idepth_scaled,
                q2 = inf2 + Kt[2] * p->idepth_scaled;
          float dx = inf0 / inf2 - q0 / q2, dy = inf1 / inf2 - q1 / q2;
          float relBS = (float)(0.01 * sqrtf(dx * dx + dy * dy));
          if (relBS > W->maxRelBaseline[res->point]) W->maxRelBaseline[res->point] = relBS;
          W->numGoodResiduals[res->point]++;
        }
      } else {
        res->flags |= 0x200u; /* toRemove[tid].push_back; dropped after setNewFrameEnergyTH */
      }
    }
  }
  set_new_frame_energy_th(W);
  if (fix) /* :148-179: ef->dropResidual for every residual in toRemove */
    for (int r = 0; r < W->R; r++)
      if (W->res[r].flags & 0x200u) W->res[r].flags = (W->res[r].flags & ~0x200u) | ORC_RF_REMOVED;
  return E;
}

static void host_apply_res(orc_window *W) {
  for (int r = 0; r < W->R; r++)
    if (!(W->res[r].flags & (SOS_RF_LINEARIZED | ORC_RF_REMOVED))) orc_apply_res_one(W, r);
}

/* solveSystemF, OB/EnergyFunctional.cpp:1029-1184 (IMU off) */
static void solve_system(orc_window *W, int nthreads) {
  int n = W->n, dim = 4 + 8 * n;
  size_t dd = (size_t)dim * dim;
  double lambda = 1e-5;
  double *HA = (double *)malloc(sizeof(double) * dd), *HL = (double *)malloc(sizeof(double) * dd),
         *Hsc = (double *)malloc(sizeof(double) * dd);
  double *bA = (double *)malloc(sizeof(double) * dim), *bL = (double *)malloc(sizeof(double) * dim),
         *bsc = (double *)malloc(sizeof(double) * dim);
  orc_accumulate(W, HA, bA, HL, bL, Hsc, bsc, &W->resInA, &W->resInL, W->truth_mode, nthreads);
  /* priors of the L stitch, OB/AccumulatedTopHessian.cpp:292-300 */
  for (int i = 0; i < 4; i++) {
    HL[(size_t)i * dim + i] += (double)W->prm.initialCalibHessian;
    bL[i] += (double)W->prm.initialCalibHessian * (double)W->cDeltaF[i];
  }
  for (int h = 0; h < n; h++)
    for (int i = 0; i < 8; i++) {
      HL[(size_t)(4 + 8 * h + i) * dim + 4 + 8 * h + i] += W->hf[h].prior[i];
      bL[4 + 8 * h + i] += W->hf[h].prior[i] * W->hf[h].delta_prior[i];
    }
  double *H = (double *)malloc(sizeof(double) * dd), *b = (double *)malloc(sizeof(double) * dim);
  for (size_t i = 0; i < dd; i++) H[i] = HL[i] + HA[i];
  for (int i = 0; i < dim; i++) b[i] = bL[i] + bA[i];
  /* marginalization prior: bM + HM*delta, :1069-1091 */
  double *delta = (double *)malloc(sizeof(double) * dim);
  for (int i = 0; i < 4; i++) delta[i] = (double)W->cDeltaF[i];
  for (int h = 0; h < n; h++)
    for (int i = 0; i < 8; i++) delta[4 + 8 * h + i] = W->hf[h].delta[i];
  if (W->imuS) { /* setting_enable_imu && imu_initialized: :1053-1171 through the restatement in orc_imu.c */
    const sosf_imu_settings *S = (const sosf_imu_settings *)W->imuS;
    sosf_imu_calib *C = (sosf_imu_calib *)W->imuC;
    sosf_imu_frame *F = (sosf_imu_frame *)W->imuF;
    for (int h = 0; h < n; h++) {
      memcpy(F[h].camToWorld, W->hf[h].PRE_camToWorld.R, sizeof(double) * 9);
      memcpy(F[h].camToWorld + 9, W->hf[h].PRE_camToWorld.t, sizeof(double) * 3);
      memcpy(F[h].evalPT_R, W->hf[h].camToWorld_evalPT.R, sizeof(double) * 9);
    }
    double *x = (double *)malloc(sizeof(double) * dim);
    free(W->imuStep);
    W->imuStep = (double *)calloc((size_t)21 * n, sizeof(double));
    orc_imu_solve(S, C, n, F, H, b, Hsc, bsc, W->imuHM, W->imuBM, delta, lambda, x, &W->imuScaleStep, W->imuStep);
    if (orc_imu_solve_tap_fn)  /* tests: the systems a running chain actually solves, before the states are stepped */
      orc_imu_solve_tap_fn(S, C, n, F, H, b, Hsc, bsc, W->imuHM, W->imuBM, delta, lambda, x, W->imuScaleStep, W->imuStep);
    memcpy(W->lastX, x, sizeof(double) * dim);
    for (int i = 0; i < 4; i++) W->c_step[i] = -x[i];
    for (int h = 0; h < n; h++) {
      for (int i = 0; i < 8; i++) W->hf[h].step[i] = -x[4 + 8 * h + i];
      W->hf[h].step[8] = W->hf[h].step[9] = 0;
    }
    /* IMU and scale update of doStepFromBackup with unit step factors, FS/FullSystemOptimize.cpp:218-230 */
    C->scale += W->imuScaleStep;
    for (int h = 0; h < n; h++)
      for (int k = 0; k < 21; k++) F[h].state_imu[k] += W->imuStep[(size_t)21 * h + k];
    orc_resubstitute(W, x, 0, nthreads);
    free(HA); free(HL); free(Hsc); free(bA); free(bL); free(bsc); free(H); free(b); free(delta); free(x);
    return;
  }
  for (int i = 0; i < dim; i++) {
    double s = W->bM[i];
    for (int j = 0; j < dim; j++) s += W->HM[(size_t)i * dim + j] * delta[j];
    b[i] += s;
  }
  double *H0 = NULL, *b0 = NULL; /* (test hook: the pieces as they are before the assembly below) */
  if (orc_solve_tap_fn) {
    H0 = (double *)malloc(sizeof(double) * dd); b0 = (double *)malloc(sizeof(double) * dim);
    for (size_t i = 0; i < dd; i++) H0[i] = HL[i] + HA[i];
    for (int i = 0; i < dim; i++) b0[i] = bL[i] + bA[i];
  }
  for (size_t i = 0; i < dd; i++) H[i] += W->HM[i];
  for (int i = 0; i < dim; i++) H[(size_t)i * dim + i] *= (1 + lambda);
  double isc = 1.0f / (1 + lambda);
  for (size_t i = 0; i < dd; i++) H[i] -= Hsc[i] * isc;
  for (int i = 0; i < dim; i++) b[i] -= bsc[i];
  /* :1143-1148 */
  double *S = (double *)malloc(sizeof(double) * dim), *x = (double *)malloc(sizeof(double) * dim);
  for (int i = 0; i < dim; i++) S[i] = 1.0 / sqrt(H[(size_t)i * dim + i] + 10);
  for (int i = 0; i < dim; i++) {
    for (int j = 0; j < dim; j++) H[(size_t)i * dim + j] *= S[i] * S[j];
    b[i] *= S[i];
  }
  orc_ldlt_solve(H, b, x, dim);
  for (int i = 0; i < dim; i++) x[i] *= S[i];
  if (orc_solve_tap_fn) {
    orc_solve_tap_fn(n, H0, b0, Hsc, bsc, W->HM, W->bM, delta, lambda, x);
    free(H0); free(b0);
  }
  memcpy(W->lastX, x, sizeof(double) * dim);
  /* resubstituteF_MT, :496-524 */
  for (int i = 0; i < 4; i++) W->c_step[i] = -x[i];
  for (int h = 0; h < n; h++) {
    for (int i = 0; i < 8; i++) W->hf[h].step[i] = -x[4 + 8 * h + i];
    W->hf[h].step[8] = W->hf[h].step[9] = 0;
  }
  orc_resubstitute(W, x, 0, nthreads);
  free(HA); free(HL); free(Hsc); free(bA); free(bL); free(bsc); free(H); free(b); free(delta); free(S); free(x);
}

void orc_host_set_imu(orc_window *W, const void *S, void *C, void *frames, const double *HM, const double *bM) {
  W->imuS = S; W->imuC = C; W->imuF = frames; W->imuHM = HM; W->imuBM = bM;
}

static void backup_state(orc_window *W) { /* :260-269 */
  memcpy(W->c_value_backup, W->c_value, sizeof(W->c_value));
  for (int f = 0; f < W->n; f++) memcpy(W->hf[f].state_backup, W->hf[f].state, sizeof(double) * 10);
  for (int p = 0; p < W->P; p++) W->idepth_backup[p] = W->pts[p].idepth_scaled;
}

static int do_step_from_backup(orc_window *W) { /* :185-257 with all step factors = 1 */
  float sumA = 0, sumB = 0, sumT = 0, sumR = 0, sumID = 0, numID = 0, sumNID = 0;
  double v[4];
  for (int i = 0; i < 4; i++) v[i] = W->c_value_backup[i] + 1.0f * W->c_step[i];
  calib_set_value(W, v);
  for (int f = 0; f < W->n; f++) {
    orc_hframe *F = &W->hf[f];
    double s[10];
    for (int i = 0; i < 10; i++) s[i] = F->state_backup[i] + 1.0 * F->step[i];
    frame_set_state(F, s);
    sumA += F->step[6] * F->step[6];
    sumB += F->step[7] * F->step[7];
    sumT += F->step[0] * F->step[0] + F->step[1] * F->step[1] + F->step[2] * F->step[2];
    sumR += F->step[3] * F->step[3] + F->step[4] * F->step[4] + F->step[5] * F->step[5];
  }
  for (int p = 0; p < W->P; p++) {
    float nid = W->idepth_backup[p] + 1.0f * W->step[p];
    W->pts[p].idepth_scaled = nid;      /* setIdepth */
    sumID += W->step[p] * W->step[p];
    sumNID += fabsf(W->idepth_backup[p]);
    numID++;
    W->pts[p].idepth_zero_scaled = nid; /* setIdepthZero */
  }
  sumA /= W->n; sumB /= W->n; sumR /= W->n; sumT /= W->n;
  sumID /= numID; sumNID /= numID;
  (void)sumID;
  orc_host_precalc(W);
  return sqrtf(sumA) < 0.0005 * SETTING_thOptIterations && sqrtf(sumB) < 0.00005 * SETTING_thOptIterations &&
         sqrtf(sumR) < 0.00005 * SETTING_thOptIterations &&
         sqrtf(sumT) * sumNID < 0.00005 * SETTING_thOptIterations;
}

/* one loop body of FS/FullSystemOptimize.cpp:358-413 (forceAceptStep = true) */
int orc_gn_iteration(orc_window *W, int iteration, int nthreads) {
  (void)iteration;
  backup_state(W);
  solve_system(W, nthreads);
  int canbreak = do_step_from_backup(W);
  host_linearize_all(W, 0, nthreads);
  host_apply_res(W);
  return canbreak;
}

/* calcLEnergyF_MT (OB/EnergyFunctional.cpp:626-642) and calcMEnergyF (:553-561): only evaluated when
 * setting_forceAceptStep is off (FS/FullSystemOptimize.cpp:289-296, 498-504) */
static double host_calc_lenergy(orc_window *W) {
  double E = 0;
  for (int f = 0; f < W->n; f++)
    for (int i = 0; i < 8; i++) E += W->hf[f].delta_prior[i] * W->hf[f].prior[i] * W->hf[f].delta_prior[i];
  float e4 = 0;
  for (int i = 0; i < 4; i++) e4 += W->cDeltaF[i] * (float)W->prm.initialCalibHessian * W->cDeltaF[i];
  E += e4;
  return E + orc_calc_lenergy(W);
}
static double host_calc_menergy(orc_window *W) {
  int n = W->n, dim = 4 + 8 * n;
  double *delta = (double *)malloc(sizeof(double) * dim);
  for (int i = 0; i < 4; i++) delta[i] = (double)W->cDeltaF[i];
  for (int h = 0; h < n; h++)
    for (int i = 0; i < 8; i++) delta[4 + 8 * h + i] = W->hf[h].delta[i];
  double e = 0;
  for (int i = 0; i < dim; i++) {
    double s = 2 * W->bM[i];
    for (int j = 0; j < dim; j++) s += W->HM[(size_t)i * dim + j] * delta[j];
    e += delta[i] * s;
  }
  free(delta);
  return e;
}

static void load_state_backup(orc_window *W) { /* loadSateBackup, FS/FullSystemOptimize.cpp:271-287 */
  calib_set_value(W, W->c_value_backup);
  for (int f = 0; f < W->n; f++) frame_set_state(&W->hf[f], W->hf[f].state_backup);
  for (int p = 0; p < W->P; p++) {
    W->pts[p].idepth_scaled = W->idepth_backup[p];
    W->pts[p].idepth_zero_scaled = W->idepth_backup[p];
  }
  orc_host_precalc(W);
}

/* FullSystem::optimize with setting_forceAceptStep selectable (FS/FullSystemOptimize.cpp:305-425): with forceAccept == 0
 * a step is kept only when E_photometric + E_L + E_M decreases; otherwise the backup is restored and the window is
 * linearised again at the old state.  rejected_out: number of rejected steps. */
float orc_optimize_ex(orc_window *W, int mnumOptIts, int nthreads, int forceAccept, int *iters_out, int *rejected_out) {
  int n = W->n;
  if (iters_out) *iters_out = 0;
  if (rejected_out) *rejected_out = 0;
  if (n < 2) return 0;
  if (n < 3) mnumOptIts = 20;
  if (n < 4) mnumOptIts = 15;
  orc_reset_oob(W);
  double lastEnergy = host_linearize_all(W, 0, nthreads);
  double lastEnergyL = forceAccept ? 0 : host_calc_lenergy(W);
  double lastEnergyM = forceAccept ? 0 : host_calc_menergy(W);
  host_apply_res(W);
  int it = 0, rej = 0;
  for (int iteration = 0; iteration < mnumOptIts; iteration++) {
    backup_state(W);
    solve_system(W, nthreads);
    int canbreak = do_step_from_backup(W);
    double newEnergy = host_linearize_all(W, 0, nthreads);
    double newEnergyL = forceAccept ? 0 : host_calc_lenergy(W);
    double newEnergyM = forceAccept ? 0 : host_calc_menergy(W);
    if (getenv("ORC_DEBUG"))
      fprintf(stderr, "[orc_optimize_ex] it %d new %.9g (%.9g %.9g %.9g) last %.9g (%.9g %.9g %.9g)\n", iteration, newEnergy + newEnergyL + newEnergyM,
              newEnergy, newEnergyL, newEnergyM, lastEnergy + lastEnergyL + lastEnergyM, lastEnergy, lastEnergyL, lastEnergyM);
    if (forceAccept || (newEnergy + newEnergyL + newEnergyM < lastEnergy + lastEnergyL + lastEnergyM)) {
      host_apply_res(W);
      lastEnergy = newEnergy; lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
    } else {
      load_state_backup(W);
      lastEnergy = host_linearize_all(W, 0, nthreads);
      lastEnergyL = host_calc_lenergy(W);
      lastEnergyM = host_calc_menergy(W);
      rej++;
    }
    it++;
    if (canbreak && iteration >= SETTING_minOptIterations) break;
  }
  if (iters_out) *iters_out = it;
  if (rejected_out) *rejected_out = rej;
  orc_hframe *L = &W->hf[n - 1];
  double nz[10];
  memset(nz, 0, sizeof(nz));
  nz[6] = L->state[6]; nz[7] = L->state[7];
  L->camToWorld_evalPT = L->PRE_camToWorld;
  frame_set_state(L, nz);
  memcpy(L->state_zero, nz, sizeof(nz));
  set_adjoints(W);
  orc_host_precalc(W);
  lastEnergy = host_linearize_all(W, 1, nthreads);
  return sqrtf((float)(lastEnergy / (8 * W->resInA)));
}

float orc_optimize(orc_window *W, int mnumOptIts, int nthreads, int *iters_out) {
  int n = W->n;
  if (iters_out) *iters_out = 0;
  if (n < 2) return 0;
  if (n < 3) mnumOptIts = 20;
  if (n < 4) mnumOptIts = 15;
  orc_reset_oob(W); /* :316-329 */
  host_linearize_all(W, 0, nthreads);
  host_apply_res(W);
  int it = 0;
  for (int iteration = 0; iteration < mnumOptIts; iteration++) {
    int canbreak = orc_gn_iteration(W, iteration, nthreads);
    it++;
    if (canbreak && iteration >= SETTING_minOptIterations) break;
  }
  if (iters_out) *iters_out = it;
  /* :415-425 */
  orc_hframe *L = &W->hf[n - 1];
  double nz[10];
  memset(nz, 0, sizeof(nz));
  nz[6] = L->state[6]; nz[7] = L->state[7];
  L->camToWorld_evalPT = L->PRE_camToWorld; /* setEvalPT */
  frame_set_state(L, nz);
  memcpy(L->state_zero, nz, sizeof(nz));
  set_adjoints(W);
  orc_host_precalc(W);
  double lastEnergy = host_linearize_all(W, 1, nthreads);
  return sqrtf((float)(lastEnergy / (8 * W->resInA)));
}

/* ================================================================================================
 * keyframe-rate marginalisation (backend part of FullSystem::makeKeyFrame, FS/FullSystem.cpp:899-931)
 * ============================================================================================== */
#define SETTING_minIdepthH_marg 50.0f /* U/settings.cpp:62 */

static void remove_point(orc_window *W, int p) { /* EnergyFunctional::removePoint, OB/EnergyFunctional.cpp:954-971 */
  for (int r = W->pt_begin[p]; r < W->pt_begin[p + 1]; r++) W->res[r].flags |= ORC_RF_REMOVED;
}

/* ef->dropPointsF for an explicit list (OB/EnergyFunctional.cpp:938-952) */
void orc_host_drop_points(orc_window *W, const int32_t *pointIdx, int count) {
  for (int k = 0; k < count; k++) remove_point(W, pointIdx[k]);
}

/* FullSystem::flagPointsForRemoval restricted to an explicit list of inlier points (FS/FullSystem.cpp:573-596),
 * ef->dropPointsF (:909) and ef->marginalizePointsF (:912; OB/EnergyFunctional.cpp:891-936, IMU off).
 * marg_flag_out[k] = 1 if point k was marginalised, 0 if it was dropped (idepth_hessian too small). */
void orc_host_marginalize_points(orc_window *W, const int32_t *pointIdx, int count, int32_t *marg_flag_out) {
  int n = W->n, dim = 4 + 8 * n;
  size_t dd = (size_t)dim * dim;
  int32_t *marg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(count > 0 ? count : 1));
  int nmarg = 0;
  for (int k = 0; k < count; k++) {
    int p = pointIdx[k];
    for (int r = W->pt_begin[p]; r < W->pt_begin[p + 1]; r++) {
      if (W->res[r].flags & ORC_RF_REMOVED) continue; /* not in ph->residuals any more */
      orc_reset_oob_one(W, r);
      W->retEnergy[r] = orc_linearize_one(W, r, W->frameEnergyTH);
      W->res[r].flags &= ~SOS_RF_LINEARIZED;
      orc_apply_res_one(W, r);
      if (W->res[r].flags & SOS_RF_ACTIVE) {
        int32_t rr = r;
        orc_fix_linearization(W, &rr, 1);
      }
    }
    if (W->idepth_hessian[p] > SETTING_minIdepthH_marg) {
      marg[nmarg++] = p;
      if (marg_flag_out) marg_flag_out[k] = 1;
    } else {
      remove_point(W, p); /* PS_DROP -> dropPointsF */
      if (marg_flag_out) marg_flag_out[k] = 0;
    }
  }
  /* marginalizePointsF */
  for (int k = 0; k < nmarg; k++) W->pts[marg[k]].priorF *= W->prm.idepthFixPriorMargFac;
  double *M = (double *)malloc(sizeof(double) * dd), *Msc = (double *)malloc(sizeof(double) * dd);
  double *Mb = (double *)malloc(sizeof(double) * dim), *Mbsc = (double *)malloc(sizeof(double) * dim);
  int resInM = 0;
  orc_accumulate_marg(W, marg, nmarg, M, Mb, Msc, Mbsc, &resInM);
  for (int k = 0; k < nmarg; k++) remove_point(W, marg[k]);
  for (size_t i = 0; i < dd; i++) W->HM[i] += W->prm.margWeightFac * (M[i] - Msc[i]);
  for (int i = 0; i < dim; i++) W->bM[i] += W->prm.margWeightFac * (Mb[i] - Mbsc[i]);
  free(M); free(Msc); free(Mb); free(Mbsc); free(marg);
}

/* EnergyFunctional::marginalizeFrame (OB/EnergyFunctional.cpp:730-889, IMU off), prior part only: the window
 * itself is not re-indexed, the (dim-8)-dimensional HM / bM the reference would hold afterwards are returned. */
void orc_host_marginalize_frame_prior(orc_window *W, int frameIdx, double *HM_out, double *bM_out) {
  int n = W->n, step = 8, odim = 4 + n * step, ndim = odim - step;
  int io = 4 + frameIdx * step, ntail = step * (n - frameIdx - 1);
  int *perm = (int *)malloc(sizeof(int) * odim);
  /* :797-811: the frame's block is moved behind the tail (three block copies = this permutation) */
  for (int i = 0; i < io; i++) perm[i] = i;
  for (int i = 0; i < ntail; i++) perm[io + i] = io + step + i;
  for (int i = 0; i < step; i++) perm[io + ntail + i] = io + i;
  double *H = (double *)malloc(sizeof(double) * (size_t)odim * odim), *b = (double *)malloc(sizeof(double) * odim);
  for (int i = 0; i < odim; i++) {
    b[i] = W->bM[perm[i]];
    for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = W->HM[(size_t)perm[i] * odim + perm[j]];
  }
  const orc_hframe *fh = &W->hf[frameIdx];
  for (int i = 0; i < 8; i++) { /* :814-815 */
    H[(size_t)(ndim + i) * odim + ndim + i] += fh->prior[i];
    b[ndim + i] += fh->prior[i] * fh->delta_prior[i];
  }
  double *SVec = (double *)malloc(sizeof(double) * odim), *SVecI = (double *)malloc(sizeof(double) * odim);
  for (int i = 0; i < odim; i++) { /* :826-828 */
    SVec[i] = sqrt(fabs(H[(size_t)i * odim + i]) + 10);
    SVecI[i] = 1.0 / SVec[i];
  }
  for (int i = 0; i < odim; i++) {
    for (int j = 0; j < odim; j++) H[(size_t)i * odim + j] = SVecI[i] * H[(size_t)i * odim + j] * SVecI[j];
    b[i] = SVecI[i] * b[i];
  }
  double hpi[64], hpinv[64];
  for (int i = 0; i < step; i++)
    for (int j = 0; j < step; j++) hpi[i * step + j] = H[(size_t)(ndim + i) * odim + ndim + j];
  orc_mat_inverse(hpi, hpinv, step); /* :840-843 (the two 0.5*(hpi+hpi) lines are identities) */
  double *bli = (double *)malloc(sizeof(double) * (size_t)ndim * step);
  for (int i = 0; i < ndim; i++) /* bli = bottomLeft^T * hpi, :847 */
    for (int j = 0; j < step; j++) {
      double s = 0;
      for (int k = 0; k < step; k++) s += H[(size_t)(ndim + k) * odim + i] * hpinv[k * step + j];
      bli[(size_t)i * step + j] = s;
    }
  for (int i = 0; i < ndim; i++) { /* :848-850 */
    for (int j = 0; j < ndim; j++) {
      double s = 0;
      for (int k = 0; k < step; k++) s += bli[(size_t)i * step + k] * H[(size_t)(ndim + k) * odim + j];
      H[(size_t)i * odim + j] -= s;
    }
    double s = 0;
    for (int k = 0; k < step; k++) s += bli[(size_t)i * step + k] * b[ndim + k];
    b[i] -= s;
  }
  for (int i = 0; i < ndim; i++) /* un-scale, symmetrise: :853-859 */
    for (int j = 0; j < ndim; j++) {
      double hij = SVec[i] * H[(size_t)i * odim + j] * SVec[j], hji = SVec[j] * H[(size_t)j * odim + i] * SVec[i];
      HM_out[(size_t)i * ndim + j] = 0.5 * (hij + hji);
    }
  for (int i = 0; i < ndim; i++) bM_out[i] = SVec[i] * b[i];
  free(perm); free(H); free(b); free(SVec); free(SVecI); free(bli);
}
