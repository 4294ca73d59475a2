/*
 * oracle/orc_backend.c -- TEST INFRASTRUCTURE (CPU oracle, "parity unpinned", see oracle.h).
 *
 * Sequential restatement of the reference's OptimizationBackend hot path, kernel level:
 *   linearize            FS/Residuals.cpp:77-271, FS/ResidualProjections.h:43-73, util/globalFuncs.h:68-82
 *   applyRes/takeDataF   FS/Residuals.cpp:304-321, OB/EnergyFunctionalStructs.cpp:36-45
 *   fixLinearizationF    OB/EnergyFunctionalStructs.cpp:75-103
 *   accumulators         OB/MatrixAccumulators.h:33-78 (XX), 152-202 (X), 80-150 (11), 744-1170 (Approx)
 *   top addPoint<mode>   OB/AccumulatedTopHessian.cpp:35-147, stitch :231-301
 *   SC addPoint          OB/AccumulatedSCHessian.cpp:32-79, stitch :80-158
 *   resubstitute         OB/EnergyFunctional.cpp:496-551
 *   calcLEnergyPt        OB/EnergyFunctional.cpp:563-624
 *   marginalizePointsF   OB/EnergyFunctional.cpp:891-936 (accumulation part)
 * Compile with -ffp-contract=off (see oracle/Makefile).
 */
#include "orc_internal.h"

#include <stdio.h>

static const int orc_pattern[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0},
                                      {0, 0},  {2, 0},   {-1, 1}, {0, 2}}; /* util/settings.cpp:307-317 [8] */

/* ================================================================================================
 * window
 * ============================================================================================== */
orc_window *orc_window_create(const sos_params *prm, int n, int P, const sos_point *pts, int R,
                              const sos_resid *res) {
  orc_window *W = (orc_window *)calloc(1, sizeof(orc_window));
  W->prm = *prm;
  W->n = n; W->P = P; W->R = R;
  W->pts = (sos_point *)malloc(sizeof(sos_point) * (size_t)(P > 0 ? P : 1));
  W->res = (sos_resid *)malloc(sizeof(sos_resid) * (size_t)(R > 0 ? R : 1));
  memcpy(W->pts, pts, sizeof(sos_point) * (size_t)P);
  memcpy(W->res, res, sizeof(sos_resid) * (size_t)R);
  W->pt_begin = (int *)calloc((size_t)P + 2, sizeof(int));
  /* residuals are contiguous per point (checked) */
  {
    int r = 0;
    for (int p = 0; p < P; p++) {
      W->pt_begin[p] = r;
      while (r < R && res[r].point == p) r++;
    }
    W->pt_begin[P] = r;
    if (r != R) fprintf(stderr, "orc_window_create: residuals not grouped by point (%d of %d)\n", r, R);
  }
  size_t Rn = (size_t)(R > 0 ? R : 1), Pn = (size_t)(P > 0 ? P : 1), nn = (size_t)n * n;
  W->J = (sos_rawjac *)calloc(Rn, sizeof(sos_rawjac));
  W->Jn = (sos_rawjac *)calloc(Rn, sizeof(sos_rawjac));
  W->res_toZeroF = (float *)calloc(Rn * 8, sizeof(float));
  W->JpJdF = (float *)calloc(Rn * 8, sizeof(float));
  W->newState = (int32_t *)calloc(Rn, sizeof(int32_t));
  W->newEnergy = (float *)calloc(Rn, sizeof(float));
  W->newEnergyWO = (float *)calloc(Rn, sizeof(float));
  W->center = (float *)calloc(Rn * 3, sizeof(float));
  W->retEnergy = (double *)calloc(Rn, sizeof(double));
  W->Hdd_accAF = (float *)calloc(Pn, sizeof(float));
  W->bd_accAF = (float *)calloc(Pn, sizeof(float));
  W->Hcd_accAF = (float *)calloc(Pn * 4, sizeof(float));
  W->Hdd_accLF = (float *)calloc(Pn, sizeof(float));
  W->bd_accLF = (float *)calloc(Pn, sizeof(float));
  W->Hcd_accLF = (float *)calloc(Pn * 4, sizeof(float));
  W->HdiF = (float *)calloc(Pn, sizeof(float));
  W->bdSumF = (float *)calloc(Pn, sizeof(float));
  W->idepth_hessian = (float *)calloc(Pn, sizeof(float));
  W->step = (float *)calloc(Pn, sizeof(float));
  W->maxRelBaseline = (float *)calloc(Pn, sizeof(float));
  W->numGoodResiduals = (int32_t *)calloc(Pn, sizeof(int32_t));
  W->idepth_backup = (float *)calloc(Pn, sizeof(float));
  W->precalc = (sos_precalc *)calloc(nn, sizeof(sos_precalc));
  W->adHTdeltaF = (float *)calloc(nn * 8, sizeof(float));
  W->adHost = (double *)calloc(nn * 64, sizeof(double));
  W->adTarget = (double *)calloc(nn * 64, sizeof(double));
  W->adHostF = (float *)calloc(nn * 64, sizeof(float));
  W->adTargetF = (float *)calloc(nn * 64, sizeof(float));
  int dim = 4 + 8 * n;
  W->HM = (double *)calloc((size_t)dim * dim, sizeof(double));
  W->bM = (double *)calloc((size_t)dim, sizeof(double));
  W->lastX = (double *)calloc((size_t)dim, sizeof(double));
  for (int i = 0; i < n; i++) W->frameEnergyTH[i] = 8 * 8 * 8;
  return W;
}

void orc_window_destroy(orc_window *W) {
  if (!W) return;
  free(W->pts); free(W->res); free(W->pt_begin); free(W->J); free(W->Jn); free(W->res_toZeroF);
  free(W->JpJdF); free(W->newState); free(W->newEnergy); free(W->newEnergyWO); free(W->center);
  free(W->retEnergy); free(W->Hdd_accAF); free(W->bd_accAF); free(W->Hcd_accAF); free(W->Hdd_accLF);
  free(W->bd_accLF); free(W->Hcd_accLF); free(W->HdiF); free(W->bdSumF); free(W->idepth_hessian);
  free(W->step); free(W->maxRelBaseline); free(W->numGoodResiduals); free(W->idepth_backup);
  free(W->precalc); free(W->adHTdeltaF); free(W->adHost); free(W->adTarget); free(W->adHostF);
  free(W->adTargetF); free(W->HM); free(W->bM); free(W->lastX);
  free(W);
}

void orc_set_image(orc_window *W, int frame, const float *dI) { W->img[frame] = dI; }

void orc_set_lin(orc_window *W, const float *res_toZeroF, const sos_rawjac *linJ) {
  if (res_toZeroF) memcpy(W->res_toZeroF, res_toZeroF, sizeof(float) * 8 * (size_t)W->R);
  if (linJ)
    for (int r = 0; r < W->R; r++)
      if (W->res[r].flags & SOS_RF_LINEARIZED) W->J[r] = linJ[r];
}

void orc_set_state(orc_window *W, const sos_calib *calib, const sos_precalc *precalc,
                   const float *adHTdeltaF, const float *cDeltaF, const double *adHost,
                   const double *adTarget, const float *idepth_scaled,
                   const float *idepth_zero_scaled, const float *deltaF) {
  size_t nn = (size_t)W->n * W->n;
  if (calib) W->calib = *calib;
  if (precalc) memcpy(W->precalc, precalc, sizeof(sos_precalc) * nn);
  if (adHTdeltaF) memcpy(W->adHTdeltaF, adHTdeltaF, sizeof(float) * 8 * nn);
  if (cDeltaF) memcpy(W->cDeltaF, cDeltaF, sizeof(float) * 4);
  if (adHost) {
    memcpy(W->adHost, adHost, sizeof(double) * 64 * nn);
    for (size_t i = 0; i < 64 * nn; i++) W->adHostF[i] = (float)adHost[i]; /* OB/EnergyFunctional.cpp:94-98 */
  }
  if (adTarget) {
    memcpy(W->adTarget, adTarget, sizeof(double) * 64 * nn);
    for (size_t i = 0; i < 64 * nn; i++) W->adTargetF[i] = (float)adTarget[i];
  }
  for (int p = 0; p < W->P; p++) {
    if (idepth_scaled) W->pts[p].idepth_scaled = idepth_scaled[p];
    if (idepth_zero_scaled) W->pts[p].idepth_zero_scaled = idepth_zero_scaled[p];
    if (deltaF) W->pts[p].deltaF = deltaF[p];
  }
}

sos_rawjac *orc_J(orc_window *W) { return W->J; }
sos_rawjac *orc_Jnew(orc_window *W) { return W->Jn; }
sos_resid *orc_res(orc_window *W) { return W->res; }
sos_point *orc_pts(orc_window *W) { return W->pts; }
int32_t *orc_new_state(orc_window *W) { return W->newState; }
float *orc_new_energy(orc_window *W) { return W->newEnergy; }
float *orc_new_energy_wo(orc_window *W) { return W->newEnergyWO; }
float *orc_center(orc_window *W) { return W->center; }
float *orc_JpJdF(orc_window *W) { return W->JpJdF; }
float *orc_res_toZeroF(orc_window *W) { return W->res_toZeroF; }
float *orc_point_field(orc_window *W, int which) {
  switch (which) {
    case 0: return W->idepth_hessian;
    case 1: return W->HdiF;
    case 2: return W->bdSumF;
    case 3: return W->Hdd_accAF;
    case 4: return W->bd_accAF;
    case 5: return W->Hcd_accAF;
    case 6: return W->Hdd_accLF;
    case 7: return W->bd_accLF;
    case 8: return W->Hcd_accLF;
    case 9: return W->step;
    case 10: return W->maxRelBaseline;
    default: return 0;
  }
}

int32_t *orc_num_good_residuals(orc_window *W) { return W->numGoodResiduals; }

/* ================================================================================================
 * linearize -- FS/Residuals.cpp:77-271
 * ============================================================================================== */
/* util/globalFuncs.h:68-82, one channel-interleaved AoS (I,dx,dy) tap */
static inline void interp33(const float *mat, float x, float y, int width, float *out) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float *bp = mat + 3 * (ix + iy * width);
  float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; c++)
    out[c] = w11 * bp[3 * (1 + width) + c] + w01 * bp[3 * width + c] + w10 * bp[3 + c] + w00 * bp[c];
}

static double linearize_one(orc_window *W, int r, const float *frameEnergyTH) {
  sos_resid *res = &W->res[r];
  const sos_point *pt = &W->pts[res->point];
  sos_rawjac *J = &W->Jn[r];
  const sos_params *prm = &W->prm;
  const sos_calib *C = &W->calib;
  W->newEnergyWO[r] = -1; /* :78 */

  if (res->state_state == SOS_RES_OOB) { /* :80-83 */
    W->newState[r] = SOS_RES_OOB;
    return (double)res->state_energy;
  }
  const sos_precalc *pc = &W->precalc[res->host + W->n * res->target];
  const float *dIl = W->img[res->target];
  const float *KRKi = pc->PRE_KRKiTll, *Kt = pc->PRE_KtTll, *R0 = pc->PRE_RTll_0, *t0 = pc->PRE_tTll_0;
  const float wM3G = (float)(prm->w - 3), hM3G = (float)(prm->h - 3);
  float energyLeft = 0;
  float affLL0 = pc->PRE_aff_mode[0], affLL1 = pc->PRE_aff_mode[1];
  float b0 = pc->PRE_b0_mode;

  float d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x, d_d_y;
  {
    /* projectPoint, FS/ResidualProjections.h:52-73 with dx = dy = 0 */
    float KliP0 = (pt->u + 0 - C->cxl) * C->fxli;
    float KliP1 = (pt->v + 0 - C->cyl) * C->fyli;
    float idz = pt->idepth_zero_scaled;
    float ptp0 = R0[0] * KliP0 + R0[1] * KliP1 + R0[2] + t0[0] * idz;
    float ptp1 = R0[3] * KliP0 + R0[4] * KliP1 + R0[5] + t0[1] * idz;
    float ptp2 = R0[6] * KliP0 + R0[7] * KliP1 + R0[8] + t0[2] * idz;
    float drescale = 1.0f / ptp2;
    float new_idepth = idz * drescale;
    int ok = 1;
    float u = 0, v = 0, Ku = 0, Kv = 0;
    if (!(drescale > 0)) ok = 0;
    if (ok) {
      u = ptp0 * drescale;
      v = ptp1 * drescale;
      Ku = u * C->fxl + C->cxl;
      Kv = v * C->fyl + C->cyl;
      ok = Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
    }
    if (!ok) { /* :107-112 */
      W->newState[r] = SOS_RES_OOB;
      return (double)res->state_energy;
    }
    W->center[3 * r + 0] = Ku; W->center[3 * r + 1] = Kv; W->center[3 * r + 2] = new_idepth; /* :114 */

    d_d_x = drescale * (t0[0] - t0[2] * u) * SOS_SCALE_IDEPTH * C->fxl; /* :117-120 */
    d_d_y = drescale * (t0[1] - t0[2] * v) * SOS_SCALE_IDEPTH * C->fyl;

    d_C_x[2] = drescale * (R0[6] * u - R0[0]); /* :123-133 */
    d_C_x[3] = C->fxl * drescale * (R0[7] * u - R0[1]) * C->fyli;
    d_C_x[0] = KliP0 * d_C_x[2];
    d_C_x[1] = KliP1 * d_C_x[3];
    d_C_y[2] = C->fyl * drescale * (R0[6] * v - R0[3]) * C->fxli;
    d_C_y[3] = drescale * (R0[7] * v - R0[4]);
    d_C_y[0] = KliP0 * d_C_y[2];
    d_C_y[1] = KliP1 * d_C_y[3];

    d_C_x[0] = (d_C_x[0] + u) * SOS_SCALE_F; /* :135-143 */
    d_C_x[1] *= SOS_SCALE_F;
    d_C_x[2] = (d_C_x[2] + 1) * SOS_SCALE_C;
    d_C_x[3] *= SOS_SCALE_C;
    d_C_y[0] *= SOS_SCALE_F;
    d_C_y[1] = (d_C_y[1] + v) * SOS_SCALE_F;
    d_C_y[2] *= SOS_SCALE_C;
    d_C_y[3] = (d_C_y[3] + 1) * SOS_SCALE_C;

    d_xi_x[0] = new_idepth * C->fxl; /* :145-157 */
    d_xi_x[1] = 0;
    d_xi_x[2] = -new_idepth * u * C->fxl;
    d_xi_x[3] = -u * v * C->fxl;
    d_xi_x[4] = (1 + u * u) * C->fxl;
    d_xi_x[5] = -v * C->fxl;
    d_xi_y[0] = 0;
    d_xi_y[1] = new_idepth * C->fyl;
    d_xi_y[2] = -new_idepth * v * C->fyl;
    d_xi_y[3] = -(1 + v * v) * C->fyl;
    d_xi_y[4] = u * v * C->fyl;
    d_xi_y[5] = u * C->fyl;
  }
  for (int i = 0; i < 6; i++) { J->Jpdxi[0][i] = d_xi_x[i]; J->Jpdxi[1][i] = d_xi_y[i]; } /* :160-169 */
  for (int i = 0; i < 4; i++) { J->Jpdc[0][i] = d_C_x[i]; J->Jpdc[1][i] = d_C_y[i]; }
  J->Jpdd[0] = d_d_x; J->Jpdd[1] = d_d_y;

  float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
  float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
  float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
  float wJI2_sum = 0;

  for (int idx = 0; idx < 8; idx++) { /* :177-243 */
    float u_pt = pt->u + orc_pattern[idx][0], v_pt = pt->v + orc_pattern[idx][1];
    float id = pt->idepth_scaled;
    /* projectPoint, FS/ResidualProjections.h:43-50 */
    float ptp0 = KRKi[0] * u_pt + KRKi[1] * v_pt + KRKi[2] + Kt[0] * id;
    float ptp1 = KRKi[3] * u_pt + KRKi[4] * v_pt + KRKi[5] + Kt[1] * id;
    float ptp2 = KRKi[6] * u_pt + KRKi[7] * v_pt + KRKi[8] + Kt[2] * id;
    float Ku = ptp0 / ptp2;
    float Kv = ptp1 / ptp2;
    if (!(Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G)) {
      W->newState[r] = SOS_RES_OOB;
      return (double)res->state_energy;
    }
    float hit[3];
    interp33(dIl, Ku, Kv, prm->w, hit);
    float residual = hit[0] - (float)(affLL0 * pt->color[idx] + affLL1);
    float drdA = (pt->color[idx] - b0);
    if (!isfinite(hit[0])) {
      W->newState[r] = SOS_RES_OOB;
      return (double)res->state_energy;
    }
    float w = sqrtf(prm->outlierTHSumComponent /
                    (prm->outlierTHSumComponent + (hit[1] * hit[1] + hit[2] * hit[2])));
    w = 0.5f * (w + pt->weights[idx]);
    float hw = fabsf(residual) < prm->huberTH ? 1 : prm->huberTH / fabsf(residual);
    energyLeft += w * w * hw * residual * residual * (2 - hw);
    {
      if (hw < 1) hw = sqrtf(hw);
      hw = hw * w;
      hit[1] *= hw;
      hit[2] *= hw;
      J->resF[idx] = residual * hw;
      J->JIdx[0][idx] = hit[1];
      J->JIdx[1][idx] = hit[2];
      J->JabF[0][idx] = drdA * hw;
      J->JabF[1][idx] = hw;

      JIdxJIdx_00 += hit[1] * hit[1];
      JIdxJIdx_11 += hit[2] * hit[2];
      JIdxJIdx_10 += hit[1] * hit[2];

      JabJIdx_00 += drdA * hw * hit[1];
      JabJIdx_01 += drdA * hw * hit[2];
      JabJIdx_10 += hw * hit[1];
      JabJIdx_11 += hw * hit[2];

      JabJab_00 += drdA * drdA * hw * hw;
      JabJab_01 += drdA * hw * hw;
      JabJab_11 += hw * hw;

      wJI2_sum += hw * hw * (hit[1] * hit[1] + hit[2] * hit[2]);

      if (prm->affineOptModeA < 0) J->JabF[0][idx] = 0;
      if (prm->affineOptModeB < 0) J->JabF[1][idx] = 0;
    }
  }
  J->JIdx2[0] = JIdxJIdx_00; J->JIdx2[1] = JIdxJIdx_10; J->JIdx2[2] = JIdxJIdx_10; J->JIdx2[3] = JIdxJIdx_11;
  J->JabJIdx[0] = JabJIdx_00; J->JabJIdx[1] = JabJIdx_01; J->JabJIdx[2] = JabJIdx_10; J->JabJIdx[3] = JabJIdx_11;
  J->Jab2[0] = JabJab_00; J->Jab2[1] = JabJab_01; J->Jab2[2] = JabJab_01; J->Jab2[3] = JabJab_11;

  W->newEnergyWO[r] = energyLeft; /* :258 */
  float th = fmaxf(frameEnergyTH[res->host], frameEnergyTH[res->target]);
  if (energyLeft > th || wJI2_sum < 2) { /* :260-267 */
    energyLeft = th;
    W->newState[r] = SOS_RES_OUTLIER;
  } else {
    W->newState[r] = SOS_RES_IN;
  }
  W->newEnergy[r] = energyLeft;
  return (double)energyLeft;
}

typedef struct lin_ctx {
  orc_window *W;
  const float *th;
} lin_ctx;
static void lin_job(void *c, int first, int last, int tid) {
  (void)tid;
  lin_ctx *L = (lin_ctx *)c;
  for (int r = first; r < last; r++) {
    if (L->W->res[r].flags & SOS_RF_LINEARIZED) { /* not in activeResiduals, FS/FullSystemOptimize.cpp:321 */
      L->W->retEnergy[r] = 0;
      L->W->newState[r] = L->W->res[r].state_state;
      L->W->newEnergyWO[r] = -1;
      continue;
    }
    L->W->retEnergy[r] = linearize_one(L->W, r, L->th);
  }
}

double orc_linearize_all(orc_window *W, const float *frameEnergyTH, int nthreads) {
  lin_ctx L = {W, frameEnergyTH};
  /* static split into nthreads chunks: treadReduce.reduce(..., 0, size, 0) */
  orc_parallel_for(nthreads, lin_job, &L, 0, W->R, 0);
  double E = 0;
  for (int r = 0; r < W->R; r++) E += W->retEnergy[r];
  return E;
}

/* ================================================================================================
 * applyRes(true) / takeDataF / resetOOB / fixLinearizationF
 * ============================================================================================== */
static void take_data(orc_window *W, int r) { /* OB/EnergyFunctionalStructs.cpp:36-45 */
  sos_rawjac tmp = W->J[r]; /* std::swap(J, data->J) */
  W->J[r] = W->Jn[r];
  W->Jn[r] = tmp;
  const sos_rawjac *J = &W->J[r];
  float v0 = J->JIdx2[0] * J->Jpdd[0] + J->JIdx2[1] * J->Jpdd[1];
  float v1 = J->JIdx2[2] * J->Jpdd[0] + J->JIdx2[3] * J->Jpdd[1];
  float *o = &W->JpJdF[8 * (size_t)r];
  for (int i = 0; i < 6; i++) o[i] = J->Jpdxi[0][i] * v0 + J->Jpdxi[1][i] * v1;
  o[6] = J->JabJIdx[0] * J->Jpdd[0] + J->JabJIdx[1] * J->Jpdd[1];
  o[7] = J->JabJIdx[2] * J->Jpdd[0] + J->JabJIdx[3] * J->Jpdd[1];
}

void orc_apply_res_one(orc_window *W, int r) { /* FS/Residuals.cpp:304-321 */
  sos_resid *res = &W->res[r];
  if (res->state_state == SOS_RES_OOB) return; /* can never go back from OOB */
  if (W->newState[r] == SOS_RES_IN) {
    res->flags |= SOS_RF_ACTIVE;
    take_data(W, r);
  } else {
    res->flags &= ~SOS_RF_ACTIVE;
  }
  res->state_state = W->newState[r];
  res->state_energy = W->newEnergy[r];
}

/* single-residual pieces of FullSystem::flagPointsForRemoval (FS/FullSystem.cpp:577-584) */
double orc_linearize_one(orc_window *W, int r, const float *frameEnergyTH) { return linearize_one(W, r, frameEnergyTH); }
void orc_reset_oob_one(orc_window *W, int r) { /* FS/Residuals.h:83-88 */
  W->newEnergy[r] = 0;
  W->res[r].state_energy = 0;
  W->newState[r] = SOS_RES_OUTLIER;
  W->res[r].state_state = SOS_RES_IN;
}

void orc_apply_res(orc_window *W) {
  for (int r = 0; r < W->R; r++)
    if (!(W->res[r].flags & SOS_RF_LINEARIZED)) orc_apply_res_one(W, r);
}

void orc_reset_oob(orc_window *W) { /* FS/Residuals.h:83-88 over activeResiduals */
  for (int r = 0; r < W->R; r++)
    if (!(W->res[r].flags & SOS_RF_LINEARIZED)) {
      W->newEnergy[r] = 0;
      W->res[r].state_energy = 0;
      W->newState[r] = SOS_RES_OUTLIER;
      W->res[r].state_state = SOS_RES_IN;
    }
}

void orc_fix_linearization(orc_window *W, const int32_t *idx, int count) {
  for (int k = 0; k < count; k++) { /* OB/EnergyFunctionalStructs.cpp:75-103 */
    int r = idx[k];
    const sos_resid *res = &W->res[r];
    const sos_rawjac *J = &W->J[r];
    const float *dp = &W->adHTdeltaF[8 * (size_t)(res->host + W->n * res->target)];
    float deltaF = W->pts[res->point].deltaF;
    float dx = 0, dy = 0, dcx = 0, dcy = 0;
    for (int i = 0; i < 6; i++) { dx += J->Jpdxi[0][i] * dp[i]; dy += J->Jpdxi[1][i] * dp[i]; }
    for (int i = 0; i < 4; i++) { dcx += J->Jpdc[0][i] * W->cDeltaF[i]; dcy += J->Jpdc[1][i] * W->cDeltaF[i]; }
    float Jp_delta_x = dx + dcx + J->Jpdd[0] * deltaF;
    float Jp_delta_y = dy + dcy + J->Jpdd[1] * deltaF;
    for (int i = 0; i < 8; i++) {
      float rtz = J->resF[i];
      rtz = rtz - J->JIdx[0][i] * Jp_delta_x;
      rtz = rtz - J->JIdx[1][i] * Jp_delta_y;
      rtz = rtz - J->JabF[0][i] * dp[6];
      rtz = rtz - J->JabF[1][i] * dp[7];
      W->res_toZeroF[8 * (size_t)r + i] = rtz;
    }
    W->res[r].flags |= SOS_RF_LINEARIZED;
  }
}

/* ================================================================================================
 * accumulators -- OB/MatrixAccumulators.h
 * ============================================================================================== */
typedef struct acc_approx { /* :744-1170 */
  float Data[60], Data1k[60], Data1m[60];
  float TR[32], TR1k[32], TR1m[32];
  float BR[8], BR1k[8], BR1m[8];
  double D64[91]; /* fp64 truth accumulation (not in the reference) */
  float numIn1, numIn1k, numIn1m;
  size_t num;
  float H[13 * 13];
} acc_approx;

static void approx_init(acc_approx *a) { memset(a, 0, sizeof(*a)); }
static void approx_shift(acc_approx *a, int force) { /* :1129-1169 */
  if (a->numIn1 > 1000 || force) {
    for (int i = 0; i < 60; i++) a->Data1k[i] = a->Data[i] + a->Data1k[i];
    for (int i = 0; i < 32; i++) a->TR1k[i] = a->TR[i] + a->TR1k[i];
    for (int i = 0; i < 8; i++) a->BR1k[i] = a->BR[i] + a->BR1k[i];
    a->numIn1k += a->numIn1;
    a->numIn1 = 0;
    memset(a->Data, 0, sizeof(a->Data)); memset(a->TR, 0, sizeof(a->TR)); memset(a->BR, 0, sizeof(a->BR));
  }
  if (a->numIn1k > 1000 || force) {
    for (int i = 0; i < 60; i++) a->Data1m[i] = a->Data1k[i] + a->Data1m[i];
    for (int i = 0; i < 32; i++) a->TR1m[i] = a->TR1k[i] + a->TR1m[i];
    for (int i = 0; i < 8; i++) a->BR1m[i] = a->BR1k[i] + a->BR1m[i];
    a->numIn1m += a->numIn1k;
    a->numIn1k = 0;
    memset(a->Data1k, 0, sizeof(a->Data1k)); memset(a->TR1k, 0, sizeof(a->TR1k)); memset(a->BR1k, 0, sizeof(a->BR1k));
  }
}
/* update :928-1055: x = [x4 x6], y = [y4 y6] */
static void approx_update(acc_approx *A, const float *x4, const float *x6, const float *y4,
                          const float *y6, float a, float b, float c, int f64) {
  float x[10], y[10];
  for (int i = 0; i < 4; i++) { x[i] = x4[i]; y[i] = y4[i]; }
  for (int i = 0; i < 6; i++) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
  int idx = 0;
  for (int r = 0; r < 10; r++)
    for (int cc = r; cc < 10; cc++) {
      if (f64)
        A->D64[idx] += (double)a * x[cc] * x[r] + (double)c * y[cc] * y[r] +
                       (double)b * ((double)x[cc] * y[r] + (double)y[cc] * x[r]);
      else
        A->Data[idx] += a * x[cc] * x[r] + c * y[cc] * y[r] + b * (x[cc] * y[r] + y[cc] * x[r]);
      idx++;
    }
  A->num++;
  A->numIn1++;
  approx_shift(A, 0);
}
static void approx_update_tr(acc_approx *A, const float *x4, const float *x6, const float *y4,
                             const float *y6, float TR00, float TR10, float TR01, float TR11,
                             float TR02, float TR12, int f64) { /* :1057-1101 */
  for (int i = 0; i < 10; i++) {
    float xi = i < 4 ? x4[i] : x6[i - 4], yi = i < 4 ? y4[i] : y6[i - 4];
    if (f64) {
      A->D64[55 + 3 * i + 0] += (double)xi * TR00 + (double)yi * TR10;
      A->D64[55 + 3 * i + 1] += (double)xi * TR01 + (double)yi * TR11;
      A->D64[55 + 3 * i + 2] += (double)xi * TR02 + (double)yi * TR12;
    } else {
      A->TR[3 * i + 0] += xi * TR00 + yi * TR10;
      A->TR[3 * i + 1] += xi * TR01 + yi * TR11;
      A->TR[3 * i + 2] += xi * TR02 + yi * TR12;
    }
  }
}
static void approx_update_br(acc_approx *A, float a00, float a01, float a02, float a11, float a12,
                             float a22, int f64) { /* :1103-1112 */
  if (f64) {
    A->D64[85] += a00; A->D64[86] += a01; A->D64[87] += a02; A->D64[88] += a11; A->D64[89] += a12; A->D64[90] += a22;
  } else {
    A->BR[0] += a00; A->BR[1] += a01; A->BR[2] += a02; A->BR[3] += a11; A->BR[4] += a12; A->BR[5] += a22;
  }
}
/* finish :766-794 -> Hd (13x13 double, row-major), layout [C(4) xi(6) a b r] */
static void approx_finish(acc_approx *A, int f64, double *Hd) {
  approx_shift(A, 1);
  int idx = 0;
  for (int r = 0; r < 10; r++)
    for (int c = r; c < 10; c++) {
      double v = f64 ? A->D64[idx] : (double)A->Data1m[idx];
      Hd[13 * r + c] = Hd[13 * c + r] = v;
      idx++;
    }
  idx = 0;
  for (int r = 0; r < 10; r++)
    for (int c = 0; c < 3; c++) {
      double v = f64 ? A->D64[55 + idx] : (double)A->TR1m[idx];
      Hd[13 * r + c + 10] = Hd[13 * (c + 10) + r] = v;
      idx++;
    }
  const int br_r[6] = {10, 10, 10, 11, 11, 12}, br_c[6] = {10, 11, 12, 11, 12, 12};
  for (int k = 0; k < 6; k++) {
    double v = f64 ? A->D64[85 + k] : (double)A->BR1m[k];
    Hd[13 * br_r[k] + br_c[k]] = Hd[13 * br_c[k] + br_r[k]] = v;
  }
  A->num = (size_t)(A->numIn1 + A->numIn1k + A->numIn1m);
}

/* AccumulatorXX<8,8> / <8,4> / <4,4> (:33-78) and AccumulatorX<8>/<4> (:152-202), flattened */
typedef struct acc_xx {
  float A[64], A1k[64], A1m[64];
  double A64[64];
  float numIn1, numIn1k, numIn1m;
  size_t num;
} acc_xx;
static void xx_shift(acc_xx *a, int sz, int force) {
  if (a->numIn1 > 1000 || force) {
    for (int i = 0; i < sz; i++) { a->A1k[i] += a->A[i]; a->A[i] = 0; }
    a->numIn1k += a->numIn1;
    a->numIn1 = 0;
  }
  if (a->numIn1k > 1000 || force) {
    for (int i = 0; i < sz; i++) { a->A1m[i] += a->A1k[i]; a->A1k[i] = 0; }
    a->numIn1m += a->numIn1k;
    a->numIn1k = 0;
  }
}
/* A += w * L * R^T, row-major rows x cols */
static void xx_update(acc_xx *a, const float *L, int rows, const float *R, int cols, float w, int f64) {
  for (int i = 0; i < rows; i++) {
    float wl = w * L[i];
    for (int j = 0; j < cols; j++) {
      if (f64) a->A64[i * cols + j] += (double)w * L[i] * R[j];
      else a->A[i * cols + j] += wl * R[j];
    }
  }
  a->numIn1++;
  xx_shift(a, rows * cols, 0);
}
static void x_update(acc_xx *a, const float *L, int rows, float w, int f64) { /* A += w*L */
  for (int i = 0; i < rows; i++) {
    if (f64) a->A64[i] += (double)w * L[i];
    else a->A[i] += w * L[i];
  }
  a->numIn1++;
  xx_shift(a, rows, 0);
}
static void xx_finish(acc_xx *a, int sz) {
  xx_shift(a, sz, 1);
  a->num = (size_t)(a->numIn1 + a->numIn1k + a->numIn1m);
}

/* ================================================================================================
 * AccumulatedTopHessianSSE::addPoint<mode> -- OB/AccumulatedTopHessian.cpp:35-147
 * ============================================================================================== */
typedef struct top_acc {
  acc_approx *acc; /* n*n */
  int nres;
} top_acc;

static void top_add_point(orc_window *W, top_acc *T, int p, int mode, int f64) {
  const float *dc = W->cDeltaF;
  float dd = W->pts[p].deltaF;
  float bd_acc = 0, Hdd_acc = 0, Hcd_acc[4] = {0, 0, 0, 0};
  int n = W->n;
  for (int r = W->pt_begin[p]; r < W->pt_begin[p + 1]; r++) {
    const sos_resid *res = &W->res[r];
    int lin = (res->flags & SOS_RF_LINEARIZED) != 0, act = (res->flags & SOS_RF_ACTIVE) != 0;
    if (res->flags & ORC_RF_REMOVED) continue;
    if (mode == 0 && (lin || !act)) continue;
    if (mode == 1 && (!lin || !act)) continue;
    if (mode == 2 && !act) continue;
    const sos_rawjac *rJ = &W->J[r];
    int htIDX = res->host + res->target * n;
    const float *dp = &W->adHTdeltaF[8 * (size_t)htIDX];
    float resApprox[8];
    if (mode == 0) memcpy(resApprox, rJ->resF, sizeof(resApprox));
    if (mode == 2) memcpy(resApprox, &W->res_toZeroF[8 * (size_t)r], sizeof(resApprox));
    if (mode == 1) { /* :74-98 */
      float dx = 0, dy = 0, dcx = 0, dcy = 0;
      for (int i = 0; i < 6; i++) { dx += rJ->Jpdxi[0][i] * dp[i]; dy += rJ->Jpdxi[1][i] * dp[i]; }
      for (int i = 0; i < 4; i++) { dcx += rJ->Jpdc[0][i] * dc[i]; dcy += rJ->Jpdc[1][i] * dc[i]; }
      float Jp_delta_x = dx + dcx + rJ->Jpdd[0] * dd;
      float Jp_delta_y = dy + dcy + rJ->Jpdd[1] * dd;
      for (int i = 0; i < 8; i++) {
        float rtz = W->res_toZeroF[8 * (size_t)r + i];
        rtz = rtz + rJ->JIdx[0][i] * Jp_delta_x;
        rtz = rtz + rJ->JIdx[1][i] * Jp_delta_y;
        rtz = rtz + rJ->JabF[0][i] * dp[6];
        rtz = rtz + rJ->JabF[1][i] * dp[7];
        resApprox[i] = rtz;
      }
    }
    float JI_r[2] = {0, 0}, Jab_r[2] = {0, 0}, rr = 0; /* :101-110 */
    for (int i = 0; i < 8; i++) {
      JI_r[0] += resApprox[i] * rJ->JIdx[0][i];
      JI_r[1] += resApprox[i] * rJ->JIdx[1][i];
      Jab_r[0] += resApprox[i] * rJ->JabF[0][i];
      Jab_r[1] += resApprox[i] * rJ->JabF[1][i];
      rr += resApprox[i] * resApprox[i];
    }
    acc_approx *A = &T->acc[htIDX];
    approx_update(A, rJ->Jpdc[0], rJ->Jpdxi[0], rJ->Jpdc[1], rJ->Jpdxi[1], rJ->JIdx2[0], rJ->JIdx2[1],
                  rJ->JIdx2[3], f64);
    approx_update_br(A, rJ->Jab2[0], rJ->Jab2[1], Jab_r[0], rJ->Jab2[3], Jab_r[1], rr, f64);
    approx_update_tr(A, rJ->Jpdc[0], rJ->Jpdxi[0], rJ->Jpdc[1], rJ->Jpdxi[1], rJ->JabJIdx[0],
                     rJ->JabJIdx[1], rJ->JabJIdx[2], rJ->JabJIdx[3], JI_r[0], JI_r[1], f64);

    float Ji2_Jpdd0 = rJ->JIdx2[0] * rJ->Jpdd[0] + rJ->JIdx2[1] * rJ->Jpdd[1]; /* :124-127 */
    float Ji2_Jpdd1 = rJ->JIdx2[2] * rJ->Jpdd[0] + rJ->JIdx2[3] * rJ->Jpdd[1];
    bd_acc += JI_r[0] * rJ->Jpdd[0] + JI_r[1] * rJ->Jpdd[1];
    Hdd_acc += Ji2_Jpdd0 * rJ->Jpdd[0] + Ji2_Jpdd1 * rJ->Jpdd[1];
    for (int i = 0; i < 4; i++) Hcd_acc[i] += rJ->Jpdc[0][i] * Ji2_Jpdd0 + rJ->Jpdc[1][i] * Ji2_Jpdd1;
    T->nres++;
  }
  if (mode == 0) {
    W->Hdd_accAF[p] = Hdd_acc; W->bd_accAF[p] = bd_acc;
    memcpy(&W->Hcd_accAF[4 * (size_t)p], Hcd_acc, sizeof(Hcd_acc));
  }
  if (mode == 1 || mode == 2) {
    W->Hdd_accLF[p] = Hdd_acc; W->bd_accLF[p] = bd_acc;
    memcpy(&W->Hcd_accLF[4 * (size_t)p], Hcd_acc, sizeof(Hcd_acc));
  }
  if (mode == 2) {
    memset(&W->Hcd_accAF[4 * (size_t)p], 0, 4 * sizeof(float));
    W->Hdd_accAF[p] = 0; W->bd_accAF[p] = 0;
  }
}

static void mat8_mul(const double *A, const double *B, double *C, int bt) { /* C = A*B or A*B^T, 8x8 */
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      double s = 0;
      for (int k = 0; k < 8; k++) s += A[8 * i + k] * (bt ? B[8 * j + k] : B[8 * k + j]);
      C[8 * i + j] = s;
    }
}

/* stitchDoubleInternal :231-290 for blocks [kmin,kmax), summing `nacc` per-thread accumulators */
static void top_stitch_range(orc_window *W, top_acc *T, int nacc, int f64, int kmin, int kmax,
                             double *H, double *b) {
  int n = W->n, dim = 4 + 8 * n;
  for (int k = kmin; k < kmax; k++) {
    int h = k % n, t = k / n;
    int hIdx = 4 + h * 8, tIdx = 4 + t * 8, aidx = h + n * t;
    double accH[169];
    memset(accH, 0, sizeof(accH));
    for (int tid = 0; tid < nacc; tid++) {
      double Hd[169];
      memset(Hd, 0, sizeof(Hd));
      approx_finish(&T[tid].acc[aidx], f64, Hd);
      if (T[tid].acc[aidx].num == 0) continue;
      for (int i = 0; i < 169; i++) accH[i] += Hd[i];
    }
    double B[64], Bpc[32], bp[8];
    for (int i = 0; i < 8; i++) {
      for (int j = 0; j < 8; j++) B[8 * i + j] = accH[13 * (4 + i) + 4 + j];
      for (int j = 0; j < 4; j++) Bpc[4 * i + j] = accH[13 * (4 + i) + j];
      bp[i] = accH[13 * (4 + i) + 12];
    }
    const double *AH = &W->adHost[64 * (size_t)aidx], *AT = &W->adTarget[64 * (size_t)aidx];
    double AHB[64], ATB[64], P1[64], P2[64], P3[64];
    mat8_mul(AH, B, AHB, 0);
    mat8_mul(AT, B, ATB, 0);
    mat8_mul(AHB, AH, P1, 1);
    mat8_mul(ATB, AT, P2, 1);
    mat8_mul(AHB, AT, P3, 1);
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) {
        H[(size_t)(hIdx + i) * dim + hIdx + j] += P1[8 * i + j];
        H[(size_t)(tIdx + i) * dim + tIdx + j] += P2[8 * i + j];
        H[(size_t)(hIdx + i) * dim + tIdx + j] += P3[8 * i + j];
      }
    for (int i = 0; i < 8; i++) {
      for (int j = 0; j < 4; j++) {
        double s1 = 0, s2 = 0;
        for (int kk = 0; kk < 8; kk++) { s1 += AH[8 * i + kk] * Bpc[4 * kk + j]; s2 += AT[8 * i + kk] * Bpc[4 * kk + j]; }
        H[(size_t)(hIdx + i) * dim + j] += s1;
        H[(size_t)(tIdx + i) * dim + j] += s2;
      }
      double s1 = 0, s2 = 0;
      for (int kk = 0; kk < 8; kk++) { s1 += AH[8 * i + kk] * bp[kk]; s2 += AT[8 * i + kk] * bp[kk]; }
      b[hIdx + i] += s1;
      b[tIdx + i] += s2;
    }
    for (int i = 0; i < 4; i++) {
      for (int j = 0; j < 4; j++) H[(size_t)i * dim + j] += accH[13 * i + j];
      b[i] += accH[13 * i + 12];
    }
  }
}
/* "make diagonal by copying over parts", OB/AccumulatedTopHessian.h:113-126 */
static void top_symmetrize(int n, double *H) {
  int dim = 4 + 8 * n;
  for (int h = 0; h < n; h++) {
    int hIdx = 4 + h * 8;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 8; j++) H[(size_t)i * dim + hIdx + j] = H[(size_t)(hIdx + j) * dim + i];
    for (int t = h + 1; t < n; t++) {
      int tIdx = 4 + t * 8;
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) H[(size_t)(hIdx + i) * dim + tIdx + j] += H[(size_t)(tIdx + j) * dim + hIdx + i];
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) H[(size_t)(tIdx + i) * dim + hIdx + j] = H[(size_t)(hIdx + j) * dim + tIdx + i];
    }
  }
}

/* ================================================================================================
 * AccumulatedSCHessianSSE -- OB/AccumulatedSCHessian.cpp:32-158
 * ============================================================================================== */
typedef struct sc_acc {
  acc_xx *accE;  /* n*n  8x4 */
  acc_xx *accEB; /* n*n  8   */
  acc_xx *accD;  /* n*n*n 8x8 */
  acc_xx accHcc, accbc;
} sc_acc;

static void sc_add_point(orc_window *W, sc_acc *S, int p, int shiftPriorToZero, int f64) {
  int n = W->n, nFrames2 = n * n;
  int ngoodres = 0;
  for (int r = W->pt_begin[p]; r < W->pt_begin[p + 1]; r++)
    if ((W->res[r].flags & SOS_RF_ACTIVE) && !(W->res[r].flags & ORC_RF_REMOVED)) ngoodres++;
  if (ngoodres == 0) {
    W->HdiF[p] = 0; W->bdSumF[p] = 0; W->idepth_hessian[p] = 0; W->maxRelBaseline[p] = 0;
    return;
  }
  float H = W->Hdd_accAF[p] + W->Hdd_accLF[p] + W->pts[p].priorF;
  if (H < 1e-10) H = 1e-10;
  W->idepth_hessian[p] = H;
  W->HdiF[p] = (float)(1.0 / H);
  W->bdSumF[p] = W->bd_accAF[p] + W->bd_accLF[p];
  if (shiftPriorToZero) W->bdSumF[p] += W->pts[p].priorF * W->pts[p].deltaF;
  float Hcd[4];
  for (int i = 0; i < 4; i++) Hcd[i] = W->Hcd_accAF[4 * (size_t)p + i] + W->Hcd_accLF[4 * (size_t)p + i];
  float Hdi = W->HdiF[p], bdSum = W->bdSumF[p];
  xx_update(&S->accHcc, Hcd, 4, Hcd, 4, Hdi, f64);
  x_update(&S->accbc, Hcd, 4, bdSum * Hdi, f64);
  for (int r1 = W->pt_begin[p]; r1 < W->pt_begin[p + 1]; r1++) {
    if (!(W->res[r1].flags & SOS_RF_ACTIVE) || (W->res[r1].flags & ORC_RF_REMOVED)) continue;
    int r1ht = W->res[r1].host + W->res[r1].target * n;
    for (int r2 = W->pt_begin[p]; r2 < W->pt_begin[p + 1]; r2++) {
      if (!(W->res[r2].flags & SOS_RF_ACTIVE) || (W->res[r2].flags & ORC_RF_REMOVED)) continue;
      xx_update(&S->accD[r1ht + W->res[r2].target * nFrames2], &W->JpJdF[8 * (size_t)r1], 8,
                &W->JpJdF[8 * (size_t)r2], 8, Hdi, f64);
    }
    xx_update(&S->accE[r1ht], &W->JpJdF[8 * (size_t)r1], 8, Hcd, 4, Hdi, f64);
    x_update(&S->accEB[r1ht], &W->JpJdF[8 * (size_t)r1], 8, Hdi * bdSum, f64);
  }
}

static void sc_get(acc_xx *a, int sz, int f64, double *out, size_t *num) {
  xx_finish(a, sz);
  for (int i = 0; i < sz; i++) out[i] = f64 ? a->A64[i] : (double)a->A1m[i];
  if (num) *num = a->num;
}

static void sc_stitch_range(orc_window *W, sc_acc *S, int nacc, int f64, int kmin, int kmax, double *H,
                            double *b) {
  int nf = W->n, nframes2 = nf * nf, dim = 4 + 8 * nf;
  for (int kk = kmin; kk < kmax; kk++) {
    int i = kk % nf, j = kk / nf;
    int iIdx = 4 + i * 8, jIdx = 4 + j * 8, ijIdx = i + nf * j;
    double Hpc[32], bp[8];
    memset(Hpc, 0, sizeof(Hpc)); memset(bp, 0, sizeof(bp));
    for (int tid = 0; tid < nacc; tid++) {
      double e[32], eb[8];
      sc_get(&S[tid].accE[ijIdx], 32, f64, e, 0);
      sc_get(&S[tid].accEB[ijIdx], 8, f64, eb, 0);
      for (int q = 0; q < 32; q++) Hpc[q] += e[q];
      for (int q = 0; q < 8; q++) bp[q] += eb[q];
    }
    const double *AH = &W->adHost[64 * (size_t)ijIdx], *AT = &W->adTarget[64 * (size_t)ijIdx];
    for (int r = 0; r < 8; r++) {
      for (int c = 0; c < 4; c++) {
        double s1 = 0, s2 = 0;
        for (int q = 0; q < 8; q++) { s1 += AH[8 * r + q] * Hpc[4 * q + c]; s2 += AT[8 * r + q] * Hpc[4 * q + c]; }
        H[(size_t)(iIdx + r) * dim + c] += s1;
        H[(size_t)(jIdx + r) * dim + c] += s2;
      }
      double s1 = 0, s2 = 0;
      for (int q = 0; q < 8; q++) { s1 += AH[8 * r + q] * bp[q]; s2 += AT[8 * r + q] * bp[q]; }
      b[iIdx + r] += s1;
      b[jIdx + r] += s2;
    }
    for (int k = 0; k < nf; k++) {
      int kIdx = 4 + k * 8, ijkIdx = ijIdx + k * nframes2, ikIdx = i + nf * k;
      double accDM[64];
      memset(accDM, 0, sizeof(accDM));
      int any = 0;
      for (int tid = 0; tid < nacc; tid++) {
        double d[64];
        size_t num;
        sc_get(&S[tid].accD[ijkIdx], 64, f64, d, &num);
        if (num == 0) continue;
        any = 1;
        for (int q = 0; q < 64; q++) accDM[q] += d[q];
      }
      if (!any) continue; /* all-zero block contributes nothing */
      const double *AHk = &W->adHost[64 * (size_t)ikIdx], *ATk = &W->adTarget[64 * (size_t)ikIdx];
      double AHD[64], ATD[64], P[64];
      mat8_mul(AH, accDM, AHD, 0);
      mat8_mul(AT, accDM, ATD, 0);
      mat8_mul(AHD, AHk, P, 1);
      for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) H[(size_t)(iIdx + r) * dim + iIdx + c] += P[8 * r + c];
      mat8_mul(ATD, ATk, P, 1);
      for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) H[(size_t)(jIdx + r) * dim + kIdx + c] += P[8 * r + c];
      mat8_mul(ATD, AHk, P, 1);
      for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) H[(size_t)(jIdx + r) * dim + iIdx + c] += P[8 * r + c];
      mat8_mul(AHD, ATk, P, 1);
      for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) H[(size_t)(iIdx + r) * dim + kIdx + c] += P[8 * r + c];
    }
  }
}
static void sc_stitch_tail(orc_window *W, sc_acc *S, int nacc, int f64, double *H, double *b) {
  int dim = 4 + 8 * W->n;
  for (int tid = 0; tid < nacc; tid++) {
    double hcc[16], bc[4];
    sc_get(&S[tid].accHcc, 16, f64, hcc, 0);
    sc_get(&S[tid].accbc, 4, f64, bc, 0);
    for (int i = 0; i < 4; i++) {
      for (int j = 0; j < 4; j++) H[(size_t)i * dim + j] += hcc[4 * i + j];
      b[i] += bc[i];
    }
  }
  /* copy transposed parts for calibration only, OB/AccumulatedSCHessian.h:118-123 */
  for (int h = 0; h < W->n; h++) {
    int hIdx = 4 + h * 8;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 8; j++) H[(size_t)i * dim + hIdx + j] = H[(size_t)(hIdx + j) * dim + i];
  }
}

/* ================================================================================================
 * accumulateAF_MT / LF_MT / SCF_MT -- OB/EnergyFunctional.cpp:197-254
 * ============================================================================================== */
typedef struct accum_ctx {
  orc_window *W;
  top_acc *T;
  sc_acc *S;
  int mode, f64, nacc;
  double **Hs, **bs;
} accum_ctx;

static void top_job(void *c, int first, int last, int tid) {
  accum_ctx *A = (accum_ctx *)c;
  for (int p = first; p < last; p++) top_add_point(A->W, &A->T[tid], p, A->mode, A->f64);
}
static void sc_job(void *c, int first, int last, int tid) {
  accum_ctx *A = (accum_ctx *)c;
  for (int p = first; p < last; p++) sc_add_point(A->W, &A->S[tid], p, 1, A->f64);
}
static void top_stitch_job(void *c, int first, int last, int tid) {
  accum_ctx *A = (accum_ctx *)c;
  top_stitch_range(A->W, A->T, A->nacc, A->f64, first, last, A->Hs[tid], A->bs[tid]);
}
static void sc_stitch_job(void *c, int first, int last, int tid) {
  accum_ctx *A = (accum_ctx *)c;
  sc_stitch_range(A->W, A->S, A->nacc, A->f64, first, last, A->Hs[tid], A->bs[tid]);
}

/* The reference keeps its per-thread accumulators alive and re-initialises them with setZero on the
 * worker threads (OB/EnergyFunctional.cpp:199-201): cache them per (n, nacc) and zero in parallel. */
static top_acc *g_top_cache = 0;
static sc_acc *g_sc_cache = 0;
static int g_cache_n = 0, g_cache_nacc = 0;
typedef struct zero_ctx { top_acc *T; sc_acc *S; int n; } zero_ctx;
static void zero_job(void *c, int first, int last, int tid) {
  (void)tid;
  zero_ctx *Z = (zero_ctx *)c;
  size_t nn = (size_t)Z->n * Z->n;
  for (int i = first; i < last; i++) {
    if (Z->T) { memset(Z->T[i].acc, 0, nn * sizeof(acc_approx)); Z->T[i].nres = 0; }
    if (Z->S) {
      memset(Z->S[i].accE, 0, nn * sizeof(acc_xx)); memset(Z->S[i].accEB, 0, nn * sizeof(acc_xx));
      memset(Z->S[i].accD, 0, nn * Z->n * sizeof(acc_xx));
      memset(&Z->S[i].accHcc, 0, sizeof(acc_xx)); memset(&Z->S[i].accbc, 0, sizeof(acc_xx));
    }
  }
}
static void cache_ensure(int n, int nacc) {
  if (g_cache_n == n && g_cache_nacc == nacc) return;
  if (g_top_cache) {
    for (int i = 0; i < g_cache_nacc; i++) { free(g_top_cache[i].acc); free(g_sc_cache[i].accE); free(g_sc_cache[i].accEB); free(g_sc_cache[i].accD); }
    free(g_top_cache); free(g_sc_cache);
  }
  g_top_cache = (top_acc *)calloc((size_t)nacc, sizeof(top_acc));
  g_sc_cache = (sc_acc *)calloc((size_t)nacc, sizeof(sc_acc));
  for (int i = 0; i < nacc; i++) {
    g_top_cache[i].acc = (acc_approx *)calloc((size_t)n * n, sizeof(acc_approx));
    g_sc_cache[i].accE = (acc_xx *)calloc((size_t)n * n, sizeof(acc_xx));
    g_sc_cache[i].accEB = (acc_xx *)calloc((size_t)n * n, sizeof(acc_xx));
    g_sc_cache[i].accD = (acc_xx *)calloc((size_t)n * n * n, sizeof(acc_xx));
  }
  g_cache_n = n; g_cache_nacc = nacc;
}
static top_acc *top_alloc(int n, int nacc) {
  cache_ensure(n, nacc);
  zero_ctx Z = {g_top_cache, 0, n};
  orc_parallel_for(nacc, zero_job, &Z, 0, nacc, 1);
  return g_top_cache;
}
static void top_free(top_acc *T, int nacc) { (void)T; (void)nacc; }
static sc_acc *sc_alloc(int n, int nacc) {
  cache_ensure(n, nacc);
  zero_ctx Z = {0, g_sc_cache, n};
  orc_parallel_for(nacc, zero_job, &Z, 0, nacc, 1);
  return g_sc_cache;
}
static void sc_free(sc_acc *S, int nacc) { (void)S; (void)nacc; }

static void run_stitch(accum_ctx *A, orc_job_fn job, int nthreads, double *H, double *b) {
  orc_window *W = A->W;
  int n = W->n, dim = 4 + 8 * n;
  int nt = nthreads > 1 ? nthreads : 1;
  double **Hs = (double **)malloc(sizeof(double *) * nt), **bs = (double **)malloc(sizeof(double *) * nt);
  for (int i = 0; i < nt; i++) {
    Hs[i] = (double *)calloc((size_t)dim * dim, sizeof(double));
    bs[i] = (double *)calloc((size_t)dim, sizeof(double));
  }
  A->Hs = Hs; A->bs = bs;
  orc_parallel_for(nthreads, job, A, 0, n * n, 0);
  memcpy(H, Hs[0], sizeof(double) * (size_t)dim * dim);
  memcpy(b, bs[0], sizeof(double) * (size_t)dim);
  for (int i = 1; i < nt; i++) {
    for (size_t q = 0; q < (size_t)dim * dim; q++) H[q] += Hs[i][q];
    for (int q = 0; q < dim; q++) b[q] += bs[i][q];
  }
  for (int i = 0; i < nt; i++) { free(Hs[i]); free(bs[i]); }
  free(Hs); free(bs);
}

void orc_accumulate(orc_window *W, double *H_A, double *b_A, double *H_L, double *b_L, double *H_sc,
                    double *b_sc, int *resInA, int *resInL, int f64, int nthreads) {
  int n = W->n, dim = 4 + 8 * n;
  int nacc = nthreads > 1 ? nthreads : 1;
  double *Ht = (double *)malloc(sizeof(double) * (size_t)dim * dim), *bt = (double *)malloc(sizeof(double) * dim);
  for (int mode = 0; mode < 2; mode++) {
    top_acc *T = top_alloc(n, nacc);
    accum_ctx A = {W, T, 0, mode, f64, nacc, 0, 0};
    orc_parallel_for(nthreads, top_job, &A, 0, W->P, 50);
    run_stitch(&A, top_stitch_job, nthreads, Ht, bt);
    top_symmetrize(n, Ht);
    int nres = 0;
    for (int i = 0; i < nacc; i++) nres += T[i].nres;
    if (mode == 0) {
      if (H_A) memcpy(H_A, Ht, sizeof(double) * (size_t)dim * dim);
      if (b_A) memcpy(b_A, bt, sizeof(double) * dim);
      if (resInA) *resInA = nres;
    } else {
      if (H_L) memcpy(H_L, Ht, sizeof(double) * (size_t)dim * dim);
      if (b_L) memcpy(b_L, bt, sizeof(double) * dim);
      if (resInL) *resInL = nres;
    }
    top_free(T, nacc);
  }
  {
    sc_acc *S = sc_alloc(n, nacc);
    accum_ctx A = {W, 0, S, 0, f64, nacc, 0, 0};
    orc_parallel_for(nthreads, sc_job, &A, 0, W->P, 50);
    run_stitch(&A, sc_stitch_job, nthreads, Ht, bt);
    sc_stitch_tail(W, S, nacc, f64, Ht, bt);
    if (H_sc) memcpy(H_sc, Ht, sizeof(double) * (size_t)dim * dim);
    if (b_sc) memcpy(b_sc, bt, sizeof(double) * dim);
    sc_free(S, nacc);
  }
  free(Ht); free(bt);
}

/* marginalizePointsF accumulation, OB/EnergyFunctional.cpp:909-921 */
void orc_accumulate_marg(orc_window *W, const int32_t *pointIdx, int count, double *M, double *Mb,
                         double *Msc, double *Mbsc, int *resInM) {
  int n = W->n, dim = 4 + 8 * n;
  top_acc *T = top_alloc(n, 1);
  sc_acc *S = sc_alloc(n, 1);
  for (int k = 0; k < count; k++) {
    top_add_point(W, T, pointIdx[k], 2, 0);
    sc_add_point(W, S, pointIdx[k], 0, 0);
  }
  memset(M, 0, sizeof(double) * (size_t)dim * dim); memset(Mb, 0, sizeof(double) * dim);
  memset(Msc, 0, sizeof(double) * (size_t)dim * dim); memset(Mbsc, 0, sizeof(double) * dim);
  top_stitch_range(W, T, 1, 0, 0, n * n, M, Mb);
  top_symmetrize(n, M);
  sc_stitch_range(W, S, 1, 0, 0, n * n, Msc, Mbsc);
  sc_stitch_tail(W, S, 1, 0, Msc, Mbsc);
  if (resInM) *resInM = T[0].nres;
  top_free(T, 1);
  sc_free(S, 1);
}

/* ================================================================================================
 * resubstituteF_MT -- OB/EnergyFunctional.cpp:496-551
 * ============================================================================================== */
typedef struct resub_ctx {
  orc_window *W;
  const float *xc;
  const float *xAd;
} resub_ctx;
static void resub_job(void *c, int first, int last, int tid) {
  (void)tid;
  resub_ctx *X = (resub_ctx *)c;
  orc_window *W = X->W;
  int n = W->n;
  for (int p = first; p < last; p++) {
    int ngoodres = 0;
    for (int r = W->pt_begin[p]; r < W->pt_begin[p + 1]; r++)
      if ((W->res[r].flags & SOS_RF_ACTIVE) && !(W->res[r].flags & ORC_RF_REMOVED)) ngoodres++;
    if (ngoodres == 0) { W->step[p] = 0; continue; }
    float b = W->bdSumF[p];
    float dot = 0;
    for (int i = 0; i < 4; i++) dot += X->xc[i] * (W->Hcd_accAF[4 * (size_t)p + i] + W->Hcd_accLF[4 * (size_t)p + i]);
    b -= dot;
    for (int r = W->pt_begin[p]; r < W->pt_begin[p + 1]; r++) {
      if (!(W->res[r].flags & SOS_RF_ACTIVE) || (W->res[r].flags & ORC_RF_REMOVED)) continue;
      const float *xa = &X->xAd[8 * (size_t)(W->res[r].host * n + W->res[r].target)];
      const float *jp = &W->JpJdF[8 * (size_t)r];
      float d = 0;
      for (int i = 0; i < 8; i++) d += xa[i] * jp[i];
      b -= d;
    }
    W->step[p] = -b * W->HdiF[p];
  }
}
void orc_resubstitute(orc_window *W, const double *x, float *pointStep, int nthreads) {
  int n = W->n, dim = 4 + 8 * n;
  float *xF = (float *)malloc(sizeof(float) * dim);
  for (int i = 0; i < dim; i++) xF[i] = (float)x[i];
  float *xAd = (float *)malloc(sizeof(float) * 8 * (size_t)n * n);
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) {
      const float *AH = &W->adHostF[64 * (size_t)(h + n * t)], *AT = &W->adTargetF[64 * (size_t)(h + n * t)];
      for (int j = 0; j < 8; j++) {
        float s1 = 0, s2 = 0;
        for (int i = 0; i < 8; i++) { s1 += xF[4 + 8 * h + i] * AH[8 * i + j]; s2 += xF[4 + 8 * t + i] * AT[8 * i + j]; }
        xAd[8 * (size_t)(n * h + t) + j] = s1 + s2;
      }
    }
  resub_ctx X = {W, xF, xAd};
  orc_parallel_for(nthreads, resub_job, &X, 0, W->P, 50);
  if (pointStep) memcpy(pointStep, W->step, sizeof(float) * (size_t)W->P);
  free(xF); free(xAd);
}

/* ================================================================================================
 * calcLEnergyPt -- OB/EnergyFunctional.cpp:563-624 (point/residual part; priors stay on the host)
 * ============================================================================================== */
double orc_calc_lenergy(orc_window *W) {
  /* Accumulator11 (OB/MatrixAccumulators.h:80-150): 4 SSE lanes, 1/1k/1M tiers */
  float D[4] = {0, 0, 0, 0}, D1k[4] = {0, 0, 0, 0}, D1m[4] = {0, 0, 0, 0};
  float numIn1 = 0, numIn1k = 0, numIn1m = 0;
  const float *dc = W->cDeltaF;
  int n = W->n;
  for (int p = 0; p < W->P; p++) {
    float dd = W->pts[p].deltaF;
    for (int r = W->pt_begin[p]; r < W->pt_begin[p + 1]; r++) {
      const sos_resid *res = &W->res[r];
      if (!(res->flags & SOS_RF_LINEARIZED) || !(res->flags & SOS_RF_ACTIVE) || (res->flags & ORC_RF_REMOVED)) continue;
      const float *dp = &W->adHTdeltaF[8 * (size_t)(res->host + n * res->target)];
      const sos_rawjac *rJ = &W->J[r];
      float dx = 0, dy = 0, dcx = 0, dcy = 0;
      for (int i = 0; i < 6; i++) { dx += rJ->Jpdxi[0][i] * dp[i]; dy += rJ->Jpdxi[1][i] * dp[i]; }
      for (int i = 0; i < 4; i++) { dcx += rJ->Jpdc[0][i] * dc[i]; dcy += rJ->Jpdc[1][i] * dc[i]; }
      float Jp_delta_x = dx + dcx + rJ->Jpdd[0] * dd;
      float Jp_delta_y = dy + dcy + rJ->Jpdd[1] * dd;
      for (int i = 0; i < 8; i += 4) {
        for (int l = 0; l < 4; l++) {
          float Jdelta = rJ->JIdx[0][i + l] * Jp_delta_x;
          Jdelta = Jdelta + rJ->JIdx[1][i + l] * Jp_delta_y;
          Jdelta = Jdelta + rJ->JabF[0][i + l] * dp[6];
          Jdelta = Jdelta + rJ->JabF[1][i + l] * dp[7];
          float r0 = W->res_toZeroF[8 * (size_t)r + i + l];
          r0 = r0 + r0;
          r0 = r0 + Jdelta;
          Jdelta = Jdelta * r0;
          D[l] += Jdelta; /* updateSSENoShift */
        }
        numIn1++;
      }
    }
    D[0] += W->pts[p].deltaF * W->pts[p].deltaF * W->pts[p].priorF; /* updateSingle */
    numIn1++;
    if (numIn1 > 1000) { for (int l = 0; l < 4; l++) { D1k[l] = D[l] + D1k[l]; D[l] = 0; } numIn1k += numIn1; numIn1 = 0; }
    if (numIn1k > 1000) { for (int l = 0; l < 4; l++) { D1m[l] = D1k[l] + D1m[l]; D1k[l] = 0; } numIn1m += numIn1k; numIn1k = 0; }
  }
  for (int l = 0; l < 4; l++) { D1k[l] = D[l] + D1k[l]; }
  for (int l = 0; l < 4; l++) { D1m[l] = D1k[l] + D1m[l]; }
  (void)numIn1m;
  float A = D1m[0] + D1m[1] + D1m[2] + D1m[3];
  return (double)A;
}
