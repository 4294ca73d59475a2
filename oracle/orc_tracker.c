/*
 * oracle/orc_tracker.c -- TEST INFRASTRUCTURE (CPU oracle, "parity unpinned", see oracle.h).
 *
 * Sequential restatement of the per-frame tracking path:
 *   FrameHessian::makeImages                 FS/HessianBlocks.cpp:121-176
 *   setGlobalCalib (pyramid depth)           util/globalCalib.cpp:39-52
 *   ScaleOptimizer::makeK                    FS/ScaleOptimizer.cpp:95-118
 *   CoarseTracker::makeCoarseDepthL0         FS/CoarseTracker.cpp:56-230
 *   CoarseTracker::calcResPose/calcGSSSEPose FS/CoarseTracker.cpp:612-764, 554-610
 *   Accumulator9::updateSSE_eighted          OB/MatrixAccumulators.h:1314-1432 (4 SSE lanes, tiers)
 *   CoarseTracker::trackNewestCoarse         FS/CoarseTracker.cpp:366-552
 *   ScaleOptimizer::calcResScale/calcGSSSEScale/optimizeScale  FS/ScaleOptimizer.cpp:273-437,232-271,120-230
 *   ScaleAccumulator                         OB/ScaleAccumulator.h:27-106
 */
#include "orc_internal.h"

#define SETTING_coarseCutoffTH(T) ((T)->prm.coarseCutoffTH)

int orc_pyr_levels(int w, int h) { /* util/globalCalib.cpp:39-47 */
  int wlvl = w, hlvl = h, lv = 1;
  while (wlvl % 2 == 0 && hlvl % 2 == 0 && wlvl * hlvl > 5000 && lv < SOS_PYR_LEVELS) {
    wlvl /= 2; hlvl /= 2; lv++;
  }
  return lv;
}

void orc_make_images(const float *color, int w, int h, const float *B, int levels, float **dIp, float **absg) {
  /* the reference leaves the first/last row of dx,dy,absSquaredGrad uninitialised (new[]); the
   * restatement zero-fills them. */
  for (int lvl = 0; lvl < levels; lvl++) {
    int wl = w >> lvl, hl = h >> lvl;
    memset(dIp[lvl], 0, sizeof(float) * 3 * (size_t)wl * hl);
    memset(absg[lvl], 0, sizeof(float) * (size_t)wl * hl);
  }
  for (int i = 0; i < w * h; i++) dIp[0][3 * i] = color[i];
  for (int lvl = 0; lvl < levels; lvl++) {
    int wl = w >> lvl, hl = h >> lvl;
    float *dI_l = dIp[lvl];
    float *dabs_l = absg[lvl];
    if (lvl > 0) {
      int wlm1 = w >> (lvl - 1);
      const float *dI_lm = dIp[lvl - 1];
      for (int y = 0; y < hl; y++)
        for (int x = 0; x < wl; x++)
          dI_l[3 * (x + y * wl)] = 0.25f * (dI_lm[3 * (2 * x + 2 * y * wlm1)] + dI_lm[3 * (2 * x + 1 + 2 * y * wlm1)] +
                                            dI_lm[3 * (2 * x + 2 * y * wlm1 + wlm1)] +
                                            dI_lm[3 * (2 * x + 1 + 2 * y * wlm1 + wlm1)]);
    }
    for (int idx = wl; idx < wl * (hl - 1); idx++) {
      float dx = 0.5f * (dI_l[3 * (idx + 1)] - dI_l[3 * (idx - 1)]);
      float dy = 0.5f * (dI_l[3 * (idx + wl)] - dI_l[3 * (idx - wl)]);
      if (!isfinite(dx)) dx = 0;
      if (!isfinite(dy)) dy = 0;
      dI_l[3 * idx + 1] = dx;
      dI_l[3 * idx + 2] = dy;
      dabs_l[idx] = dx * dx + dy * dy;
      if (B) { /* setting_gammaWeightsPixelSelect == 1 && HCalib != 0; getBGradOnly FS/HessianBlocks.h:496-503 */
        int c = (int)(dI_l[3 * idx] + 0.5f);
        if (c < 5) c = 5;
        if (c > 250) c = 250;
        float gw = B[c + 1] - B[c];
        dabs_l[idx] *= gw * gw;
      }
    }
  }
}

struct orc_tracker {
  sos_params prm;
  int levels;
  int w[SOS_PYR_LEVELS], h[SOS_PYR_LEVELS];
  float fx[SOS_PYR_LEVELS], fy[SOS_PYR_LEVELS], cx[SOS_PYR_LEVELS], cy[SOS_PYR_LEVELS];
  float Ki[SOS_PYR_LEVELS][9];
  float *idepth[SOS_PYR_LEVELS], *weight_sums[SOS_PYR_LEVELS], *weight_sums_bak[SOS_PYR_LEVELS];
  float *pc_u[SOS_PYR_LEVELS], *pc_v[SOS_PYR_LEVELS], *pc_idepth[SOS_PYR_LEVELS], *pc_color[SOS_PYR_LEVELS];
  int pc_n[SOS_PYR_LEVELS];
  /* warp buffers: idepth/rx1, u/rx2, v/rx3, dx, dy, residual, weight, refColor */
  float *buf[8];
  int buf_n;
  float *const *ref_dI;
  /* loop-closure aligner (PoseEstimator): 3-D points with one colour per level */
  int loop_mode, l_n;
  float *l_xyz, *l_col[SOS_PYR_LEVELS];
  int lastInners[SOS_PYR_LEVELS];
  int truth; /* yardstick only: sums of calcRes / calcGSSSE accumulated in fp64 (not the reference's behaviour) */
};

orc_tracker *orc_tracker_create(const sos_params *prm, int w, int h) {
  orc_tracker *T = (orc_tracker *)calloc(1, sizeof(orc_tracker));
  T->prm = *prm;
  T->levels = orc_pyr_levels(w, h);
  for (int l = 0; l < T->levels; l++) {
    int wl = w >> l, hl = h >> l;
    T->w[l] = wl; T->h[l] = hl;
    size_t n = (size_t)wl * hl;
    T->idepth[l] = (float *)calloc(n, sizeof(float));
    T->weight_sums[l] = (float *)calloc(n, sizeof(float));
    T->weight_sums_bak[l] = (float *)calloc(n, sizeof(float));
    T->pc_u[l] = (float *)calloc(n, sizeof(float));
    T->pc_v[l] = (float *)calloc(n, sizeof(float));
    T->pc_idepth[l] = (float *)calloc(n, sizeof(float));
    T->pc_color[l] = (float *)calloc(n, sizeof(float));
  }
  for (int k = 0; k < 8; k++) T->buf[k] = (float *)calloc((size_t)w * h + 4, sizeof(float));
  return T;
}
void orc_tracker_destroy(orc_tracker *T) {
  if (!T) return;
  for (int l = 0; l < T->levels; l++) {
    free(T->idepth[l]); free(T->weight_sums[l]); free(T->weight_sums_bak[l]);
    free(T->pc_u[l]); free(T->pc_v[l]); free(T->pc_idepth[l]); free(T->pc_color[l]);
  }
  for (int k = 0; k < 8; k++) free(T->buf[k]);
  free(T->l_xyz);
  for (int l = 0; l < SOS_PYR_LEVELS; l++) free(T->l_col[l]);
  free(T);
}

static void make_K(orc_tracker *T, const sos_calib *C) { /* FS/ScaleOptimizer.cpp:95-118 */
  T->fx[0] = C->fxl; T->fy[0] = C->fyl; T->cx[0] = C->cxl; T->cy[0] = C->cyl;
  for (int l = 1; l < T->levels; l++) {
    T->fx[l] = T->fx[l - 1] * 0.5;
    T->fy[l] = T->fy[l - 1] * 0.5;
    T->cx[l] = (T->cx[0] + 0.5) / ((int)1 << l) - 0.5;
    T->cy[l] = (T->cy[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  for (int l = 0; l < T->levels; l++) {
    float *Ki = T->Ki[l];
    memset(Ki, 0, 9 * sizeof(float));
    Ki[0] = 1.0f / T->fx[l]; Ki[2] = -T->cx[l] / T->fx[l];
    Ki[4] = 1.0f / T->fy[l]; Ki[5] = -T->cy[l] / T->fy[l];
    Ki[8] = 1;
  }
}

static void dilate(orc_tracker *T, int lvl, int diag) { /* FS/CoarseTracker.cpp:105-190 */
  int wl = T->w[lvl], wh = T->w[lvl] * T->h[lvl] - T->w[lvl];
  float *ws = T->weight_sums[lvl], *bak = T->weight_sums_bak[lvl], *id = T->idepth[lvl];
  memcpy(bak, ws, sizeof(float) * (size_t)T->w[lvl] * T->h[lvl]);
  int off[4];
  if (diag) { off[0] = 1 + wl; off[1] = -1 - wl; off[2] = wl - 1; off[3] = -wl + 1; }
  else { off[0] = 1; off[1] = -1; off[2] = wl; off[3] = -wl; }
  for (int i = wl; i < wh; i++) {
    if (bak[i] <= 0) {
      float sum = 0, num = 0, numn = 0;
      for (int k = 0; k < 4; k++) {
        /* the reference reads one element outside the map for the first / last cell of the loop (index -1
         * and w*h with the diagonal pattern); both cells are border cells that never reach the template,
         * so the tap is skipped here */
        if (i + off[k] < 0 || i + off[k] >= wl * T->h[lvl]) continue;
        if (bak[i + off[k]] > 0) { sum += id[i + off[k]]; num += bak[i + off[k]]; numn++; }
      }
      if (numn > 0) { id[i] = sum / numn; ws[i] = num / numn; }
    }
  }
}

void orc_tracker_set_ref(orc_tracker *T, const sos_calib *C, float *const *ref_dI, int npts, const float *u,
                         const float *v, const float *idepth, const float *hdi, int32_t *pc_n_out) {
  T->loop_mode = 0;
  make_K(T, C);
  T->ref_dI = ref_dI;
  memset(T->idepth[0], 0, sizeof(float) * (size_t)T->w[0] * T->h[0]);
  memset(T->weight_sums[0], 0, sizeof(float) * (size_t)T->w[0] * T->h[0]);
  for (int i = 0; i < npts; i++) { /* :62-79 */
    int ui = u[i] + 0.5f;
    int vi = v[i] + 0.5f;
    float new_idepth = idepth[i];
    float weight = sqrtf(1e-3 / (hdi[i] + 1e-12));
    T->idepth[0][ui + T->w[0] * vi] += new_idepth * weight;
    T->weight_sums[0][ui + T->w[0] * vi] += weight;
  }
  for (int lvl = 1; lvl < T->levels; lvl++) { /* :81-102 */
    int wl = T->w[lvl], hl = T->h[lvl], wlm1 = T->w[lvl - 1];
    float *il = T->idepth[lvl], *wsl = T->weight_sums[lvl];
    const float *ilm = T->idepth[lvl - 1], *wslm = T->weight_sums[lvl - 1];
    for (int y = 0; y < hl; y++)
      for (int x = 0; x < wl; x++) {
        int bidx = 2 * x + 2 * y * wlm1;
        il[x + y * wl] = ilm[bidx] + ilm[bidx + 1] + ilm[bidx + wlm1] + ilm[bidx + wlm1 + 1];
        wsl[x + y * wl] = wslm[bidx] + wslm[bidx + 1] + wslm[bidx + wlm1] + wslm[bidx + wlm1 + 1];
      }
  }
  for (int lvl = 0; lvl < 2 && lvl < T->levels; lvl++) dilate(T, lvl, 1);
  for (int lvl = 2; lvl < T->levels; lvl++) dilate(T, lvl, 0);
  for (int lvl = 0; lvl < T->levels; lvl++) { /* :193-229 */
    float *ws = T->weight_sums[lvl], *id = T->idepth[lvl];
    const float *dIRefl = ref_dI[lvl];
    int wl = T->w[lvl], hl = T->h[lvl], lpc_n = 0;
    for (int y = 2; y < hl - 2; y++)
      for (int x = 2; x < wl - 2; x++) {
        int i = x + y * wl;
        if (ws[i] > 0) {
          id[i] /= ws[i];
          T->pc_u[lvl][lpc_n] = x;
          T->pc_v[lvl][lpc_n] = y;
          T->pc_idepth[lvl][lpc_n] = id[i];
          T->pc_color[lvl][lpc_n] = dIRefl[3 * i];
          if (!isfinite(T->pc_color[lvl][lpc_n]) || !(id[i] > 0)) {
            id[i] = -1;
            continue;
          }
          lpc_n++;
        } else
          id[i] = -1;
        ws[i] = 1;
      }
    T->pc_n[lvl] = lpc_n;
    if (pc_n_out) pc_n_out[lvl] = lpc_n;
  }
}

void orc_tracker_get_pc(orc_tracker *T, int lvl, float *pu, float *pv, float *pi, float *pcol) {
  size_t n = (size_t)T->pc_n[lvl];
  if (pu) memcpy(pu, T->pc_u[lvl], n * sizeof(float));
  if (pv) memcpy(pv, T->pc_v[lvl], n * sizeof(float));
  if (pi) memcpy(pi, T->pc_idepth[lvl], n * sizeof(float));
  if (pcol) memcpy(pcol, T->pc_color[lvl], n * sizeof(float));
}
void orc_tracker_scale_depth(orc_tracker *T, float scale) { /* FS/CoarseTracker.cpp:244-251 */
  for (int l = 0; l < T->levels; l++)
    for (int p = 0; p < T->pc_n[l]; p++) T->pc_idepth[l][p] /= scale;
}
int orc_tracker_warp_n(orc_tracker *T) { return T->buf_n; }
void orc_tracker_set_truth_mode(orc_tracker *T, int on) { T->truth = on; }

static inline void interp33t(const float *mat, float x, float y, int width, float *out) {
  int ix = (int)x, iy = (int)y;
  float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float *bp = mat + 3 * (ix + iy * width);
  float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; c++)
    out[c] = w11 * bp[3 * (1 + width) + c] + w01 * bp[3 * width + c] + w10 * bp[3 + c] + w00 * bp[c];
}

/* shared body of calcResPose (scaleMode 0) and calcResScale (scaleMode 1) */
static void calc_res(orc_tracker *T, int lvl, const float *dINewl, const float *RKi, const float *t, float aff0,
                     float aff1, const float *K1, float scale, int scaleMode, float cutoffTH, double *rs) {
  float E = 0;
  double Ed = 0, sT = 0, sRT = 0; /* truth-mode (fp64) shadows of E and the flow sums */
  int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
  int wl = T->w[lvl], hl = T->h[lvl];
  float fxl = scaleMode ? K1[0] : T->fx[lvl], fyl = scaleMode ? K1[1] : T->fy[lvl];
  float cxl = scaleMode ? K1[2] : T->cx[lvl], cyl = scaleMode ? K1[3] : T->cy[lvl];
  const float *Ki = T->Ki[lvl];
  float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
  float huber = T->prm.huberTH;
  float maxEnergy = 2 * huber * cutoffTH - huber * huber;
  int nl = T->pc_n[lvl];
  /* scale * rot * Ki: Eigen evaluates (scale*M) coefficient-wise before the product */
  float M[9], KiS[9];
  for (int i = 0; i < 9; i++) { M[i] = scaleMode ? scale * RKi[i] : RKi[i]; KiS[i] = scaleMode ? scale * Ki[i] : Ki[i]; }
  for (int i = 0; i < nl; i++) {
    float id = T->pc_idepth[lvl][i], x = T->pc_u[lvl][i], y = T->pc_v[lvl][i];
    float pt0 = M[0] * x + M[1] * y + M[2] + t[0] * id;
    float pt1 = M[3] * x + M[4] * y + M[5] + t[1] * id;
    float pt2 = M[6] * x + M[7] * y + M[8] + t[2] * id;
    float u = pt0 / pt2, v = pt1 / pt2;
    float Ku = fxl * u + cxl, Kv = fyl * v + cyl;
    float new_idepth = id / pt2;
    float rx0 = 0, rx1 = 0, rx2 = 0;
    if (scaleMode) { /* FS/ScaleOptimizer.cpp:333 */
      rx0 = (RKi[0] * x + RKi[1] * y + RKi[2]) / id;
      rx1 = (RKi[3] * x + RKi[4] * y + RKi[5]) / id;
      rx2 = (RKi[6] * x + RKi[7] * y + RKi[8]) / id;
    }
    if (lvl == 0 && i % 32 == 0) {
      float a0 = KiS[0] * x + KiS[1] * y + KiS[2], a1 = KiS[3] * x + KiS[4] * y + KiS[5], a2 = KiS[6] * x + KiS[7] * y + KiS[8];
      float pT0 = a0 + t[0] * id, pT1 = a1 + t[1] * id, pT2 = a2 + t[2] * id;
      float KuT = fxl * (pT0 / pT2) + cxl, KvT = fyl * (pT1 / pT2) + cyl;
      float qT0 = a0 - t[0] * id, qT1 = a1 - t[1] * id, qT2 = a2 - t[2] * id;
      float KuT2 = fxl * (qT0 / qT2) + cxl, KvT2 = fyl * (qT1 / qT2) + cyl;
      float m0 = M[0] * x + M[1] * y + M[2], m1 = M[3] * x + M[4] * y + M[5], m2 = M[6] * x + M[7] * y + M[8];
      float p30 = m0 - t[0] * id, p31 = m1 - t[1] * id, p32 = m2 - t[2] * id;
      float Ku3 = fxl * (p30 / p32) + cxl, Kv3 = fyl * (p31 / p32) + cyl;
      sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      sT += (double)((KuT - x) * (KuT - x) + (KvT - y) * (KvT - y)) + (double)((KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y));
      sRT += (double)((Ku - x) * (Ku - x) + (Kv - y) * (Kv - y)) + (double)((Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y));
      sumSquaredShiftNum += 2;
    }
    if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;
    float refColor = T->pc_color[lvl][i];
    float hit[3];
    interp33t(dINewl, Ku, Kv, wl, hit);
    if (!isfinite(hit[0])) continue;
    float residual = scaleMode ? hit[0] - refColor : hit[0] - (float)(aff0 * refColor + aff1);
    float hw = fabsf(residual) < huber ? 1 : huber / fabsf(residual); /* std::fabs(float) in the C++ reference */
    if (fabsf(residual) > cutoffTH) {
      E += maxEnergy;
      Ed += (double)maxEnergy;
      numTermsInE++;
      numSaturated++;
    } else {
      E += hw * residual * residual * (2 - hw);
      Ed += (double)(hw * residual * residual * (2 - hw));
      numTermsInE++;
      int k = numTermsInWarped;
      T->buf[0][k] = scaleMode ? rx0 : new_idepth;
      T->buf[1][k] = scaleMode ? rx1 : u;
      T->buf[2][k] = scaleMode ? rx2 : v;
      T->buf[3][k] = hit[1];
      T->buf[4][k] = hit[2];
      T->buf[5][k] = residual;
      T->buf[6][k] = hw;
      T->buf[7][k] = refColor;
      numTermsInWarped++;
    }
  }
  while (numTermsInWarped % 4 != 0) {
    for (int k = 0; k < 8; k++) T->buf[k][numTermsInWarped] = 0;
    numTermsInWarped++;
  }
  T->buf_n = numTermsInWarped;
  rs[0] = E;
  rs[1] = numTermsInE;
  rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
  rs[3] = 0;
  rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
  if (T->truth) {
    rs[0] = Ed;
    rs[2] = sT / ((double)sumSquaredShiftNum + 0.1);
    rs[4] = sRT / ((double)sumSquaredShiftNum + 0.1);
  }
}

/* PoseEstimator::calcRes, src/LoopClosure/PoseEstimator.cpp:128-286 */
static void calc_res_loop(orc_tracker *T, int lvl, const float *dINewl, const float *R, const float *t, float aff0, float aff1,
                          float cutoffTH, double *rs) {
  float E = 0;
  double Ed = 0; /* truth-mode (fp64) shadow of E */
  int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
  int wl = T->w[lvl], hl = T->h[lvl];
  float fxl = T->fx[lvl], fyl = T->fy[lvl], cxl = T->cx[lvl], cyl = T->cy[lvl];
  float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
  float huber = T->prm.huberTH;
  float maxEnergy = 2 * huber * cutoffTH - huber * huber;
  for (int i = 0; i < T->l_n; i++) {
    float x = T->l_xyz[3 * i], y = T->l_xyz[3 * i + 1], z = T->l_xyz[3 * i + 2];
    float u0 = x / z, v0 = y / z;
    float Ku0 = fxl * u0 + cxl, Kv0 = fyl * v0 + cyl;
    float pt0 = R[0] * x + R[1] * y + R[2] * z + t[0];
    float pt1 = R[3] * x + R[4] * y + R[5] * z + t[1];
    float pt2 = R[6] * x + R[7] * y + R[8] * z + t[2];
    float u = pt0 / pt2, v = pt1 / pt2;
    float Ku = fxl * u + cxl, Kv = fyl * v + cyl;
    float new_idepth = 1 / pt2;
    if (lvl == 0 && i % 32 == 0) { /* :190-223 */
      float pT0 = x + t[0], pT1 = y + t[1], pT2 = 1 + t[2];
      float KuT = fxl * (pT0 / pT2) + cxl, KvT = fyl * (pT1 / pT2) + cyl;
      float qT0 = x - t[0], qT1 = y - t[1], qT2 = 1 - t[2];
      float KuT2 = fxl * (qT0 / qT2) + cxl, KvT2 = fyl * (qT1 / qT2) + cyl;
      float p30 = R[0] * x + R[1] * y + R[2] - t[0], p31 = R[3] * x + R[4] * y + R[5] - t[1], p32 = R[6] * x + R[7] * y + R[8] - t[2];
      float Ku3 = fxl * (p30 / p32) + cxl, Kv3 = fyl * (p31 / p32) + cyl;
      sumSquaredShiftT += (KuT - Ku0) * (KuT - Ku0) + (KvT - Kv0) * (KvT - Kv0);
      sumSquaredShiftT += (KuT2 - Ku0) * (KuT2 - Ku0) + (KvT2 - Kv0) * (KvT2 - Kv0);
      sumSquaredShiftRT += (Ku - Ku0) * (Ku - Ku0) + (Kv - Kv0) * (Kv - Kv0);
      sumSquaredShiftRT += (Ku3 - Ku0) * (Ku3 - Ku0) + (Kv3 - Kv0) * (Kv3 - Kv0);
      sumSquaredShiftNum += 2;
    }
    if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;
    float refColor = T->l_col[lvl][i];
    float hit[3];
    interp33t(dINewl, Ku, Kv, wl, hit);
    if (!isfinite(hit[0])) continue;
    float residual = hit[0] - (float)(aff0 * refColor + aff1);
    float hw = fabsf(residual) < huber ? 1 : huber / fabsf(residual);
    if (fabsf(residual) > cutoffTH) {
      E += maxEnergy;
      Ed += (double)maxEnergy;
      numTermsInE++;
      numSaturated++;
    } else {
      E += hw * residual * residual * (2 - hw);
      Ed += (double)(hw * residual * residual * (2 - hw));
      numTermsInE++;
      int k = numTermsInWarped;
      T->buf[0][k] = new_idepth; T->buf[1][k] = u; T->buf[2][k] = v; T->buf[3][k] = hit[1]; T->buf[4][k] = hit[2];
      T->buf[5][k] = residual; T->buf[6][k] = hw; T->buf[7][k] = refColor;
      numTermsInWarped++;
    }
  }
  while (numTermsInWarped % 4 != 0) {
    for (int k = 0; k < 8; k++) T->buf[k][numTermsInWarped] = 0;
    numTermsInWarped++;
  }
  T->buf_n = numTermsInWarped;
  rs[0] = T->truth ? Ed : E;
  rs[1] = numTermsInE;
  rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
  rs[3] = 0;
  rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
}

void orc_tracker_calc_res(orc_tracker *T, int lvl, const float *new_dI, const float *RKi, const float *t,
                          const float *affLL, float cutoffTH, double *rs) {
  if (T->loop_mode) calc_res_loop(T, lvl, new_dI, RKi, t, affLL[0], affLL[1], cutoffTH, rs);
  else calc_res(T, lvl, new_dI, RKi, t, affLL[0], affLL[1], 0, 1.0f, 0, cutoffTH, rs);
}

/* PoseEstimator::makeK + pts = matched_frame->pts_dso (:129-148, 296-297): xyz AoS, colors[l * n + i] */
void orc_tracker_set_points3d(orc_tracker *T, const sos_calib *calib, int n, const float *xyz, const float *colors) {
  make_K(T, calib);
  for (int l = 0; l < T->levels; l++) { /* the estimator projects with R alone: no K^-1 in front */
    for (int q = 0; q < 9; q++) T->Ki[l][q] = (q % 4 == 0) ? 1.0f : 0.0f;
    free(T->l_col[l]);
    T->l_col[l] = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    memcpy(T->l_col[l], colors + (size_t)l * n, sizeof(float) * (size_t)n);
  }
  free(T->l_xyz);
  T->l_xyz = (float *)malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
  memcpy(T->l_xyz, xyz, sizeof(float) * 3 * (size_t)n);
  T->l_n = n;
  T->loop_mode = 1;
}
void orc_tracker_last_inners(orc_tracker *T, int *out) {
  for (int l = 0; l < T->levels; l++) out[l] = T->lastInners[l];
}
void orc_tracker_calc_res_scale(orc_tracker *T, int lvl, const float *stereo_dI, const float *RKi, const float *t,
                                const float *K1, float scale, float cutoffTH, double *rs) {
  calc_res(T, lvl, stereo_dI, RKi, t, 1, 0, K1, scale, 1, cutoffTH, rs);
}

/* 4-lane, 3-tier accumulation of `nvals` weighted products per 4-group (Accumulator9 / ScaleAccumulator) */
typedef struct acc_sse {
  float D[45 * 4], D1k[45 * 4], D1m[45 * 4];
  float numIn1, numIn1k;
  int nvals;
} acc_sse;
static void sse_shift(acc_sse *a, int force) {
  if (a->numIn1 > 1000 || force) {
    for (int i = 0; i < a->nvals * 4; i++) { a->D1k[i] = a->D[i] + a->D1k[i]; a->D[i] = 0; }
    a->numIn1k += a->numIn1;
    a->numIn1 = 0;
  }
  if (a->numIn1k > 1000 || force) {
    for (int i = 0; i < a->nvals * 4; i++) { a->D1m[i] = a->D1k[i] + a->D1m[i]; a->D1k[i] = 0; }
    a->numIn1k = 0;
  }
}

void orc_tracker_calc_gs(orc_tracker *T, int lvl, float a, float b0, double *H_out, double *b_out) {
  acc_sse A;
  memset(&A, 0, sizeof(A));
  A.nvals = 45;
  double Hd[81];
  memset(Hd, 0, sizeof(Hd));
  float fxl = T->fx[lvl], fyl = T->fy[lvl];
  int n = T->buf_n;
  for (int i = 0; i < n; i += 4) {
    for (int l = 0; l < 4; l++) { /* FS/CoarseTracker.cpp:571-591 */
      float dx = T->buf[3][i + l] * fxl, dy = T->buf[4][i + l] * fyl;
      float u = T->buf[1][i + l], v = T->buf[2][i + l], id = T->buf[0][i + l];
      float J[9];
      J[0] = id * dx;
      J[1] = id * dy;
      J[2] = 0 - id * (u * dx + v * dy);
      J[3] = 0 - (u * v * dx + dy * (1 + v * v));
      J[4] = u * v * dy + dx * (1 + u * u);
      J[5] = u * dy - v * dx;
      J[6] = a * (b0 - T->buf[7][i + l]);
      J[7] = -1;
      J[8] = T->buf[5][i + l];
      float w = T->buf[6][i + l];
      int idx = 0;
      for (int r = 0; r < 9; r++) { /* updateSSE_eighted */
        float Jw = J[r] * w;
        for (int c = r; c < 9; c++) { A.D[4 * idx + l] += Jw * J[c]; Hd[9 * r + c] += (double)(Jw * J[c]); idx++; }
      }
    }
    A.numIn1++;
    sse_shift(&A, 0);
  }
  sse_shift(&A, 1);
  float Hf[81];
  int idx = 0;
  for (int r = 0; r < 9; r++)
    for (int c = r; c < 9; c++) {
      float d = A.D1m[4 * idx + 0] + A.D1m[4 * idx + 1] + A.D1m[4 * idx + 2] + A.D1m[4 * idx + 3];
      Hf[9 * r + c] = Hf[9 * c + r] = d;
      idx++;
    }
  double inv = 1.0f / n;
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) H_out[8 * r + c] = (double)Hf[9 * r + c] * inv;
    b_out[r] = (double)Hf[9 * r + 8] * inv;
  }
  if (T->truth) { /* yardstick: the same fp32 products summed in fp64 */
    for (int r = 0; r < 8; r++) {
      for (int c = 0; c < 8; c++) H_out[8 * r + c] = Hd[9 * (r < c ? r : c) + (r < c ? c : r)] * inv;
      b_out[r] = Hd[9 * r + 8] * inv;
    }
  }
  const double sc[8] = {SOS_SCALE_XI_ROT, SOS_SCALE_XI_ROT, SOS_SCALE_XI_ROT, SOS_SCALE_XI_TRANS,
                        SOS_SCALE_XI_TRANS, SOS_SCALE_XI_TRANS, SOS_SCALE_A, SOS_SCALE_B}; /* :598-609 */
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) H_out[8 * r + c] *= sc[r] * sc[c];
    b_out[r] *= sc[r];
  }
}

void orc_tracker_calc_gs_scale(orc_tracker *T, int lvl, const float *t, const float *K1, float s, float *H_out,
                               float *b_out) { /* FS/ScaleOptimizer.cpp:232-271 */
  (void)lvl;
  acc_sse A;
  memset(&A, 0, sizeof(A));
  A.nvals = 3;
  double d00 = 0, d01 = 0;
  int n = T->buf_n;
  float tx = t[0], ty = t[1], tz = t[2];
  for (int i = 0; i < n; i += 4) {
    for (int l = 0; l < 4; l++) {
      float dxfx = T->buf[3][i + l] * K1[0], dyfy = T->buf[4][i + l] * K1[1];
      float rx1 = T->buf[0][i + l], rx2 = T->buf[1][i + l], rx3 = T->buf[2][i + l];
      float deno_sqrt = s * rx3 + tz;
      float deno = 1.0f / (deno_sqrt * deno_sqrt);
      float xno = rx1 * tz - rx3 * tx, yno = rx2 * tz - rx3 * ty;
      float J0 = dxfx * (deno * xno) + dyfy * (deno * yno);
      float J1 = T->buf[5][i + l], w = T->buf[6][i + l];
      float J0w = J0 * w, J1w = J1 * w;
      A.D[0 + l] += J0w * J0;
      A.D[4 + l] += J0w * J1;
      A.D[8 + l] += J1w * J1;
      d00 += (double)(J0w * J0);
      d01 += (double)(J0w * J1);
    }
    A.numIn1++;
    sse_shift(&A, 0);
  }
  sse_shift(&A, 1);
  float h00 = A.D1m[0] + A.D1m[1] + A.D1m[2] + A.D1m[3];
  float h01 = A.D1m[4] + A.D1m[5] + A.D1m[6] + A.D1m[7];
  *H_out = h00 * (1.0f / n);
  *b_out = h01 * (1.0f / n);
  if (T->truth) { *H_out = (float)(d00 * (1.0 / n)); *b_out = (float)(d01 * (1.0 / n)); }
}

/* util/NumType.h:156-168 */
static void aff_from_to(float eF, float eT, double Fa, double Fb, double Ta, double Tb, float *out) {
  if (eF == 0 || eT == 0) eT = eF = 1;
  double a = exp(Ta - Fa) * eT / eF;
  double b = Tb - a * Fb;
  out[0] = (float)a; out[1] = (float)b;
}

int orc_tracker_track(orc_tracker *T, float *const *new_dI, float ref_ab, float new_ab, const double *ref_aff,
                      double *lastToNew12, double *aff2, int coarsestLvl, const double *minResForAbort,
                      double *lastResiduals, double *flow3) { /* FS/CoarseTracker.cpp:366-552 */
  for (int i = 0; i < 5; i++) lastResiduals[i] = NAN;
  flow3[0] = flow3[1] = flow3[2] = 1000;
  int maxIterations[] = {10, 20, 50, 50, 50};
  float lambdaExtrapolationLimit = 0.001;
  orc_se3 cur;
  memcpy(cur.R, lastToNew12, 9 * sizeof(double)); memcpy(cur.t, lastToNew12 + 9, 3 * sizeof(double));
  double aff_a = aff2[0], aff_b = aff2[1];
  int haveRepeated = 0;
  float modeA = T->prm.affineOptModeA, modeB = T->prm.affineOptModeB;
  for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
    double H[64], b[8], resOld[6], resNew[6];
    float levelCutoffRepeat = 1;
    float RKi[9], tf[3], affLL[2];
#define ORC_SETUP(pose, a_, b_)                                                              \
  do {                                                                                       \
    float Rf[9];                                                                             \
    for (int q = 0; q < 9; q++) Rf[q] = (float)(pose).R[q];                                  \
    for (int r_ = 0; r_ < 3; r_++)                                                           \
      for (int c_ = 0; c_ < 3; c_++)                                                         \
        RKi[3 * r_ + c_] = Rf[3 * r_] * T->Ki[lvl][c_] + Rf[3 * r_ + 1] * T->Ki[lvl][3 + c_] + \
                           Rf[3 * r_ + 2] * T->Ki[lvl][6 + c_];                              \
    for (int q = 0; q < 3; q++) tf[q] = (float)(pose).t[q];                                  \
    aff_from_to(ref_ab, new_ab, ref_aff[0], ref_aff[1], (a_), (b_), affLL);                  \
  } while (0)
    ORC_SETUP(cur, aff_a, aff_b);
    orc_tracker_calc_res(T, lvl, new_dI[lvl], RKi, tf, affLL, SETTING_coarseCutoffTH(T) * levelCutoffRepeat, resOld);
    while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
      levelCutoffRepeat *= 2;
      orc_tracker_calc_res(T, lvl, new_dI[lvl], RKi, tf, affLL, SETTING_coarseCutoffTH(T) * levelCutoffRepeat, resOld);
    }
    orc_tracker_calc_gs(T, lvl, affLL[0], (float)ref_aff[1], H, b);
    float lambda = 0.01;
    for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
      double Hl[64], nb[8], inc[8];
      memcpy(Hl, H, sizeof(Hl));
      for (int i = 0; i < 8; i++) { Hl[9 * i] *= (1 + lambda); nb[i] = -b[i]; }
      orc_ldlt_solve(Hl, nb, inc, 8);
      if (modeA < 0 && modeB < 0) {
        double Hs[36], bs[6], xs[6];
        for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) Hs[6 * r + c] = Hl[8 * r + c]; bs[r] = nb[r]; }
        orc_ldlt_solve(Hs, bs, xs, 6);
        for (int r = 0; r < 6; r++) inc[r] = xs[r];
        inc[6] = inc[7] = 0;
      }
      if (!(modeA < 0) && modeB < 0) {
        double Hs[49], bs[7], xs[7];
        for (int r = 0; r < 7; r++) { for (int c = 0; c < 7; c++) Hs[7 * r + c] = Hl[8 * r + c]; bs[r] = nb[r]; }
        orc_ldlt_solve(Hs, bs, xs, 7);
        for (int r = 0; r < 7; r++) inc[r] = xs[r];
        inc[7] = 0;
      }
      if (modeA < 0 && !(modeB < 0)) {
        double HS[64], bS[8], Hs[49], bs[7], xs[7];
        memcpy(HS, Hl, sizeof(HS)); memcpy(bS, b, sizeof(bS));
        for (int r = 0; r < 8; r++) HS[8 * r + 6] = HS[8 * r + 7];
        for (int c = 0; c < 8; c++) HS[8 * 6 + c] = HS[8 * 7 + c];
        bS[6] = bS[7];
        for (int r = 0; r < 7; r++) { for (int c = 0; c < 7; c++) Hs[7 * r + c] = HS[8 * r + c]; bs[r] = -bS[r]; }
        orc_ldlt_solve(Hs, bs, xs, 7);
        memset(inc, 0, sizeof(inc));
        for (int r = 0; r < 6; r++) inc[r] = xs[r];
        inc[6] = 0; inc[7] = xs[6];
      }
      float extrapFac = 1;
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit / lambda));
      for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
      double incS[8];
      memcpy(incS, inc, sizeof(incS));
      for (int i = 0; i < 3; i++) incS[i] *= SOS_SCALE_XI_ROT;
      for (int i = 3; i < 6; i++) incS[i] *= SOS_SCALE_XI_TRANS;
      incS[6] *= SOS_SCALE_A; incS[7] *= SOS_SCALE_B;
      double sum = 0;
      for (int i = 0; i < 8; i++) sum += incS[i];
      if (!isfinite(sum)) memset(incS, 0, sizeof(incS));
      orc_se3 E = orc_se3_exp(incS);
      orc_se3 nw = orc_se3_mul(&E, &cur);
      double na = aff_a + incS[6], nbb = aff_b + incS[7];
      ORC_SETUP(nw, na, nbb);
      orc_tracker_calc_res(T, lvl, new_dI[lvl], RKi, tf, affLL, SETTING_coarseCutoffTH(T) * levelCutoffRepeat, resNew);
      int accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      if (accept) {
        orc_tracker_calc_gs(T, lvl, affLL[0], (float)ref_aff[1], H, b);
        memcpy(resOld, resNew, sizeof(resOld));
        aff_a = na; aff_b = nbb;
        cur = nw;
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      double nrm = 0;
      for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
      if (!(sqrt(nrm) > 1e-3)) break;
    }
    lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
    T->lastInners[lvl] = (int)resOld[1];
    flow3[0] = resOld[2]; flow3[1] = resOld[3]; flow3[2] = resOld[4];
    if (lastResiduals[lvl] > 1.5 * minResForAbort[lvl]) return 0;
    if (levelCutoffRepeat > 1 && !haveRepeated) {
      lvl++;
      haveRepeated = 1;
    }
  }
#undef ORC_SETUP
  memcpy(lastToNew12, cur.R, 9 * sizeof(double)); memcpy(lastToNew12 + 9, cur.t, 3 * sizeof(double));
  aff2[0] = aff_a; aff2[1] = aff_b;
  if ((modeA != 0 && (fabsf((float)aff_a) > 1.2)) || (modeB != 0 && (fabsf((float)aff_b) > 200))) return 0;
  float relAff[2];
  aff_from_to(ref_ab, new_ab, ref_aff[0], ref_aff[1], aff_a, aff_b, relAff);
  if ((modeA == 0 && (fabsf(logf(relAff[0])) > 1.5)) || (modeB == 0 && (fabsf(relAff[1]) > 200))) return 0;
  if (modeA < 0) aff2[0] = 0;
  if (modeB < 0) aff2[1] = 0;
  return 1;
}

float orc_tracker_optimize_scale(orc_tracker *T, float *const *stereo_dI, const double *tfm12, const float *K1_0,
                                 float *scale_io, int coarsestLvl) { /* FS/ScaleOptimizer.cpp:120-230 */
  double last_residuals[5] = {NAN, NAN, NAN, NAN, NAN};
  int maxIterations[] = {10, 20, 50, 50, 50};
  float lambdaExtrapolationLimit = 0.001;
  float scale_current = *scale_io;
  int haveRepeated = 0;
  float fx1[SOS_PYR_LEVELS], fy1[SOS_PYR_LEVELS], cx1[SOS_PYR_LEVELS], cy1[SOS_PYR_LEVELS];
  fx1[0] = K1_0[0]; fy1[0] = K1_0[1]; cx1[0] = K1_0[2]; cy1[0] = K1_0[3]; /* FS/ScaleOptimizer.cpp:66-76 */
  for (int l = 1; l < T->levels; l++) {
    fx1[l] = fx1[l - 1] * 0.5; fy1[l] = fy1[l - 1] * 0.5;
    cx1[l] = (cx1[0] + 0.5) / ((int)1 << l) - 0.5;
    cy1[l] = (cy1[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  float tf[3] = {(float)tfm12[9], (float)tfm12[10], (float)tfm12[11]};
  for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
    float H, b, levelCutoffRepeat = 1;
    double resOld[6], resNew[6];
    float K1[4] = {fx1[lvl], fy1[lvl], cx1[lvl], cy1[lvl]};
    float RKi[9], Rf[9];
    for (int q = 0; q < 9; q++) Rf[q] = (float)tfm12[q];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        RKi[3 * r + c] = Rf[3 * r] * T->Ki[lvl][c] + Rf[3 * r + 1] * T->Ki[lvl][3 + c] + Rf[3 * r + 2] * T->Ki[lvl][6 + c];
    orc_tracker_calc_res_scale(T, lvl, stereo_dI[lvl], RKi, tf, K1, scale_current, SETTING_coarseCutoffTH(T) * levelCutoffRepeat, resOld);
    while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
      levelCutoffRepeat *= 2;
      orc_tracker_calc_res_scale(T, lvl, stereo_dI[lvl], RKi, tf, K1, scale_current, SETTING_coarseCutoffTH(T) * levelCutoffRepeat, resOld);
    }
    /* tx,ty,tz are _mm_set1_ps(double) -> float */
    orc_tracker_calc_gs_scale(T, lvl, tf, K1, scale_current, &H, &b);
    float lambda = 0.01;
    for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
      float Hl = H;
      Hl *= (1 + lambda);
      float inc = -b / Hl;
      float extrapFac = 1;
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit / lambda));
      inc *= extrapFac;
      if (!isfinite(inc) || fabs(inc) > scale_current) inc = 0.0;
      float scale_new = scale_current + inc;
      orc_tracker_calc_res_scale(T, lvl, stereo_dI[lvl], RKi, tf, K1, scale_new, SETTING_coarseCutoffTH(T) * levelCutoffRepeat, resNew);
      int accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      if (accept) {
        orc_tracker_calc_gs_scale(T, lvl, tf, K1, scale_new, &H, &b);
        memcpy(resOld, resNew, sizeof(resOld));
        scale_current = scale_new;
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      if (!(inc > 1e-3)) break;
    }
    last_residuals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
    if (levelCutoffRepeat > 1 && !haveRepeated) {
      lvl++;
      haveRepeated = 1;
    }
  }
  *scale_io = scale_current;
  return (float)last_residuals[0];
}
