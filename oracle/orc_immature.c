/* oracle/orc_immature.c -- TEST INFRASTRUCTURE (CPU oracle, see oracle.h; PARITY UNPINNED).
 *
 * Restatement of the immature-point front of the window:
 *   ImmaturePoint::ImmaturePoint   FS/ImmaturePoint.cpp:30-59   (pattern colours, weights, gradH, energyTH)
 *   ImmaturePoint::traceOn         FS/ImmaturePoint.cpp:70-415  (epipolar search, GN refinement, new idepth interval)
 * with getInterpolatedElement31 / 33 / 33BiLin of U/globalFuncs.h:68-82,122-136,161-182.  fp32, no FMA contraction,
 * sums in source order -- the convention of the rest of the oracle. */
#include <math.h>
#include <string.h>

#include "oracle.h"

static const int PATTERN[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}}; /* U/settings.h:64-75 staticPattern[8] */

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* taps are clamped to the image (the reference reads whatever lies there; the search keeps 4 px of margin) */
static inline const float *texel(const float *dI, int ix, int iy, int w, int h) {
  return dI + 3 * ((size_t)clampi(ix, 0, w - 1) + (size_t)clampi(iy, 0, h - 1) * w);
}
static float interp31(const float *dI, float x, float y, int w, int h) {
  int ix = (int)x, iy = (int)y;
  float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  return dxdy * texel(dI, ix + 1, iy + 1, w, h)[0] + (dy - dxdy) * texel(dI, ix, iy + 1, w, h)[0] +
         (dx - dxdy) * texel(dI, ix + 1, iy, w, h)[0] + (1 - dx - dy + dxdy) * texel(dI, ix, iy, w, h)[0];
}
static void interp33(const float *dI, float x, float y, int w, int h, float *o) {
  int ix = (int)x, iy = (int)y;
  float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float *a = texel(dI, ix, iy, w, h), *b = texel(dI, ix + 1, iy, w, h), *c = texel(dI, ix, iy + 1, w, h),
              *d = texel(dI, ix + 1, iy + 1, w, h);
  for (int k = 0; k < 3; k++) o[k] = dxdy * d[k] + (dy - dxdy) * c[k] + (dx - dxdy) * b[k] + (1 - dx - dy + dxdy) * a[k];
}
static void interp33bilin(const float *dI, float x, float y, int w, int h, float *o) {
  int ix = (int)x, iy = (int)y;
  float tl = texel(dI, ix, iy, w, h)[0], tr = texel(dI, ix + 1, iy, w, h)[0], bl = texel(dI, ix, iy + 1, w, h)[0],
        br = texel(dI, ix + 1, iy + 1, w, h)[0];
  float dx = x - ix, dy = y - iy;
  float topInt = dx * tr + (1 - dx) * tl, botInt = dx * br + (1 - dx) * bl;
  float leftInt = dy * bl + (1 - dy) * tl, rightInt = dy * br + (1 - dy) * tr;
  o[0] = dx * rightInt + (1 - dx) * leftInt;
  o[1] = rightInt - leftInt;
  o[2] = botInt - topInt;
}

void orc_immature_init(const sos_trace_params *P, const float *host_dI, int w, int h, int count, const int32_t *u,
                       const int32_t *v, sos_immature *out) { /* FS/ImmaturePoint.cpp:30-59 */
  for (int i = 0; i < count; i++) {
    sos_immature *p = &out[i];
    memset(p, 0, sizeof(*p));
    p->u = (float)u[i];
    p->v = (float)v[i];
    p->idepth_min = 0;
    p->idepth_max = NAN;
    p->lastTraceStatus = SOS_IPS_UNINITIALIZED;
    float g00 = 0, g01 = 0, g10 = 0, g11 = 0;
    int bad = 0;
    for (int idx = 0; idx < 8; idx++) {
      float ptc[3];
      interp33bilin(host_dI, p->u + PATTERN[idx][0], p->v + PATTERN[idx][1], w, h, ptc);
      p->color[idx] = ptc[0];
      if (!isfinite(p->color[idx])) {
        p->energyTH = NAN;
        bad = 1;
        break;
      }
      g00 += ptc[1] * ptc[1]; g01 += ptc[1] * ptc[2]; g10 += ptc[2] * ptc[1]; g11 += ptc[2] * ptc[2];
      p->weights[idx] = sqrtf(P->outlierTHSumComponent / (P->outlierTHSumComponent + (ptc[1] * ptc[1] + ptc[2] * ptc[2])));
    }
    p->gradH[0] = g00; p->gradH[1] = g01; p->gradH[2] = g10; p->gradH[3] = g11;
    if (bad) continue; /* the reference returns from the constructor: quality stays unset */
    p->energyTH = 8 * P->outlierTH;
    p->energyTH *= P->overallEnergyTHWeight * P->overallEnergyTHWeight;
    p->quality = 10000;
  }
}

static int trace_one(const sos_trace_params *P, const float *dI, int w, int h, sos_immature *p, const float *K, const float *Kt,
                     const float *aff) { /* FS/ImmaturePoint.cpp:70-415 */
#define OOB_RETURN(st)              \
  do {                              \
    p->lastTraceUV[0] = -1;         \
    p->lastTraceUV[1] = -1;         \
    p->lastTracePixelInterval = 0;  \
    return p->lastTraceStatus = (st); \
  } while (0)
  if (p->lastTraceStatus == SOS_IPS_OOB) return p->lastTraceStatus;
  float maxPixSearch = (w + h) * P->maxPixSearch;
  float pr0 = K[0] * p->u + K[1] * p->v + K[2] * 1.0f, pr1 = K[3] * p->u + K[4] * p->v + K[5] * 1.0f,
        pr2 = K[6] * p->u + K[7] * p->v + K[8] * 1.0f;
  float ptpMin0 = pr0 + Kt[0] * p->idepth_min, ptpMin1 = pr1 + Kt[1] * p->idepth_min, ptpMin2 = pr2 + Kt[2] * p->idepth_min;
  float uMin = ptpMin0 / ptpMin2, vMin = ptpMin1 / ptpMin2;
  if (!(uMin > 4 && vMin > 4 && uMin < w - 5 && vMin < h - 5)) OOB_RETURN(SOS_IPS_OOB);
  float dist, uMax, vMax;
  if (isfinite(p->idepth_max)) {
    float q0 = pr0 + Kt[0] * p->idepth_max, q1 = pr1 + Kt[1] * p->idepth_max, q2 = pr2 + Kt[2] * p->idepth_max;
    uMax = q0 / q2;
    vMax = q1 / q2;
    if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) OOB_RETURN(SOS_IPS_OOB);
    dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
    dist = sqrtf(dist);
    if (dist < P->slackInterval) {
      p->lastTraceUV[0] = (uMax + uMin) * 0.5f;
      p->lastTraceUV[1] = (vMax + vMin) * 0.5f;
      p->lastTracePixelInterval = dist;
      return p->lastTraceStatus = SOS_IPS_SKIPPED;
    }
  } else {
    dist = maxPixSearch;
    float q0 = pr0 + Kt[0] * 0.01f, q1 = pr1 + Kt[1] * 0.01f, q2 = pr2 + Kt[2] * 0.01f;
    uMax = q0 / q2;
    vMax = q1 / q2;
    float ddx = uMax - uMin, ddy = vMax - vMin;
    float d = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
    uMax = uMin + dist * ddx * d;
    vMax = vMin + dist * ddy * d;
    if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) OOB_RETURN(SOS_IPS_OOB);
  }
  if (!(p->idepth_min < 0 || (ptpMin2 > 0.75f && ptpMin2 < 1.5f))) OOB_RETURN(SOS_IPS_OOB);

  float dx = P->stepsize * (uMax - uMin), dy = P->stepsize * (vMax - vMin);
  const float *G = p->gradH;
  float a = (dx * G[0] + dy * G[2]) * dx + (dx * G[1] + dy * G[3]) * dy;
  float b = (dy * G[0] + (-dx) * G[2]) * dy + (dy * G[1] + (-dx) * G[3]) * (-dx);
  float errorInPixel = 0.2f + 0.2f * (a + b) / a;
  if (errorInPixel * P->minImprovementFactor > dist && isfinite(p->idepth_max)) {
    p->lastTraceUV[0] = (uMax + uMin) * 0.5f;
    p->lastTraceUV[1] = (vMax + vMin) * 0.5f;
    p->lastTracePixelInterval = dist;
    return p->lastTraceStatus = SOS_IPS_BADCONDITION;
  }
  if (errorInPixel > 10) errorInPixel = 10;

  dx /= dist;
  dy /= dist;
  if (dist > maxPixSearch) {
    uMax = uMin + maxPixSearch * dx;
    vMax = vMin + maxPixSearch * dy;
    dist = maxPixSearch;
  }
  int numSteps = (int)(1.9999f + dist / P->stepsize);
  float randShift = uMin * 1000 - floorf(uMin * 1000);
  float ptx = uMin - randShift * dx, pty = vMin - randShift * dy;
  float rp[8][2];
  for (int idx = 0; idx < 8; idx++) {
    rp[idx][0] = K[0] * PATTERN[idx][0] + K[1] * PATTERN[idx][1];
    rp[idx][1] = K[3] * PATTERN[idx][0] + K[4] * PATTERN[idx][1];
  }
  if (!isfinite(dx) || !isfinite(dy)) OOB_RETURN(SOS_IPS_OOB);

  float errors[100];
  float bestU = 0, bestV = 0, bestEnergy = 1e10f;
  int bestIdx = -1;
  if (numSteps >= 100) numSteps = 99;
  for (int i = 0; i < numSteps; i++) {
    float energy = 0;
    for (int idx = 0; idx < 8; idx++) {
      float hitColor = interp31(dI, (float)(ptx + rp[idx][0]), (float)(pty + rp[idx][1]), w, h);
      if (!isfinite(hitColor)) {
        energy += 1e5f;
        continue;
      }
      float residual = hitColor - (float)(aff[0] * p->color[idx] + aff[1]);
      float hw = fabsf(residual) < P->huberTH ? 1 : P->huberTH / fabsf(residual);
      energy += hw * residual * residual * (2 - hw);
    }
    errors[i] = energy;
    if (energy < bestEnergy) {
      bestU = ptx;
      bestV = pty;
      bestEnergy = energy;
      bestIdx = i;
    }
    ptx += dx;
    pty += dy;
  }
  float secondBest = 1e10f;
  for (int i = 0; i < numSteps; i++)
    if ((i < bestIdx - P->minTraceTestRadius || i > bestIdx + P->minTraceTestRadius) && errors[i] < secondBest) secondBest = errors[i];
  float newQuality = secondBest / bestEnergy;
  if (newQuality < p->quality || numSteps > 10) p->quality = newQuality;

  float uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;
  if (P->GNIterations > 0) bestEnergy = 1e5f;
  for (int it = 0; it < P->GNIterations; it++) {
    float H = 1, bb = 0, energy = 0;
    for (int idx = 0; idx < 8; idx++) {
      float hit[3];
      interp33(dI, (float)(bestU + rp[idx][0]), (float)(bestV + rp[idx][1]), w, h, hit);
      if (!isfinite(hit[0])) {
        energy += 1e5f;
        continue;
      }
      float residual = hit[0] - (aff[0] * p->color[idx] + aff[1]);
      float dResdDist = dx * hit[1] + dy * hit[2];
      float hw = fabsf(residual) < P->huberTH ? 1 : P->huberTH / fabsf(residual);
      H += hw * dResdDist * dResdDist;
      bb += hw * residual * dResdDist;
      energy += p->weights[idx] * p->weights[idx] * hw * residual * residual * (2 - hw);
    }
    if (energy > bestEnergy) {
      stepBack *= 0.5f;
      bestU = uBak + stepBack * dx;
      bestV = vBak + stepBack * dy;
    } else {
      float step = -gnstepsize * bb / H;
      if (step < -0.5f) step = -0.5f;
      else if (step > 0.5f) step = 0.5f;
      if (!isfinite(step)) step = 0;
      uBak = bestU;
      vBak = bestV;
      stepBack = step;
      bestU += step * dx;
      bestV += step * dy;
      bestEnergy = energy;
    }
    if (fabsf(stepBack) < P->GNThreshold) break;
  }
  if (!(bestEnergy < p->energyTH * P->extraSlackOnTH)) {
    p->lastTracePixelInterval = 0;
    p->lastTraceUV[0] = p->lastTraceUV[1] = -1;
    if (p->lastTraceStatus == SOS_IPS_OUTLIER) return p->lastTraceStatus = SOS_IPS_OOB;
    return p->lastTraceStatus = SOS_IPS_OUTLIER;
  }
  if (dx * dx > dy * dy) {
    p->idepth_min = (pr2 * (bestU - errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
    p->idepth_max = (pr2 * (bestU + errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
  } else {
    p->idepth_min = (pr2 * (bestV - errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
    p->idepth_max = (pr2 * (bestV + errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
  }
  if (p->idepth_min > p->idepth_max) {
    float t = p->idepth_min;
    p->idepth_min = p->idepth_max;
    p->idepth_max = t;
  }
  if (!isfinite(p->idepth_min) || !isfinite(p->idepth_max) || (p->idepth_max < 0)) {
    p->lastTracePixelInterval = 0;
    p->lastTraceUV[0] = p->lastTraceUV[1] = -1;
    return p->lastTraceStatus = SOS_IPS_OUTLIER;
  }
  p->lastTracePixelInterval = 2 * errorInPixel;
  p->lastTraceUV[0] = bestU;
  p->lastTraceUV[1] = bestV;
  return p->lastTraceStatus = SOS_IPS_GOOD;
#undef OOB_RETURN
}

void orc_immature_trace(const sos_trace_params *P, const float *frame_dI, int w, int h, int count, sos_immature *pts,
                        const float *KRKi, const float *Kt, const float *aff) { /* the loop of FS/FullSystem.cpp:334-350 */
  for (int i = 0; i < count; i++) trace_one(P, frame_dI, w, h, &pts[i], KRKi, Kt, aff);
}

/* ---- point activation ------------------------------------------------------------------------------------------------
 * ImmaturePoint::linearizeResidual  FS/ImmaturePoint.cpp:475-545 (projectPoint: FS/ResidualProjections.h:52-73,
 * derive_idepth: :33-41) and FullSystem::optimizeImmaturePoint  FS/FullSystemOptPoint.cpp:47-192. */
typedef struct {
  int state_state, state_NewState;
  float state_energy, state_NewEnergy; /* doubles in the reference, but they only ever hold float values */
  int target;
} tmp_res;

static float lin_residual(const sos_activate_params *P, const sos_calib *C, int w, int h, const float *dIl, const sos_pair_tfm *T,
                          const sos_immature *p, float slack, tmp_res *tr, float *Hdd, float *bd, float idepth) {
  if (tr->state_state == SOS_RES_OOB) { /* :479-482 */
    tr->state_NewState = SOS_RES_OOB;
    return tr->state_energy;
  }
  const float wM3G = (float)(w - 3), hM3G = (float)(h - 3);
  const float *R = T->R, *t = T->t;
  float energyLeft = 0;
  for (int idx = 0; idx < 8; idx++) { /* :497-533 */
    int dx = PATTERN[idx][0], dy = PATTERN[idx][1];
    float KliP0 = (p->u + dx - C->cxl) * C->fxli;
    float KliP1 = (p->v + dy - C->cyl) * C->fyli;
    float ptp0 = R[0] * KliP0 + R[1] * KliP1 + R[2] + t[0] * idepth;
    float ptp1 = R[3] * KliP0 + R[4] * KliP1 + R[5] + t[1] * idepth;
    float ptp2 = R[6] * KliP0 + R[7] * KliP1 + R[8] + t[2] * idepth;
    float drescale = 1.0f / ptp2;
    int ok = drescale > 0;
    float u = 0, v = 0, Ku = 0, Kv = 0;
    if (ok) {
      u = ptp0 * drescale;
      v = ptp1 * drescale;
      Ku = u * C->fxl + C->cxl;
      Kv = v * C->fyl + C->cyl;
      ok = Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
    }
    if (!ok) { /* :505-509 -- Hdd / bd keep the terms of the pattern pixels before this one */
      tr->state_NewState = SOS_RES_OOB;
      return tr->state_energy;
    }
    float hit[3];
    interp33(dIl, Ku, Kv, w, h, hit);
    if (!isfinite(hit[0])) { /* :513-516 */
      tr->state_NewState = SOS_RES_OOB;
      return tr->state_energy;
    }
    float residual = hit[0] - (T->aff[0] * p->color[idx] + T->aff[1]);
    float hw = fabsf(residual) < P->huberTH ? 1 : P->huberTH / fabsf(residual);
    energyLeft += p->weights[idx] * p->weights[idx] * hw * residual * residual * (2 - hw);
    float dxInterp = hit[1] * C->fxl, dyInterp = hit[2] * C->fyl;
    float d_idepth = (dxInterp * drescale * (t[0] - t[2] * u) + dyInterp * drescale * (t[1] - t[2] * v)) * SOS_SCALE_IDEPTH;
    hw *= p->weights[idx] * p->weights[idx];
    *Hdd += (hw * d_idepth) * d_idepth;
    *bd += (hw * residual) * d_idepth;
  }
  if (energyLeft > p->energyTH * slack) { /* :535-541 */
    energyLeft = p->energyTH * slack;
    tr->state_NewState = SOS_RES_OUTLIER;
  } else {
    tr->state_NewState = SOS_RES_IN;
  }
  tr->state_NewEnergy = energyLeft;
  return energyLeft;
}

void orc_immature_activate(const sos_activate_params *P, const sos_calib *C, int w, int h, int n, const float *const *dI,
                           const sos_pair_tfm *pairs, int count, const sos_immature *pts, const int32_t *hostOf,
                           sos_activation *out) {
  for (int i = 0; i < count; i++) {
    const sos_immature *p = &pts[i];
    sos_activation *o = &out[i];
    memset(o, 0, sizeof(*o));
    const int host = hostOf[i];
    tmp_res tr[SOS_MAX_FRAMES];
    int nres = 0;
    for (int f = 0; f < n; f++) /* :49-58 */
      if (f != host) {
        tr[nres].state_NewEnergy = tr[nres].state_energy = 0;
        tr[nres].state_NewState = SOS_RES_OUTLIER;
        tr[nres].state_state = SOS_RES_IN;
        tr[nres].target = f;
        nres++;
      }
    float lastEnergy = 0, lastHdd = 0, lastbd = 0;
    float currentIdepth = (p->idepth_max + p->idepth_min) * 0.5f;
    for (int k = 0; k < nres; k++) { /* :68-73 */
      lastEnergy = (float)((double)lastEnergy + (double)lin_residual(P, C, w, h, dI[tr[k].target], &pairs[host + n * tr[k].target], p,
                                                                     1000, &tr[k], &lastHdd, &lastbd, currentIdepth));
      tr[k].state_state = tr[k].state_NewState;
      tr[k].state_energy = tr[k].state_NewEnergy;
    }
    o->idepth = currentIdepth; o->energy = lastEnergy; o->Hdd = lastHdd; o->bd = lastbd;
    if (!isfinite(lastEnergy) || lastHdd < P->minIdepthH_act) { /* :75-80 */
      o->status = SOS_ACT_SKIP;
      continue;
    }
    float lambda = 0.1f;
    int skip = 0, it = 0;
    for (int iteration = 0; iteration < P->GNIts; iteration++) { /* :88-130 */
      it++;
      float H = lastHdd;
      H *= 1 + lambda;
      float step = (float)((1.0 / (double)H) * (double)lastbd);
      float newIdepth = currentIdepth - step;
      float newHdd = 0, newbd = 0, newEnergy = 0;
      for (int k = 0; k < nres; k++)
        newEnergy = (float)((double)newEnergy + (double)lin_residual(P, C, w, h, dI[tr[k].target], &pairs[host + n * tr[k].target], p, 1,
                                                                     &tr[k], &newHdd, &newbd, newIdepth));
      if (!isfinite(lastEnergy) || newHdd < P->minIdepthH_act) { /* :102-107 */
        skip = 1;
        break;
      }
      if (newEnergy < lastEnergy) {
        currentIdepth = newIdepth;
        lastHdd = newHdd;
        lastbd = newbd;
        lastEnergy = newEnergy;
        for (int k = 0; k < nres; k++) {
          tr[k].state_state = tr[k].state_NewState;
          tr[k].state_energy = tr[k].state_NewEnergy;
        }
        lambda = (float)(lambda * 0.5);
      } else {
        lambda *= 5;
      }
      if ((double)fabsf(step) < 0.0001 * (double)currentIdepth) break; /* :128-129 */
    }
    o->idepth = currentIdepth; o->energy = lastEnergy; o->Hdd = lastHdd; o->bd = lastbd; o->iterations = it;
    if (skip) {
      o->status = SOS_ACT_SKIP;
      continue;
    }
    if (!isfinite(currentIdepth)) { /* :132-137 */
      o->status = SOS_ACT_DELETE;
      continue;
    }
    int numGoodRes = 0;
    for (int k = 0; k < nres; k++)
      if (tr[k].state_state == SOS_RES_IN) {
        numGoodRes++;
        o->inMask |= 1u << tr[k].target;
      }
    if (numGoodRes < P->minObs) { /* :144-149 */
      o->status = SOS_ACT_DELETE;
      continue;
    }
    if (!isfinite(p->energyTH)) { /* PointHessian ctor copies energyTH (FS/HessianBlocks.cpp:33-52); :151-155 */
      o->status = SOS_ACT_DELETE;
      continue;
    }
    o->status = SOS_ACT_ACTIVATED;
  }
}

/* ---- candidate selection -----------------------------------------------------------------------------------------------
 * CoarseDistanceMap::makeDistanceMap / growDistBFS / addIntoDistFinal  FS/CoarseTracker.cpp:793-925 and the candidate
 * loop of FullSystem::activatePointsMT  FS/FullSystem.cpp:375-470, as written there (two coordinate lists swapped per
 * round, the even / odd round bodies spelled out). */
#include <stdlib.h>

typedef struct {
  int w1, h1;
  float *dist;
  int *l1, *l2; /* (x, y) pairs */
} distmap;

static void grow_bfs(distmap *m, int bfsNum) { /* :828-917 */
  int w1 = m->w1, h1 = m->h1;
  float *D = m->dist;
  for (int k = 1; k < 40; k++) {
    int bfsNum2 = bfsNum;
    int *t = m->l1; m->l1 = m->l2; m->l2 = t;
    bfsNum = 0;
    if (k % 2 == 0) {
      for (int i = 0; i < bfsNum2; i++) {
        int x = m->l2[2 * i], y = m->l2[2 * i + 1];
        if (x == 0 || y == 0 || x == w1 - 1 || y == h1 - 1) continue;
        int idx = x + y * w1;
        if (D[idx + 1] > k) { D[idx + 1] = k; m->l1[2 * bfsNum] = x + 1; m->l1[2 * bfsNum + 1] = y; bfsNum++; }
        if (D[idx - 1] > k) { D[idx - 1] = k; m->l1[2 * bfsNum] = x - 1; m->l1[2 * bfsNum + 1] = y; bfsNum++; }
        if (D[idx + w1] > k) { D[idx + w1] = k; m->l1[2 * bfsNum] = x; m->l1[2 * bfsNum + 1] = y + 1; bfsNum++; }
        if (D[idx - w1] > k) { D[idx - w1] = k; m->l1[2 * bfsNum] = x; m->l1[2 * bfsNum + 1] = y - 1; bfsNum++; }
      }
    } else {
      for (int i = 0; i < bfsNum2; i++) {
        int x = m->l2[2 * i], y = m->l2[2 * i + 1];
        if (x == 0 || y == 0 || x == w1 - 1 || y == h1 - 1) continue;
        int idx = x + y * w1;
        if (D[idx + 1] > k) { D[idx + 1] = k; m->l1[2 * bfsNum] = x + 1; m->l1[2 * bfsNum + 1] = y; bfsNum++; }
        if (D[idx - 1] > k) { D[idx - 1] = k; m->l1[2 * bfsNum] = x - 1; m->l1[2 * bfsNum + 1] = y; bfsNum++; }
        if (D[idx + w1] > k) { D[idx + w1] = k; m->l1[2 * bfsNum] = x; m->l1[2 * bfsNum + 1] = y + 1; bfsNum++; }
        if (D[idx - w1] > k) { D[idx - w1] = k; m->l1[2 * bfsNum] = x; m->l1[2 * bfsNum + 1] = y - 1; bfsNum++; }
        if (D[idx + 1 + w1] > k) { D[idx + 1 + w1] = k; m->l1[2 * bfsNum] = x + 1; m->l1[2 * bfsNum + 1] = y + 1; bfsNum++; }
        if (D[idx - 1 + w1] > k) { D[idx - 1 + w1] = k; m->l1[2 * bfsNum] = x - 1; m->l1[2 * bfsNum + 1] = y + 1; bfsNum++; }
        if (D[idx - 1 - w1] > k) { D[idx - 1 - w1] = k; m->l1[2 * bfsNum] = x - 1; m->l1[2 * bfsNum + 1] = y - 1; bfsNum++; }
        if (D[idx + 1 - w1] > k) { D[idx + 1 - w1] = k; m->l1[2 * bfsNum] = x + 1; m->l1[2 * bfsNum + 1] = y - 1; bfsNum++; }
      }
    }
  }
}

float orc_next_min_act_dist(float d, int nPoints, float desired) { /* FS/FullSystem.cpp:377-399 */
  if (nPoints < desired * 0.66) d -= 0.8;
  if (nPoints < desired * 0.8) d -= 0.5;
  else if (nPoints < desired * 0.9) d -= 0.2;
  else if (nPoints < desired) d -= 0.1;
  if (nPoints > desired * 1.5) d += 0.8;
  if (nPoints > desired * 1.3) d += 0.5;
  if (nPoints > desired * 1.15) d += 0.2;
  if (nPoints > desired) d += 0.1;
  if (d < 0) d = 0;
  if (d > 4) d = 4;
  return d;
}

int orc_activate_select(int w1, int h1, int nFrames, int newest, const float *KRKi, const float *Kt, int nActive,
                        const float *act_u, const float *act_v, const float *act_id, const int32_t *act_host,
                        float currentMinActDist, float minTraceQuality, int nCand, const sos_immature *cand,
                        const int32_t *cand_host, const float *cand_type, const uint8_t *hostFlagged, int8_t *decision,
                        float *distFinal) {
  (void)nFrames;
  distmap m;
  m.w1 = w1; m.h1 = h1;
  m.dist = (float *)malloc(sizeof(float) * (size_t)w1 * h1);
  /* (the reference sizes its lists by the image, FS/CoarseTracker.cpp:774-776; several active points may project into one cell, so the
   * seed list is given room for all of them here -- test windows smaller than their point count must not write past the end) */
  const size_t qn = (size_t)w1 * h1 > (size_t)nActive + 1 ? (size_t)w1 * h1 : (size_t)nActive + 1;
  m.l1 = (int *)malloc(sizeof(int) * 2 * qn);
  m.l2 = (int *)malloc(sizeof(int) * 2 * qn);
  for (int i = 0; i < w1 * h1; i++) m.dist[i] = 1000; /* :797-798 */
  int numItems = 0;
  for (int i = 0; i < nActive; i++) { /* :803-823 */
    int f = act_host[i];
    if (f == newest) continue;
    const float *K = KRKi + 9 * f, *T = Kt + 3 * f;
    float p0 = K[0] * act_u[i] + K[1] * act_v[i] + K[2] + T[0] * act_id[i];
    float p1 = K[3] * act_u[i] + K[4] * act_v[i] + K[5] + T[1] * act_id[i];
    float p2 = K[6] * act_u[i] + K[7] * act_v[i] + K[8] + T[2] * act_id[i];
    int u = p0 / p2 + 0.5f;
    int v = p1 / p2 + 0.5f;
    if (!(u > 0 && v > 0 && u < w1 && v < h1)) continue;
    m.dist[u + w1 * v] = 0;
    m.l1[2 * numItems] = u; m.l1[2 * numItems + 1] = v;
    numItems++;
  }
  grow_bfs(&m, numItems);
  for (int i = 0; i < nCand; i++) { /* FS/FullSystem.cpp:417-470 */
    const sos_immature *ph = &cand[i];
    int f = cand_host[i];
    if (!isfinite(ph->idepth_max) || ph->lastTraceStatus == SOS_IPS_OUTLIER) { decision[i] = -1; continue; }
    int canActivate = (ph->lastTraceStatus == SOS_IPS_GOOD || ph->lastTraceStatus == SOS_IPS_SKIPPED ||
                       ph->lastTraceStatus == SOS_IPS_BADCONDITION || ph->lastTraceStatus == SOS_IPS_OOB) &&
                      ph->lastTracePixelInterval < 8 && ph->quality > minTraceQuality && (ph->idepth_max + ph->idepth_min) > 0;
    if (!canActivate) {
      decision[i] = (hostFlagged[f] || ph->lastTraceStatus == SOS_IPS_OOB) ? -1 : 0;
      continue;
    }
    const float *K = KRKi + 9 * f, *T = Kt + 3 * f;
    float idm = 0.5f * (ph->idepth_max + ph->idepth_min);
    float p0 = K[0] * ph->u + K[1] * ph->v + K[2] + T[0] * idm;
    float p1 = K[3] * ph->u + K[4] * ph->v + K[5] + T[1] * idm;
    float p2 = K[6] * ph->u + K[7] * ph->v + K[8] + T[2] * idm;
    int u = p0 / p2 + 0.5f;
    int v = p1 / p2 + 0.5f;
    if (u > 0 && v > 0 && u < w1 && v < h1) {
      float dist = m.dist[u + w1 * v] + (p0 - floorf(p0));
      if (dist >= currentMinActDist * cand_type[i]) {
        m.l1[0] = u; m.l1[1] = v; /* addIntoDistFinal, FS/CoarseTracker.cpp:919-925 */
        m.dist[u + w1 * v] = 0;
        grow_bfs(&m, 1);
        decision[i] = 1;
      } else {
        decision[i] = 0;
      }
    } else {
      decision[i] = -1;
    }
  }
  if (distFinal) memcpy(distFinal, m.dist, sizeof(float) * (size_t)w1 * h1);
  free(m.dist); free(m.l1); free(m.l2);
  return 0;
}
