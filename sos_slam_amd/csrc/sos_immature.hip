// sos_immature.hip -- immature-point front of the window on gfx950 (first part):
//   k_immature_init    ImmaturePoint::ImmaturePoint   FS/ImmaturePoint.cpp:30-59
//   k_immature_trace   ImmaturePoint::traceOn         FS/ImmaturePoint.cpp:70-415  (loop of FS/FullSystem.cpp:334-350)
// One thread per immature point: the epipolar search is a data-dependent sequential walk (up to 99 steps x 8 bilinear
// taps, then <= 3 Gauss-Newton refinements), so points are the only parallel axis; a frame traces ~10^3..10^4 of them.
// Same fp32 convention as the backend kernels (no FMA contraction, source-order sums): results are bit-identical to
// oracle/orc_immature.c.  Records travel through device-mapped pinned memory (one in/out block per call).
#include "sos_common.h"

namespace {
__constant__ int c_pattern[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ const float *texel(const float *dI, int ix, int iy, int w, int h) {
  return dI + 3 * ((size_t)clampi(ix, 0, w - 1) + (size_t)clampi(iy, 0, h - 1) * w);
}
__device__ __forceinline__ float interp31(const float *dI, float x, float y, int w, int h) {  // U/globalFuncs.h:122-136
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  return dxdy * texel(dI, ix + 1, iy + 1, w, h)[0] + (dy - dxdy) * texel(dI, ix, iy + 1, w, h)[0] +
         (dx - dxdy) * texel(dI, ix + 1, iy, w, h)[0] + (1 - dx - dy + dxdy) * texel(dI, ix, iy, w, h)[0];
}
__device__ __forceinline__ void interp33(const float *dI, float x, float y, int w, int h, float *o) {  // :68-82
  const int ix = (int)x, iy = (int)y;
  const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const float *a = texel(dI, ix, iy, w, h), *b = texel(dI, ix + 1, iy, w, h), *c = texel(dI, ix, iy + 1, w, h),
              *d = texel(dI, ix + 1, iy + 1, w, h);
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = dxdy * d[k] + (dy - dxdy) * c[k] + (dx - dxdy) * b[k] + (1 - dx - dy + dxdy) * a[k];
}
__device__ __forceinline__ void interp33bilin(const float *dI, float x, float y, int w, int h, float *o) {  // :161-182
  const int ix = (int)x, iy = (int)y;
  const float tl = texel(dI, ix, iy, w, h)[0], tr = texel(dI, ix + 1, iy, w, h)[0], bl = texel(dI, ix, iy + 1, w, h)[0],
              br = texel(dI, ix + 1, iy + 1, w, h)[0];
  const float dx = x - ix, dy = y - iy;
  const float topInt = dx * tr + (1 - dx) * tl, botInt = dx * br + (1 - dx) * bl;
  const float leftInt = dy * bl + (1 - dy) * tl, rightInt = dy * br + (1 - dy) * tr;
  o[0] = dx * rightInt + (1 - dx) * leftInt;
  o[1] = rightInt - leftInt;
  o[2] = botInt - topInt;
}

__global__ void k_immature_init(sos_trace_params P, const float *__restrict__ dI, int w, int h, int count,
                                const int *__restrict__ u, const int *__restrict__ v, sos_immature *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  sos_immature p;
  memset(&p, 0, sizeof(p));
  p.u = (float)u[i];
  p.v = (float)v[i];
  p.idepth_min = 0;
  p.idepth_max = NAN;
  p.lastTraceStatus = SOS_IPS_UNINITIALIZED;
  float g00 = 0, g01 = 0, g10 = 0, g11 = 0;
  bool bad = false;
  for (int idx = 0; idx < 8; idx++) {
    float ptc[3];
    interp33bilin(dI, p.u + c_pattern[idx][0], p.v + c_pattern[idx][1], w, h, ptc);
    p.color[idx] = ptc[0];
    if (!isfinite(p.color[idx])) {
      p.energyTH = NAN;
      bad = true;
      break;
    }
    g00 += ptc[1] * ptc[1]; g01 += ptc[1] * ptc[2]; g10 += ptc[2] * ptc[1]; g11 += ptc[2] * ptc[2];
    p.weights[idx] = sqrtf(P.outlierTHSumComponent / (P.outlierTHSumComponent + (ptc[1] * ptc[1] + ptc[2] * ptc[2])));
  }
  p.gradH[0] = g00; p.gradH[1] = g01; p.gradH[2] = g10; p.gradH[3] = g11;
  if (!bad) {
    p.energyTH = 8 * P.outlierTH;
    p.energyTH *= P.overallEnergyTHWeight * P.overallEnergyTHWeight;
    p.quality = 10000;
  }
  out[i] = p;
}

struct TraceArgs {
  float K[9], Kt[3], aff[2];
};

__device__ __forceinline__ int trace_one(const sos_trace_params &P, const float *dI, int w, int h, sos_immature &p,
                                         const TraceArgs &A) {
#define OOB_RETURN(st)             \
  do {                             \
    p.lastTraceUV[0] = -1;         \
    p.lastTraceUV[1] = -1;         \
    p.lastTracePixelInterval = 0;  \
    return p.lastTraceStatus = (st); \
  } while (0)
  const float *K = A.K, *Kt = A.Kt, *aff = A.aff;
  if (p.lastTraceStatus == SOS_IPS_OOB) return p.lastTraceStatus;
  const float maxPixSearch = (w + h) * P.maxPixSearch;
  const float pr0 = K[0] * p.u + K[1] * p.v + K[2] * 1.0f, pr1 = K[3] * p.u + K[4] * p.v + K[5] * 1.0f,
              pr2 = K[6] * p.u + K[7] * p.v + K[8] * 1.0f;
  const float ptpMin0 = pr0 + Kt[0] * p.idepth_min, ptpMin1 = pr1 + Kt[1] * p.idepth_min, ptpMin2 = pr2 + Kt[2] * p.idepth_min;
  const float uMin = ptpMin0 / ptpMin2, vMin = ptpMin1 / ptpMin2;
  if (!(uMin > 4 && vMin > 4 && uMin < w - 5 && vMin < h - 5)) OOB_RETURN(SOS_IPS_OOB);
  float dist, uMax, vMax;
  if (isfinite(p.idepth_max)) {
    const float q0 = pr0 + Kt[0] * p.idepth_max, q1 = pr1 + Kt[1] * p.idepth_max, q2 = pr2 + Kt[2] * p.idepth_max;
    uMax = q0 / q2;
    vMax = q1 / q2;
    if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) OOB_RETURN(SOS_IPS_OOB);
    dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
    dist = sqrtf(dist);
    if (dist < P.slackInterval) {
      p.lastTraceUV[0] = (uMax + uMin) * 0.5f;
      p.lastTraceUV[1] = (vMax + vMin) * 0.5f;
      p.lastTracePixelInterval = dist;
      return p.lastTraceStatus = SOS_IPS_SKIPPED;
    }
  } else {
    dist = maxPixSearch;
    const float q0 = pr0 + Kt[0] * 0.01f, q1 = pr1 + Kt[1] * 0.01f, q2 = pr2 + Kt[2] * 0.01f;
    uMax = q0 / q2;
    vMax = q1 / q2;
    const float ddx = uMax - uMin, ddy = vMax - vMin;
    const float d = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
    uMax = uMin + dist * ddx * d;
    vMax = vMin + dist * ddy * d;
    if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) OOB_RETURN(SOS_IPS_OOB);
  }
  if (!(p.idepth_min < 0 || (ptpMin2 > 0.75f && ptpMin2 < 1.5f))) OOB_RETURN(SOS_IPS_OOB);

  float dx = P.stepsize * (uMax - uMin), dy = P.stepsize * (vMax - vMin);
  const float *G = p.gradH;
  const float a = (dx * G[0] + dy * G[2]) * dx + (dx * G[1] + dy * G[3]) * dy;
  const float b = (dy * G[0] + (-dx) * G[2]) * dy + (dy * G[1] + (-dx) * G[3]) * (-dx);
  float errorInPixel = 0.2f + 0.2f * (a + b) / a;
  if (errorInPixel * P.minImprovementFactor > dist && isfinite(p.idepth_max)) {
    p.lastTraceUV[0] = (uMax + uMin) * 0.5f;
    p.lastTraceUV[1] = (vMax + vMin) * 0.5f;
    p.lastTracePixelInterval = dist;
    return p.lastTraceStatus = SOS_IPS_BADCONDITION;
  }
  if (errorInPixel > 10) errorInPixel = 10;

  dx /= dist;
  dy /= dist;
  if (dist > maxPixSearch) {
    uMax = uMin + maxPixSearch * dx;
    vMax = vMin + maxPixSearch * dy;
    dist = maxPixSearch;
  }
  int numSteps = (int)(1.9999f + dist / P.stepsize);
  const float randShift = uMin * 1000 - floorf(uMin * 1000);
  float ptx = uMin - randShift * dx, pty = vMin - randShift * dy;
  float rp[8][2];
#pragma unroll
  for (int idx = 0; idx < 8; idx++) {
    rp[idx][0] = K[0] * c_pattern[idx][0] + K[1] * c_pattern[idx][1];
    rp[idx][1] = K[3] * c_pattern[idx][0] + K[4] * c_pattern[idx][1];
  }
  if (!isfinite(dx) || !isfinite(dy)) OOB_RETURN(SOS_IPS_OOB);

  // the discrete search.  errors[] is only needed for the second-best score outside +-radius of the best: kept in
  // local memory (99 floats per thread would not fit the register budget anyway)
  float errors[100];
  float bestU = 0, bestV = 0, bestEnergy = 1e10f;
  int bestIdx = -1;
  if (numSteps >= 100) numSteps = 99;
  for (int i = 0; i < numSteps; i++) {
    float energy = 0;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
      const float hitColor = interp31(dI, (float)(ptx + rp[idx][0]), (float)(pty + rp[idx][1]), w, h);
      if (!isfinite(hitColor)) {
        energy += 1e5f;
        continue;
      }
      const float residual = hitColor - (float)(aff[0] * p.color[idx] + aff[1]);
      const float hw = fabsf(residual) < P.huberTH ? 1 : P.huberTH / fabsf(residual);
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
    if ((i < bestIdx - P.minTraceTestRadius || i > bestIdx + P.minTraceTestRadius) && errors[i] < secondBest) secondBest = errors[i];
  const float newQuality = secondBest / bestEnergy;
  if (newQuality < p.quality || numSteps > 10) p.quality = newQuality;

  float uBak = bestU, vBak = bestV, stepBack = 0;
  const float gnstepsize = 1;
  if (P.GNIterations > 0) bestEnergy = 1e5f;
  for (int it = 0; it < P.GNIterations; it++) {
    float H = 1, bb = 0, energy = 0;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
      float hit[3];
      interp33(dI, (float)(bestU + rp[idx][0]), (float)(bestV + rp[idx][1]), w, h, hit);
      if (!isfinite(hit[0])) {
        energy += 1e5f;
        continue;
      }
      const float residual = hit[0] - (aff[0] * p.color[idx] + aff[1]);
      const float dResdDist = dx * hit[1] + dy * hit[2];
      const float hw = fabsf(residual) < P.huberTH ? 1 : P.huberTH / fabsf(residual);
      H += hw * dResdDist * dResdDist;
      bb += hw * residual * dResdDist;
      energy += p.weights[idx] * p.weights[idx] * hw * residual * residual * (2 - hw);
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
    if (fabsf(stepBack) < P.GNThreshold) break;
  }
  if (!(bestEnergy < p.energyTH * P.extraSlackOnTH)) {
    p.lastTracePixelInterval = 0;
    p.lastTraceUV[0] = p.lastTraceUV[1] = -1;
    if (p.lastTraceStatus == SOS_IPS_OUTLIER) return p.lastTraceStatus = SOS_IPS_OOB;
    return p.lastTraceStatus = SOS_IPS_OUTLIER;
  }
  if (dx * dx > dy * dy) {
    p.idepth_min = (pr2 * (bestU - errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
    p.idepth_max = (pr2 * (bestU + errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
  } else {
    p.idepth_min = (pr2 * (bestV - errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
    p.idepth_max = (pr2 * (bestV + errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
  }
  if (p.idepth_min > p.idepth_max) {
    const float t = p.idepth_min;
    p.idepth_min = p.idepth_max;
    p.idepth_max = t;
  }
  if (!isfinite(p.idepth_min) || !isfinite(p.idepth_max) || (p.idepth_max < 0)) {
    p.lastTracePixelInterval = 0;
    p.lastTraceUV[0] = p.lastTraceUV[1] = -1;
    return p.lastTraceStatus = SOS_IPS_OUTLIER;
  }
  p.lastTracePixelInterval = 2 * errorInPixel;
  p.lastTraceUV[0] = bestU;
  p.lastTraceUV[1] = bestV;
  return p.lastTraceStatus = SOS_IPS_GOOD;
#undef OOB_RETURN
}

__global__ __launch_bounds__(64) void k_immature_trace(sos_trace_params P, const float *__restrict__ dI, int w, int h, int count,
                                                       sos_immature *__restrict__ pts, TraceArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  sos_immature p = pts[i];
  trace_one(P, dI, w, h, p, A);
  pts[i] = p;
}

// whole traceNewCoarse in one launch: every point carries the index of its host keyframe, the host -> frame
// quantities come from a table
__global__ __launch_bounds__(64) void k_immature_trace_batch(sos_trace_params P, const float *__restrict__ dI, int w, int h, int count,
                                                             sos_immature *__restrict__ pts, const int *__restrict__ hostOf,
                                                             const TraceArgs *__restrict__ table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  sos_immature p = pts[i];
  const TraceArgs A = table[hostOf[i]];
  trace_one(P, dI, w, h, p, A);
  pts[i] = p;
}

// device-mapped pinned in/out block for the point records, grown on demand (process lifetime, per device)
struct Stage {
  char *host = nullptr, *dev = nullptr;
  size_t bytes = 0;
};
Stage g_stage[16];
int stage_ensure(int device, size_t bytes, Stage **out) {
  Stage &s = g_stage[device & 15];
  if (bytes > s.bytes) {
    if (s.host) hipHostFree(s.host);
    s.host = nullptr;
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipHostMalloc((void **)&s.host, want, hipHostMallocMapped) != hipSuccess) return SOS_ERR_NOMEM;
    if (hipHostGetDevicePointer((void **)&s.dev, s.host, 0) != hipSuccess) return SOS_ERR_HIP;
    s.bytes = want;
  }
  *out = &s;
  return SOS_OK;
}
}  // namespace

extern "C" int sos_immature_init(sos_ctx *c, const sos_trace_params *prm, int hostSlot, int count, const int32_t *u,
                                 const int32_t *v, sos_immature *out) {
  if (!c || !prm || count < 0 || (count && (!u || !v || !out))) return SOS_ERR_ARG;
  if (hostSlot < 0 || hostSlot >= SOS_MAX_SLOTS || !c->dI[hostSlot][0]) return SOS_ERR_STATE;
  if (count == 0) return SOS_OK;
  SOS_HIP(hipSetDevice(c->device));
  Stage *st;
  const size_t off_v = sizeof(int32_t) * (size_t)count, off_o = (2 * off_v + 127) / 128 * 128;
  int rc = stage_ensure(c->device, off_o + sizeof(sos_immature) * (size_t)count, &st);
  if (rc) return rc;
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(st->host, u, off_v);
  memcpy(st->host + off_v, v, off_v);
  k_immature_init<<<(count + 63) / 64, 64, 0, c->stream>>>(*prm, c->dI[hostSlot][0], c->w, c->h, count,
                                                           reinterpret_cast<const int *>(st->dev), reinterpret_cast<const int *>(st->dev + off_v),
                                                           reinterpret_cast<sos_immature *>(st->dev + off_o));
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(out, st->host + off_o, sizeof(sos_immature) * (size_t)count);
  return SOS_OK;
}

extern "C" int sos_immature_trace(sos_ctx *c, const sos_trace_params *prm, int frameSlot, int count, sos_immature *pts,
                                  const float *KRKi, const float *Kt, const float *aff) {
  if (!c || !prm || count < 0 || (count && !pts) || !KRKi || !Kt || !aff) return SOS_ERR_ARG;
  if (frameSlot < 0 || frameSlot >= SOS_MAX_SLOTS || !c->dI[frameSlot][0]) return SOS_ERR_STATE;
  if (count == 0) return SOS_OK;
  SOS_HIP(hipSetDevice(c->device));
  Stage *st;
  int rc = stage_ensure(c->device, sizeof(sos_immature) * (size_t)count, &st);
  if (rc) return rc;
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(st->host, pts, sizeof(sos_immature) * (size_t)count);
  TraceArgs A;
  memcpy(A.K, KRKi, sizeof(A.K));
  memcpy(A.Kt, Kt, sizeof(A.Kt));
  memcpy(A.aff, aff, sizeof(A.aff));
  k_immature_trace<<<(count + 63) / 64, 64, 0, c->stream>>>(*prm, c->dI[frameSlot][0], c->w, c->h, count,
                                                            reinterpret_cast<sos_immature *>(st->dev), A);
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(pts, st->host, sizeof(sos_immature) * (size_t)count);
  return SOS_OK;
}

extern "C" int sos_immature_trace_all(sos_ctx *c, const sos_trace_params *prm, int frameSlot, int count, sos_immature *pts,
                                      const int32_t *hostOfPoint, int nhosts, const float *KRKi, const float *Kt, const float *aff) {
  if (!c || !prm || count < 0 || nhosts < 1 || (count && (!pts || !hostOfPoint)) || !KRKi || !Kt || !aff) return SOS_ERR_ARG;
  if (frameSlot < 0 || frameSlot >= SOS_MAX_SLOTS || !c->dI[frameSlot][0]) return SOS_ERR_STATE;
  if (count == 0) return SOS_OK;
  for (int i = 0; i < count; i++)
    if (hostOfPoint[i] < 0 || hostOfPoint[i] >= nhosts) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(c->device));
  Stage *st;
  const size_t off_h = (sizeof(sos_immature) * (size_t)count + 127) / 128 * 128;
  const size_t off_t = (off_h + sizeof(int32_t) * (size_t)count + 127) / 128 * 128;
  int rc = stage_ensure(c->device, off_t + sizeof(TraceArgs) * (size_t)nhosts, &st);
  if (rc) return rc;
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(st->host, pts, sizeof(sos_immature) * (size_t)count);
  memcpy(st->host + off_h, hostOfPoint, sizeof(int32_t) * (size_t)count);
  TraceArgs *tab = reinterpret_cast<TraceArgs *>(st->host + off_t);
  for (int k = 0; k < nhosts; k++) {
    memcpy(tab[k].K, KRKi + 9 * k, sizeof(tab[k].K));
    memcpy(tab[k].Kt, Kt + 3 * k, sizeof(tab[k].Kt));
    memcpy(tab[k].aff, aff + 2 * k, sizeof(tab[k].aff));
  }
  k_immature_trace_batch<<<(count + 63) / 64, 64, 0, c->stream>>>(*prm, c->dI[frameSlot][0], c->w, c->h, count,
                                                                  reinterpret_cast<sos_immature *>(st->dev),
                                                                  reinterpret_cast<const int *>(st->dev + off_h),
                                                                  reinterpret_cast<const TraceArgs *>(st->dev + off_t));
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(pts, st->host, sizeof(sos_immature) * (size_t)count);
  return SOS_OK;
}
