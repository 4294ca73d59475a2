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

#ifndef TRACE_G
#define TRACE_G 16   // lanes per immature point in the trace kernels (1 = thread per point)
#endif
#define TRACE_PPB (64 / TRACE_G)   // points per 64-thread block
struct TraceArgs {
  float K[9], Kt[3], aff[2];
};

// GL = 1: one thread per point.  GL = 16: one 16-lane group (a DPP row) per point -- every lane runs the scalar logic on its own copy
// of the record (identical values, identical branches within the group), and the three loops that carry the work are split over the
// lanes without changing a single operation or its order:
//   * the discrete search: lane gl evaluates the steps gl, gl + 16, ... (positions by the reference's repeated addition, every step's
//     energy summed over the pattern in the reference's order by ONE lane), then the first minimum in step order = the minimum with
//     ties to the lower index (a symmetric 4-stage exchange), the second-best score likewise a plain minimum;
//   * the Gauss-Newton refinement: lanes 0..7 fetch one pattern pixel each, the three running sums are folded in pattern order by
//     every lane from the lanes' addends (the reference's `continue` for a non-finite pixel included).
// A point's dependent memory round trips drop from (steps / 4 + iterations) to about (steps / 16 + iterations), and four times as
// many waves are in flight.  Lanes of a group return together, so the exchanges (width 16) only ever read active lanes.
template <int GL>
__device__ __forceinline__ int trace_one(const sos_trace_params &P, const float *dI, int w, int h, sos_immature &p,
                                         const TraceArgs &A, int gl = 0) {
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

  float bestU = 0, bestV = 0, bestEnergy = 1e10f;
  int bestIdx = -1;
  if (numSteps >= 100) numSteps = 99;
  auto step_energy = [&](float qx, float qy) -> float {
    float energy = 0;
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
      const float hitColor = interp31(dI, (float)(qx + rp[idx][0]), (float)(qy + rp[idx][1]), w, h);
      const float residual = hitColor - (float)(aff[0] * p.color[idx] + aff[1]);
      const float hw = fabsf(residual) < P.huberTH ? 1 : P.huberTH / fabsf(residual);
      const float e = hw * residual * residual * (2 - hw);
      energy += isfinite(hitColor) ? e : 1e5f;
    }
    return energy;
  };
  float secondBest = 1e10f;
  if constexpr (GL == 1) {
    // errors[] is only needed for the second-best score outside +-radius of the best: kept in local memory (99 floats per
    // thread would not fit the register budget anyway).
    // The steps are independent of each other (only the best-so-far bookkeeping is sequential): four of them are
    // evaluated together so that their 4 x 32 texel requests are in flight at once instead of one step per memory round
    // trip; positions by repeated addition and the bookkeeping in step order, as in the one-step loop -> same records.
    float errors[100];
    constexpr int TU = 4;
    for (int i0 = 0; i0 < numSteps; i0 += TU) {
      float qx[TU], qy[TU], en[TU];
#pragma unroll
      for (int k = 0; k < TU; k++) {
        qx[k] = ptx;
        qy[k] = pty;
        ptx += dx;
        pty += dy;
      }
      if (i0 + TU <= numSteps) {  // full group: no per-step branches between the requests
#pragma unroll
        for (int k = 0; k < TU; k++) en[k] = step_energy(qx[k], qy[k]);
      } else {
#pragma unroll
        for (int k = 0; k < TU; k++) en[k] = (i0 + k < numSteps) ? step_energy(qx[k], qy[k]) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < TU; k++) {
        const int i = i0 + k;
        if (i < numSteps) {
          errors[i] = en[k];
          if (en[k] < bestEnergy) {
            bestU = qx[k];
            bestV = qy[k];
            bestEnergy = en[k];
            bestIdx = i;
          }
        }
      }
    }
    for (int i = 0; i < numSteps; i++)
      if ((i < bestIdx - P.minTraceTestRadius || i > bestIdx + P.minTraceTestRadius) && errors[i] < secondBest) secondBest = errors[i];
  } else {
    constexpr int RMAX = (99 + GL - 1) / GL;
    float qx[RMAX], qy[RMAX], en[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
      qx[r] = qy[r] = en[r] = 0.f;
      if (GL * r < numSteps) {  // (uniform in the group)
        for (int k = 0; k < GL && GL * r + k < numSteps; k++) {
          if (k == gl) { qx[r] = ptx; qy[r] = pty; }
          ptx += dx;
          pty += dy;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; r++)
      if (GL * r + gl < numSteps) en[r] = step_energy(qx[r], qy[r]);
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
      const int i = GL * r + gl;
      if (i < numSteps && en[r] < bestEnergy) {
        bestU = qx[r];
        bestV = qy[r];
        bestEnergy = en[r];
        bestIdx = i;
      }
    }
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) {
      const float oe = __shfl_xor(bestEnergy, o, GL), ou = __shfl_xor(bestU, o, GL), ov = __shfl_xor(bestV, o, GL);
      const int oi = __shfl_xor(bestIdx, o, GL);
      // a lane without a candidate carries (1e10, -1): any candidate (energy < 1e10) beats it; equal energies -> the earlier step
      const bool take = (oi >= 0) && (bestIdx < 0 || oe < bestEnergy || (oe == bestEnergy && oi < bestIdx));
      if (take) { bestEnergy = oe; bestU = ou; bestV = ov; bestIdx = oi; }
    }
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
      const int i = GL * r + gl;
      if (i < numSteps && (i < bestIdx - P.minTraceTestRadius || i > bestIdx + P.minTraceTestRadius) && en[r] < secondBest) secondBest = en[r];
    }
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) {
      const float os = __shfl_xor(secondBest, o, GL);
      if (os < secondBest) secondBest = os;
    }
  }
  const float newQuality = secondBest / bestEnergy;
  if (newQuality < p.quality || numSteps > 10) p.quality = newQuality;

  float uBak = bestU, vBak = bestV, stepBack = 0;
  const float gnstepsize = 1;
  if (P.GNIterations > 0) bestEnergy = 1e5f;
  for (int it = 0; it < P.GNIterations; it++) {
    float H = 1, bb = 0, energy = 0;
    if constexpr (GL == 1) {
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
    } else {
      // lane gl < 8: the addends of pattern pixel gl (rp / color / weights picked by a select chain: no dynamic register index)
      float rx = 0, ry = 0, col = 0, wgt = 0;
#pragma unroll
      for (int idx = 0; idx < 8; idx++)
        if ((gl & 7) == idx) { rx = rp[idx][0]; ry = rp[idx][1]; col = p.color[idx]; wgt = p.weights[idx]; }
      float hit[3];
      interp33(dI, (float)(bestU + rx), (float)(bestV + ry), w, h, hit);
      const float residual = hit[0] - (aff[0] * col + aff[1]);
      const float dResdDist = dx * hit[1] + dy * hit[2];
      const float hw = fabsf(residual) < P.huberTH ? 1 : P.huberTH / fabsf(residual);
      const float aH = hw * dResdDist * dResdDist, ab = hw * residual * dResdDist, ae = wgt * wgt * hw * residual * residual * (2 - hw);
      const int fin = isfinite(hit[0]) ? 1 : 0;
#pragma unroll
      for (int idx = 0; idx < 8; idx++) {
        const int f = __shfl(fin, idx, GL);
        const float tH = __shfl(aH, idx, GL), tb = __shfl(ab, idx, GL), te = __shfl(ae, idx, GL);
        if (!f) {
          energy += 1e5f;
        } else {
          H += tH;
          bb += tb;
          energy += te;
        }
      }
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
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) / TRACE_G, gl = threadIdx.x % TRACE_G;
  if (i >= count) return;
  sos_immature p = pts[i];
  trace_one<TRACE_G>(P, dI, w, h, p, A, gl);
  if (gl == 0) pts[i] = p;
}

// whole traceNewCoarse in one launch: every point carries the index of its host keyframe, the host -> frame
// quantities come from a table
__global__ __launch_bounds__(64) void k_immature_trace_batch(sos_trace_params P, const float *__restrict__ dI, int w, int h, int count,
                                                             sos_immature *__restrict__ pts, const int *__restrict__ hostOf,
                                                             const TraceArgs *__restrict__ table) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) / TRACE_G, gl = threadIdx.x % TRACE_G;
  if (i >= count) return;
  sos_immature p = pts[i];
  const TraceArgs A = table[hostOf[i]];
  trace_one<TRACE_G>(P, dI, w, h, p, A, gl);
  if (gl == 0) pts[i] = p;
}

// the device-resident sets (sos_immset): up to SOS_MAX_FRAMES segments, one per host keyframe; block -> segment through a table in
// the kernel arguments
struct ImmSegs {
  sos_immature *ptr[SOS_MAX_FRAMES];
  int blockBegin[SOS_MAX_FRAMES + 1];
  int count[SOS_MAX_FRAMES];
  TraceArgs A[SOS_MAX_FRAMES];
  int nseg;
};
__global__ __launch_bounds__(64) void k_immature_trace_sets(sos_trace_params P, const float *__restrict__ dI, int w, int h, ImmSegs S) {
  int s = 0;
  while (s + 1 < S.nseg && (int)blockIdx.x >= S.blockBegin[s + 1]) s++;
  const int i = ((int)blockIdx.x - S.blockBegin[s]) * TRACE_PPB + (int)threadIdx.x / TRACE_G, gl = threadIdx.x % TRACE_G;
  if (i >= S.count[s]) return;
  sos_immature p = S.ptr[s][i];
  trace_one<TRACE_G>(P, dI, w, h, p, S.A[s], gl);
  if (gl == 0) S.ptr[s][i] = p;
}

// ---- point activation: FullSystem::optimizeImmaturePoint (FS/FullSystemOptPoint.cpp:47-192) ---------------------------
// One wave per candidate: lane = (residual slot g = lane >> 3, pattern pixel = lane & 7), residuals beyond 8 in further
// chunks.  The reference accumulates Hdd / bd in ONE running float over residuals x pattern pixels (and keeps the
// partial terms of a residual that leaves the image half way through its pattern), so the wave replays exactly that
// chain: every lane computes its addend, then the addends are folded in lane order by wave-uniform code.
struct ActImages {
  const float *dI[SOS_MAX_FRAMES];
};
constexpr int ACT_MAXC = (SOS_MAX_FRAMES - 1 + 7) / 8;

__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

struct ActState {
  int st[ACT_MAXC], nw[ACT_MAXC];        // state_state / state_NewState of the residual of this lane group, per chunk
  float stE[ACT_MAXC], nwE[ACT_MAXC];    // state_energy / state_NewEnergy
};

__device__ __forceinline__ float act_eval(const sos_activate_params &P, const sos_calib &C, int w, int h, int n, int host,
                                          const ActImages &A, const sos_pair_tfm *__restrict__ pairs, float pu, float pv,
                                          float color, float weight, float energyTH, float slack, float idepth, ActState &S,
                                          float &Hdd, float &bd) {
  const int lane = threadIdx.x, g = lane >> 3, pix = lane & 7;
  const int nres = n - 1;
  const float wM3G = (float)(w - 3), hM3G = (float)(h - 3);
  float E = 0;
#pragma unroll
  for (int c = 0; c < ACT_MAXC; c++) {
    if (c * 8 >= nres) break;
    const int r = c * 8 + g;
    const bool valid = r < nres;
    const bool skip = !valid || S.st[c] == SOS_RES_OOB;  // FS/ImmaturePoint.cpp:479-482
    bool fail = false;
    float et = 0, ht = 0, bt = 0;
    if (!skip) {
      const int target = r < host ? r : r + 1;
      const sos_pair_tfm &T = pairs[host + n * target];
      const float KliP0 = (pu + c_pattern[pix][0] - C.cxl) * C.fxli;
      const float KliP1 = (pv + c_pattern[pix][1] - C.cyl) * C.fyli;
      const float ptp0 = T.R[0] * KliP0 + T.R[1] * KliP1 + T.R[2] + T.t[0] * idepth;
      const float ptp1 = T.R[3] * KliP0 + T.R[4] * KliP1 + T.R[5] + T.t[1] * idepth;
      const float ptp2 = T.R[6] * KliP0 + T.R[7] * KliP1 + T.R[8] + T.t[2] * idepth;
      const float drescale = 1.0f / ptp2;
      bool ok = drescale > 0;
      float u = 0, v = 0, Ku = 0, Kv = 0;
      if (ok) {
        u = ptp0 * drescale;
        v = ptp1 * drescale;
        Ku = u * C.fxl + C.cxl;
        Kv = v * C.fyl + C.cyl;
        ok = Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
      }
      if (ok) {
        float hit[3];
        interp33(A.dI[target], Ku, Kv, w, h, hit);
        if (isfinite(hit[0])) {
          const float residual = hit[0] - (T.aff[0] * color + T.aff[1]);
          float hw = fabsf(residual) < P.huberTH ? 1 : P.huberTH / fabsf(residual);
          et = weight * weight * hw * residual * residual * (2 - hw);
          const float dxInterp = hit[1] * C.fxl, dyInterp = hit[2] * C.fyl;
          const float d_idepth = (dxInterp * drescale * (T.t[0] - T.t[2] * u) + dyInterp * drescale * (T.t[1] - T.t[2] * v)) * SOS_SCALE_IDEPTH;
          hw *= weight * weight;
          ht = (hw * d_idepth) * d_idepth;
          bt = (hw * residual) * d_idepth;
        } else {
          fail = true;
        }
      } else {
        fail = true;
      }
    }
    const unsigned long long failB = __ballot(fail);
    const unsigned m8 = (unsigned)(failB >> (g * 8)) & 0xffu;
    const int first = m8 ? __ffs(m8) - 1 : 8;  // first pattern pixel at which the residual returns OOB
    unsigned long long cB = __ballot(!skip && pix < first);
    while (cB) {  // the running sums of FS/ImmaturePoint.cpp:531-532, residual-major, pattern order
      const int l = __ffsll((long long)cB) - 1;
      cB &= cB - 1;
      Hdd += lane_value(ht, l);
      bd += lane_value(bt, l);
    }
    // energyLeft of the group: 0 + e0 + e1 + ... + e7 in order
    float eg = et;
#pragma unroll
    for (int k = 1; k < 8; k++) {
      const float prev = __shfl_up(eg, 1, 8);
      if (pix == k) eg = prev + et;
    }
    eg = __shfl(eg, 7, 8);
    float ret;
    if (skip) {
      if (valid) S.nw[c] = SOS_RES_OOB;
      ret = S.stE[c];
    } else if (first < 8) {
      S.nw[c] = SOS_RES_OOB;
      ret = S.stE[c];
    } else {
      const float th = energyTH * slack;  // :535-541
      if (eg > th) {
        eg = th;
        S.nw[c] = SOS_RES_OUTLIER;
      } else {
        S.nw[c] = SOS_RES_IN;
      }
      S.nwE[c] = eg;
      ret = eg;
    }
#pragma unroll
    for (int gg = 0; gg < 8; gg++)
      if (c * 8 + gg < nres) E = E + lane_value(ret, gg * 8);
  }
  return E;
}

__global__ __launch_bounds__(64) void k_immature_activate(sos_activate_params P, sos_calib C, int w, int h, int n, ActImages A,
                                                          const sos_pair_tfm *__restrict__ pairs, int count,
                                                          const sos_immature *__restrict__ pts, const int *__restrict__ hostOf,
                                                          sos_activation *__restrict__ out) {
  const int i = blockIdx.x;
  if (i >= count) return;
  const int lane = threadIdx.x, g = lane >> 3, pix = lane & 7;
  const sos_immature &p = pts[i];
  const float pu = p.u, pv = p.v, color = p.color[pix], weight = p.weights[pix], energyTH = p.energyTH;
  const int host = hostOf[i];
  const int nres = n - 1;
  ActState S;
#pragma unroll
  for (int c = 0; c < ACT_MAXC; c++) {  // FS/FullSystemOptPoint.cpp:49-58
    S.st[c] = SOS_RES_IN;
    S.nw[c] = SOS_RES_OUTLIER;
    S.stE[c] = S.nwE[c] = 0;
  }
  float lastHdd = 0, lastbd = 0;
  float currentIdepth = (p.idepth_max + p.idepth_min) * 0.5f;
  float lastEnergy = act_eval(P, C, w, h, n, host, A, pairs, pu, pv, color, weight, energyTH, 1000.0f, currentIdepth, S, lastHdd, lastbd);
#pragma unroll
  for (int c = 0; c < ACT_MAXC; c++) {
    S.st[c] = S.nw[c];
    S.stE[c] = S.nwE[c];
  }
  int status = SOS_ACT_ACTIVATED, it = 0;
  if (!isfinite(lastEnergy) || lastHdd < P.minIdepthH_act) {  // :75-80
    status = SOS_ACT_SKIP;
  } else {
    float lambda = 0.1f;
    for (int iteration = 0; iteration < P.GNIts; iteration++) {  // :88-130
      it++;
      float H = lastHdd;
      H *= 1 + lambda;
      const float step = (float)((1.0 / (double)H) * (double)lastbd);
      const float newIdepth = currentIdepth - step;
      float newHdd = 0, newbd = 0;
      const float newEnergy = act_eval(P, C, w, h, n, host, A, pairs, pu, pv, color, weight, energyTH, 1.0f, newIdepth, S, newHdd, newbd);
      if (!isfinite(lastEnergy) || newHdd < P.minIdepthH_act) {  // :102-107
        status = SOS_ACT_SKIP;
        break;
      }
      if (newEnergy < lastEnergy) {
        currentIdepth = newIdepth;
        lastHdd = newHdd;
        lastbd = newbd;
        lastEnergy = newEnergy;
#pragma unroll
        for (int c = 0; c < ACT_MAXC; c++) {
          S.st[c] = S.nw[c];
          S.stE[c] = S.nwE[c];
        }
        lambda *= 0.5f;
      } else {
        lambda *= 5;
      }
      if ((double)fabsf(step) < 0.0001 * (double)currentIdepth) break;  // :128-129
    }
  }
  unsigned inMask = 0;
  int numGood = 0;
  if (status == SOS_ACT_ACTIVATED) {
    if (!isfinite(currentIdepth)) {  // :132-137
      status = SOS_ACT_DELETE;
    } else {
#pragma unroll
      for (int c = 0; c < ACT_MAXC; c++) {
        if (c * 8 >= nres) break;
        const int r = c * 8 + g;
        const unsigned long long b = __ballot(r < nres && pix == 0 && S.st[c] == SOS_RES_IN);
#pragma unroll
        for (int gg = 0; gg < 8; gg++)
          if ((b >> (gg * 8)) & 1ull) {
            const int rr = c * 8 + gg;
            inMask |= 1u << (rr < host ? rr : rr + 1);
            numGood++;
          }
      }
      if (numGood < P.minObs) status = SOS_ACT_DELETE;          // :144-149
      else if (!isfinite(energyTH)) status = SOS_ACT_DELETE;    // :151-155
    }
  }
  if (lane == 0) {
    sos_activation o;
    o.status = status;
    o.idepth = currentIdepth;
    o.inMask = inMask;
    o.energy = lastEnergy;
    o.Hdd = lastHdd;
    o.bd = lastbd;
    o.iterations = it;
    o.pad = 0;
    out[i] = o;
  }
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
  k_immature_trace<<<(count + TRACE_PPB - 1) / TRACE_PPB, 64, 0, c->stream>>>(*prm, c->dI[frameSlot][0], c->w, c->h, count,
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
  static const bool timing = getenv("SOS_TIMING") != nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, c->stream); }
  k_immature_trace_batch<<<(count + TRACE_PPB - 1) / TRACE_PPB, 64, 0, c->stream>>>(*prm, c->dI[frameSlot][0], c->w, c->h, count,
                                                                  reinterpret_cast<sos_immature *>(st->dev),
                                                                  reinterpret_cast<const int *>(st->dev + off_h),
                                                                  reinterpret_cast<const TraceArgs *>(st->dev + off_t));
  SOS_HIP(hipGetLastError());
  if (timing) hipEventRecord(e1, c->stream);
  SOS_HIP(hipStreamSynchronize(c->stream));
  if (timing) {
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "[sos] immature_trace_all: %d points, kernel %.1f us\n", count, ms * 1e3f);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  memcpy(pts, st->host, sizeof(sos_immature) * (size_t)count);
  return SOS_OK;
}

extern "C" int sos_immature_activate(sos_ctx *c, const sos_activate_params *prm, const sos_calib *calib, int nFrames,
                                     const int32_t *frameSlot, const sos_pair_tfm *pairs, int count, const sos_immature *pts,
                                     const int32_t *hostOfPoint, sos_activation *out) {
  if (!c || !prm || !calib || !frameSlot || !pairs || count < 0 || (count && (!pts || !hostOfPoint || !out))) return SOS_ERR_ARG;
  if (nFrames < 1 || nFrames > SOS_MAX_FRAMES) return SOS_ERR_ARG;
  ActImages A;
  memset(&A, 0, sizeof(A));
  for (int f = 0; f < nFrames; f++) {
    if (frameSlot[f] < 0 || frameSlot[f] >= SOS_MAX_SLOTS || !c->dI[frameSlot[f]][0]) return SOS_ERR_STATE;
    A.dI[f] = c->dI[frameSlot[f]][0];
  }
  for (int i = 0; i < count; i++)
    if (hostOfPoint[i] < 0 || hostOfPoint[i] >= nFrames) return SOS_ERR_ARG;
  if (count == 0) return SOS_OK;
  SOS_HIP(hipSetDevice(c->device));
  Stage *st;
  const size_t szP = sizeof(sos_immature) * (size_t)count, szH = (sizeof(int32_t) * (size_t)count + 127) / 128 * 128,
               szT = sizeof(sos_pair_tfm) * (size_t)nFrames * nFrames, szO = sizeof(sos_activation) * (size_t)count;
  int rc = stage_ensure(c->device, szP + szH + szT + szO, &st);
  if (rc) return rc;
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(st->host, pts, szP);
  memcpy(st->host + szP, hostOfPoint, sizeof(int32_t) * (size_t)count);
  memcpy(st->host + szP + szH, pairs, szT);
  k_immature_activate<<<count, 64, 0, c->stream>>>(*prm, *calib, c->w, c->h, nFrames, A,
                                                   reinterpret_cast<const sos_pair_tfm *>(st->dev + szP + szH), count,
                                                   reinterpret_cast<const sos_immature *>(st->dev),
                                                   reinterpret_cast<const int *>(st->dev + szP),
                                                   reinterpret_cast<sos_activation *>(st->dev + szP + szH + szT));
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(c->stream));
  memcpy(out, st->host + szP + szH + szT, szO);
  return SOS_OK;
}

// ------------------------------------------------------------------------------------------------
// device-resident immature sets
// ------------------------------------------------------------------------------------------------
struct sos_immset {
  sos_ctx *ctx = nullptr;
  struct Seg {
    int key = 0, count = 0;
    sos_immature *dev = nullptr;
    size_t cap = 0;
  };
  std::vector<Seg> segs;
  Seg *find(int key) {
    for (auto &g : segs)
      if (g.key == key) return &g;
    return nullptr;
  }
};

extern "C" int sos_immset_create(sos_ctx *c, sos_immset **out) {
  if (!c || !out) return SOS_ERR_ARG;
  sos_immset *s = new (std::nothrow) sos_immset();
  if (!s) return SOS_ERR_NOMEM;
  s->ctx = c;
  *out = s;
  return SOS_OK;
}
extern "C" void sos_immset_destroy(sos_immset *s) {
  if (!s) return;
  if (s->ctx) {
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
  }
  for (auto &g : s->segs)
    if (g.dev) (void)hipFree(g.dev);
  delete s;
}
extern "C" int sos_immset_put(sos_immset *s, int hostKey, int count, const sos_immature *pts) {
  if (!s || count < 0 || (count && !pts)) return SOS_ERR_ARG;
  sos_ctx *c = s->ctx;
  SOS_HIP(hipSetDevice(c->device));
  sos_immset::Seg *g = s->find(hostKey);
  if (count == 0) {
    if (g) {
      SOS_HIP(hipStreamSynchronize(c->stream));  // a trace of the old list may still be in flight
      if (g->dev) SOS_HIP(hipFree(g->dev));
      s->segs.erase(s->segs.begin() + (g - s->segs.data()));
    }
    return SOS_OK;
  }
  if (!g) {
    if ((int)s->segs.size() >= SOS_MAX_FRAMES) return SOS_ERR_STATE;
    s->segs.emplace_back();
    g = &s->segs.back();
    g->key = hostKey;
  }
  if ((size_t)count > g->cap) {
    SOS_HIP(hipStreamSynchronize(c->stream));
    if (g->dev) SOS_HIP(hipFree(g->dev));
    g->dev = nullptr;
    g->cap = 0;
    g->count = 0;
    const size_t want = (size_t)count + (size_t)count / 4 + 64;
    if (hipMalloc((void **)&g->dev, sizeof(sos_immature) * want) != hipSuccess) return SOS_ERR_NOMEM;
    g->cap = want;
  }
  SOS_HIP(hipMemcpyAsync(g->dev, pts, sizeof(sos_immature) * (size_t)count, hipMemcpyHostToDevice, c->stream));
  SOS_HIP(hipStreamSynchronize(c->stream));  // pts belongs to the caller again
  g->count = count;
  return SOS_OK;
}
extern "C" int sos_immset_count(sos_immset *s, int hostKey, int *count) {
  if (!s || !count) return SOS_ERR_ARG;
  sos_immset::Seg *g = s->find(hostKey);
  *count = g ? g->count : 0;
  return SOS_OK;
}
extern "C" int sos_immset_get(sos_immset *s, int hostKey, int count, sos_immature *out) {
  if (!s || count < 0 || (count && !out)) return SOS_ERR_ARG;
  sos_immset::Seg *g = s->find(hostKey);
  if ((g ? g->count : 0) != count) return SOS_ERR_STATE;
  if (count == 0) return SOS_OK;
  sos_ctx *c = s->ctx;
  SOS_HIP(hipSetDevice(c->device));
  SOS_HIP(hipMemcpyAsync(out, g->dev, sizeof(sos_immature) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
  SOS_HIP(hipStreamSynchronize(c->stream));
  return SOS_OK;
}
extern "C" int sos_immset_trace(sos_immset *s, const sos_trace_params *prm, int frameSlot, int nhosts, const int32_t *hostKeys,
                                const float *KRKi, const float *Kt, const float *aff) {
  if (!s || !prm || nhosts < 0 || nhosts > SOS_MAX_FRAMES || (nhosts && (!hostKeys || !KRKi || !Kt || !aff))) return SOS_ERR_ARG;
  sos_ctx *c = s->ctx;
  if (frameSlot < 0 || frameSlot >= SOS_MAX_SLOTS || !c->dI[frameSlot][0]) return SOS_ERR_STATE;
  for (int k = 0; k < nhosts; k++)
    for (int j = 0; j < k; j++)
      if (hostKeys[j] == hostKeys[k]) return SOS_ERR_ARG;
  ImmSegs S;
  memset(&S, 0, sizeof(S));
  int blocks = 0;
  for (int k = 0; k < nhosts; k++) {
    sos_immset::Seg *g = s->find(hostKeys[k]);
    if (!g || g->count == 0) continue;
    const int q = S.nseg++;
    S.ptr[q] = g->dev;
    S.count[q] = g->count;
    S.blockBegin[q] = blocks;
    blocks += (g->count + TRACE_PPB - 1) / TRACE_PPB;
    memcpy(S.A[q].K, KRKi + 9 * k, sizeof(S.A[q].K));
    memcpy(S.A[q].Kt, Kt + 3 * k, sizeof(S.A[q].Kt));
    memcpy(S.A[q].aff, aff + 2 * k, sizeof(S.A[q].aff));
  }
  if (S.nseg == 0) return SOS_OK;
  S.blockBegin[S.nseg] = blocks;
  SOS_HIP(hipSetDevice(c->device));
  k_immature_trace_sets<<<blocks, 64, 0, c->stream>>>(*prm, c->dI[frameSlot][0], c->w, c->h, S);
  SOS_HIP(hipGetLastError());
  return SOS_OK;
}
