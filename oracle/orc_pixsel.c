/* oracle/orc_pixsel.c -- TEST INFRASTRUCTURE (CPU oracle, see oracle.h; PARITY UNPINNED).
 *
 * Restatement of PixelSelector (FS/PixelSelector2.cpp): makeHists :69-155, makeMaps :157-290, select :292-424 --
 * the candidate-pixel selection in front of the ImmaturePoint constructor (FullSystem::makeNewTraces,
 * FS/FullSystem.cpp:1071-1097).  The random pattern (glibc rand() & 0xFF after srand(3141592), :37-40) is an input.
 * One documented deviation: for image widths that are no multiple of 32 the reference indexes thsSmoothed with
 * (xf >> 5) == w/32, which for the last cell row lands one element behind the written part of the array (never
 * initialised there); here that slack reads as 0. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static int hist_quantil(const int *hist, float below) { /* computeHistQuantil, :59-67 */
  int th = hist[0] * below + 0.5f;
  for (int i = 0; i < 90; i++) {
    th -= (i + 1 < 50) ? hist[i + 1] : 0;
    if (th < 0) return i;
  }
  return 90;
}

void orc_pixsel_make_hists(const sos_pixsel_params *P, const float *absg0, int w, int h, float *ths, float *thsSmoothed) {
  int w32 = w / 32, h32 = h / 32;
  for (int y = 0; y < h32; y++)
    for (int x = 0; x < w32; x++) {
      const float *map0 = absg0 + 32 * x + 32 * y * w;
      int hist0[50];
      memset(hist0, 0, sizeof(hist0));
      for (int j = 0; j < 32; j++)
        for (int i = 0; i < 32; i++) {
          int it = i + 32 * x, jt = j + 32 * y;
          if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
          int g = sqrtf(map0[i + j * w]);
          if (g > 48) g = 48;
          hist0[g + 1]++;
          hist0[0]++;
        }
      ths[x + y * w32] = hist_quantil(hist0, P->minGradHistCut) + P->minGradHistAdd;
    }
  for (int y = 0; y < h32; y++)
    for (int x = 0; x < w32; x++) { /* :108-154, the nine-neighbour mean in the reference's order of additions */
      float sum = 0, num = 0;
      if (x > 0) {
        if (y > 0) { num++; sum += ths[x - 1 + (y - 1) * w32]; }
        if (y < h32 - 1) { num++; sum += ths[x - 1 + (y + 1) * w32]; }
        num++; sum += ths[x - 1 + y * w32];
      }
      if (x < w32 - 1) {
        if (y > 0) { num++; sum += ths[x + 1 + (y - 1) * w32]; }
        if (y < h32 - 1) { num++; sum += ths[x + 1 + (y + 1) * w32]; }
        num++; sum += ths[x + 1 + y * w32];
      }
      if (y > 0) { num++; sum += ths[x + (y - 1) * w32]; }
      if (y < h32 - 1) { num++; sum += ths[x + (y + 1) * w32]; }
      num++; sum += ths[x + y * w32];
      thsSmoothed[x + y * w32] = (sum / num) * (sum / num);
    }
}

static const float DIRS[16][2] = {{0, 1.0000f},       {0.3827f, 0.9239f},  {0.1951f, 0.9808f}, {0.9239f, 0.3827f},
                                  {0.7071f, 0.7071f}, {0.3827f, -0.9239f}, {0.8315f, 0.5556f}, {0.8315f, -0.5556f},
                                  {0.5556f, -0.8315f}, {0.9808f, 0.1951f}, {0.9239f, -0.3827f}, {0.7071f, -0.7071f},
                                  {0.5556f, 0.8315f}, {0.9808f, -0.1951f}, {1.0000f, 0.0000f}, {0.1951f, -0.9808f}};

static inline int imin(int a, int b) { return a < b ? a : b; }

/* select, :292-424.  dI = level-0 (I,dx,dy) AoS, absg0/1/2 = absSquaredGrad of levels 0..2, thsSm = thsSmoothed with
 * thsN = w32*h32 valid entries */
void orc_pixsel_select(const sos_pixsel_params *P, const float *dI, const float *absg0, const float *absg1, const float *absg2,
                       int w, int h, const uint8_t *randomPattern, const float *thsSm, int pot, float thFactor,
                       float *map_out, int32_t n_out[3]) {
  int w1 = w >> 1, w2 = w >> 2;
  int thsStep = w / 32, thsN = (w / 32) * (h / 32);
  memset(map_out, 0, sizeof(float) * (size_t)w * h);
  float dw1 = P->gradDownweightPerLevel, dw2 = dw1 * dw1;
  int n3 = 0, n2 = 0, n4 = 0;
  for (int y4 = 0; y4 < h; y4 += 4 * pot)
    for (int x4 = 0; x4 < w; x4 += 4 * pot) {
      int my3 = imin(4 * pot, h - y4), mx3 = imin(4 * pot, w - x4);
      int bestIdx4 = -1;
      float bestVal4 = 0;
      const float *dir4 = DIRS[randomPattern[n2] & 0xF];
      for (int y3 = 0; y3 < my3; y3 += 2 * pot)
        for (int x3 = 0; x3 < mx3; x3 += 2 * pot) {
          int x34 = x3 + x4, y34 = y3 + y4;
          int my2 = imin(2 * pot, h - y34), mx2 = imin(2 * pot, w - x34);
          int bestIdx3 = -1;
          float bestVal3 = 0;
          const float *dir3 = DIRS[randomPattern[n2] & 0xF];
          for (int y2 = 0; y2 < my2; y2 += pot)
            for (int x2 = 0; x2 < mx2; x2 += pot) {
              int x234 = x2 + x34, y234 = y2 + y34;
              int my1 = imin(pot, h - y234), mx1 = imin(pot, w - x234);
              int bestIdx2 = -1;
              float bestVal2 = 0;
              const float *dir2 = DIRS[randomPattern[n2] & 0xF];
              for (int y1 = 0; y1 < my1; y1++)
                for (int x1 = 0; x1 < mx1; x1++) {
                  int xf = x1 + x234, yf = y1 + y234;
                  int idx = xf + w * yf;
                  if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
                  int ti = (xf >> 5) + (yf >> 5) * thsStep;
                  float pixelTH0 = ti < thsN ? thsSm[ti] : 0.0f;
                  float pixelTH1 = pixelTH0 * dw1;
                  float pixelTH2 = pixelTH1 * dw2;
                  float ag0 = absg0[idx];
                  if (ag0 > pixelTH0 * thFactor) {
                    float dirNorm = fabsf((float)(dI[3 * idx + 1] * dir2[0] + dI[3 * idx + 2] * dir2[1]));
                    if (!P->selectDirectionDistribution) dirNorm = ag0;
                    if (dirNorm > bestVal2) { bestVal2 = dirNorm; bestIdx2 = idx; bestIdx3 = -2; bestIdx4 = -2; }
                  }
                  if (bestIdx3 == -2) continue;
                  float ag1 = absg1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1];
                  if (ag1 > pixelTH1 * thFactor) {
                    float dirNorm = fabsf((float)(dI[3 * idx + 1] * dir3[0] + dI[3 * idx + 2] * dir3[1]));
                    if (!P->selectDirectionDistribution) dirNorm = ag1;
                    if (dirNorm > bestVal3) { bestVal3 = dirNorm; bestIdx3 = idx; bestIdx4 = -2; }
                  }
                  if (bestIdx4 == -2) continue;
                  float ag2 = absg2[(int)(xf * 0.25f + 0.125) + (int)(yf * 0.25f + 0.125) * w2];
                  if (ag2 > pixelTH2 * thFactor) {
                    float dirNorm = fabsf((float)(dI[3 * idx + 1] * dir4[0] + dI[3 * idx + 2] * dir4[1]));
                    if (!P->selectDirectionDistribution) dirNorm = ag2;
                    if (dirNorm > bestVal4) { bestVal4 = dirNorm; bestIdx4 = idx; }
                  }
                }
              if (bestIdx2 > 0) { map_out[bestIdx2] = 1; bestVal3 = 1e10f; n2++; }
            }
          if (bestIdx3 > 0) { map_out[bestIdx3] = 2; bestVal4 = 1e10f; n3++; }
        }
      if (bestIdx4 > 0) { map_out[bestIdx4] = 4; n4++; }
    }
  n_out[0] = n2; n_out[1] = n3; n_out[2] = n4;
}

/* makeMaps, :157-290 (the FAST branch is commented out in the reference).  *pot is PixelSelector::currentPotential
 * (in / out); returns numHaveSub. */
int orc_pixsel_make_maps(const sos_pixsel_params *P, const float *dI, const float *absg0, const float *absg1, const float *absg2,
                         int w, int h, const uint8_t *randomPattern, const float *thsSm, float density, int recursionsLeft,
                         float thFactor, int *pot, float *map_out) {
  float numHave = 0, numWant = density, quotia;
  int idealPotential = *pot;
  int32_t n[3];
  orc_pixsel_select(P, dI, absg0, absg1, absg2, w, h, randomPattern, thsSm, *pot, thFactor, map_out, n);
  numHave = n[0] + n[1] + n[2];
  quotia = numWant / numHave;
  float K = numHave * (*pot + 1) * (*pot + 1);
  idealPotential = sqrtf(K / numWant) - 1;
  if (idealPotential < 1) idealPotential = 1;
  if (recursionsLeft > 0 && quotia > 1.25 && *pot > 1) {
    if (idealPotential >= *pot) idealPotential = *pot - 1;
    *pot = idealPotential;
    return orc_pixsel_make_maps(P, dI, absg0, absg1, absg2, w, h, randomPattern, thsSm, density, recursionsLeft - 1, thFactor, pot, map_out);
  } else if (recursionsLeft > 0 && quotia < 0.25) {
    if (idealPotential <= *pot) idealPotential = *pot + 1;
    *pot = idealPotential;
    return orc_pixsel_make_maps(P, dI, absg0, absg1, absg2, w, h, randomPattern, thsSm, density, recursionsLeft - 1, thFactor, pot, map_out);
  }
  int numHaveSub = numHave;
  if (quotia < 0.95) {
    int wh = w * h, rn = 0;
    unsigned char charTH = 255 * quotia;
    for (int i = 0; i < wh; i++) {
      if (map_out[i] != 0) {
        if (randomPattern[rn] > charTH) { map_out[i] = 0; numHaveSub--; }
        rn++;
      }
    }
  }
  *pot = idealPotential;
  return numHaveSub;
}
