/* oracle/orc_undistort.c -- TEST INFRASTRUCTURE (CPU oracle, see oracle.h; PARITY UNPINNED).
 *
 * Restatement of the image front-end, U/Undistort.cpp: getUndistorterForFile :240-351 and readFromFile :679-890 (camera
 * file, rectified K, remap table), makeOptimalK_crop :557-672, distortCoordinates of the five camera models
 * :902-1126, PhotometricUndistorter :38-161 / processFrame :194-227, Undistort::undistort :361-458 (without the
 * benchmark noise options, which default to off).  The reference's mixed float / double expressions are kept as written
 * (parameters are doubles read with %lf, narrowed to float at the top of distortCoordinates; literals like 2.0 and 0.5
 * promote their sub-expressions to double). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

int orc_camera_parse(const char *text, sos_camera_model *out) {
  char l[4][512];
  memset(l, 0, sizeof(l));
  memset(out, 0, sizeof(*out));
  const char *p = text;
  for (int k = 0; k < 4; k++) {
    int n = 0;
    while (*p && *p != '\n' && n < 511) l[k][n++] = *p++;
    while (n > 0 && (l[k][n - 1] == '\r')) n--;
    l[k][n] = 0;
    if (*p == '\n') p++;
  }
  float ic[10];
  double *q = out->pars;
  int nPars = 0;
  const char *prefix = "";
  /* model selection, :261-345 */
  if (sscanf(l[0], "%f %f %f %f %f %f %f %f", &ic[0], &ic[1], &ic[2], &ic[3], &ic[4], &ic[5], &ic[6], &ic[7]) == 8) { out->model = SOS_CAM_RADTAN; nPars = 8; }
  else if (sscanf(l[0], "%f %f %f %f %f", &ic[0], &ic[1], &ic[2], &ic[3], &ic[4]) == 5) { out->model = ic[4] == 0 ? SOS_CAM_PINHOLE : SOS_CAM_FOV; nPars = 5; }
  else if (sscanf(l[0], "KannalaBrandt %f %f %f %f %f %f %f %f", &ic[0], &ic[1], &ic[2], &ic[3], &ic[4], &ic[5], &ic[6], &ic[7]) == 8) { out->model = SOS_CAM_KB; nPars = 8; prefix = "KannalaBrandt "; }
  else if (sscanf(l[0], "RadTan %f %f %f %f %f %f %f %f", &ic[0], &ic[1], &ic[2], &ic[3], &ic[4], &ic[5], &ic[6], &ic[7]) == 8) { out->model = SOS_CAM_RADTAN; nPars = 8; prefix = "RadTan "; }
  else if (sscanf(l[0], "EquiDistant %f %f %f %f %f %f %f %f", &ic[0], &ic[1], &ic[2], &ic[3], &ic[4], &ic[5], &ic[6], &ic[7]) == 8) { out->model = SOS_CAM_EQUIDISTANT; nPars = 8; prefix = "EquiDistant "; }
  else if (sscanf(l[0], "FOV %f %f %f %f %f", &ic[0], &ic[1], &ic[2], &ic[3], &ic[4]) == 5) { out->model = SOS_CAM_FOV; nPars = 5; prefix = "FOV "; }
  else if (sscanf(l[0], "Pinhole %f %f %f %f %f", &ic[0], &ic[1], &ic[2], &ic[3], &ic[4]) == 5) { out->model = SOS_CAM_PINHOLE; nPars = 5; prefix = "Pinhole "; }
  else return -1;
  char buf[256];
  if (nPars == 5) { /* :705-723 */
    snprintf(buf, sizeof(buf), "%s%%lf %%lf %%lf %%lf %%lf", prefix);
    if (sscanf(l[0], buf, &q[0], &q[1], &q[2], &q[3], &q[4]) != 5) return -1;
  } else {
    snprintf(buf, sizeof(buf), "%s%%lf %%lf %%lf %%lf %%lf %%lf %%lf %%lf", prefix);
    if (sscanf(l[0], buf, &q[0], &q[1], &q[2], &q[3], &q[4], &q[5], &q[6], &q[7]) != 8) return -1;
  }
  if (sscanf(l[1], "%d %d", &out->wOrg, &out->hOrg) != 2) return -1;
  if (q[2] < 1 && q[3] < 1) { /* the "relative" format, :753-774 */
    q[0] = q[0] * out->wOrg;
    q[1] = q[1] * out->hOrg;
    q[2] = q[2] * out->wOrg - 0.5;
    q[3] = q[3] * out->hOrg - 0.5;
  }
  if (strcmp(l[2], "crop") == 0) out->rect = SOS_RECT_CROP; /* :777-796 */
  else if (strcmp(l[2], "full") == 0) return -1;            /* makeOptimalK_full is assert(false) in the reference */
  else if (strcmp(l[2], "none") == 0) out->rect = SOS_RECT_NONE;
  else if (sscanf(l[2], "%f %f %f %f %f", &out->outCal[0], &out->outCal[1], &out->outCal[2], &out->outCal[3], &out->outCal[4]) == 5) out->rect = SOS_RECT_GIVEN;
  else return -1;
  if (sscanf(l[3], "%d %d", &out->w, &out->h) != 2) return -1;
  return 0;
}

static void distort(const sos_camera_model *M, const double Kd[4], const float *in_x, const float *in_y, float *out_x, float *out_y, int n) {
  const double *P = M->pars;
  float fx = P[0], fy = P[1], cx = P[2], cy = P[3];
  float ofx = Kd[0], ofy = Kd[1], ocx = Kd[2], ocy = Kd[3];
  for (int i = 0; i < n; i++) {
    float x = in_x[i], y = in_y[i];
    float ix = (x - ocx) / ofx;
    float iy = (y - ocy) / ofy;
    switch (M->model) {
      case SOS_CAM_FOV: { /* :902-933 */
        float dist = P[4];
        float d2t = 2.0f * tanf(dist / 2.0f);
        float r = sqrtf(ix * ix + iy * iy);
        float fac = (r == 0 || dist == 0) ? 1 : atanf(r * d2t) / (dist * r);
        ix = fx * fac * ix + cx;
        iy = fy * fac * iy + cy;
        out_x[i] = ix; out_y[i] = iy;
      } break;
      case SOS_CAM_RADTAN: { /* :945-984 */
        float k1 = P[4], k2 = P[5], r1 = P[6], r2 = P[7];
        float mx2_u = ix * ix, my2_u = iy * iy, mxy_u = ix * iy;
        float rho2_u = mx2_u + my2_u;
        float rad_dist_u = k1 * rho2_u + k2 * rho2_u * rho2_u;
        float x_dist = ix + ix * rad_dist_u + 2.0 * r1 * mxy_u + r2 * (rho2_u + 2.0 * mx2_u);
        float y_dist = iy + iy * rad_dist_u + 2.0 * r2 * mxy_u + r1 * (rho2_u + 2.0 * my2_u);
        out_x[i] = fx * x_dist + cx; out_y[i] = fy * y_dist + cy;
      } break;
      case SOS_CAM_EQUIDISTANT: { /* :997-1037 */
        float k1 = P[4], k2 = P[5], k3 = P[6], k4 = P[7];
        float r = sqrtf(ix * ix + iy * iy);
        float theta = atanf(r);
        float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
        float thetad = theta * (1 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
        float scaling = (r > 1e-8) ? thetad / r : 1.0;
        out_x[i] = fx * ix * scaling + cx; out_y[i] = fy * iy * scaling + cy;
      } break;
      case SOS_CAM_KB: { /* :1049-1092 */
        float k0 = P[4], k1 = P[5], k2 = P[6], k3 = P[7];
        float Xsq_plus_Ysq = ix * ix + iy * iy;
        float sqrt_Xsq_Ysq = sqrtf(Xsq_plus_Ysq);
        float theta = atan2f(sqrt_Xsq_Ysq, 1);
        float theta2 = theta * theta, theta3 = theta2 * theta, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
        float r = theta + k0 * theta3 + k1 * theta5 + k2 * theta7 + k3 * theta9;
        if (sqrt_Xsq_Ysq < 1e-6) { out_x[i] = fx * ix + cx; out_y[i] = fy * iy + cy; }
        else { out_x[i] = (r / sqrt_Xsq_Ysq) * fx * ix + cx; out_y[i] = (r / sqrt_Xsq_Ysq) * fy * iy + cy; }
      } break;
      default: /* pinhole, :1102-1126 */
        out_x[i] = fx * ix + cx; out_y[i] = fy * iy + cy;
    }
  }
}

/* readFromFile from line 4 on: K (as 4 doubles fx fy cx cy, like the Mat33 of the reference), remap table, passthrough.
 * Returns 0, or -1 where the reference exits. */
int orc_undistort_setup(const sos_camera_model *M, double Kd[4], float *remapX, float *remapY, int *passthrough) {
  const int w = M->w, h = M->h, wOrg = M->wOrg, hOrg = M->hOrg;
  *passthrough = 0;
  Kd[0] = Kd[1] = 1; Kd[2] = Kd[3] = 0;
  if (M->rect == SOS_RECT_CROP) { /* makeOptimalK_crop, :557-672 */
    float *tgX = (float *)malloc(sizeof(float) * 100000), *tgY = (float *)malloc(sizeof(float) * 100000);
    float minX = 0, maxX = 0, minY = 0, maxY = 0;
    for (int x = 0; x < 100000; x++) { tgX[x] = (x - 50000.0f) / 10000.0f; tgY[x] = 0; }
    distort(M, Kd, tgX, tgY, tgX, tgY, 100000);
    for (int x = 0; x < 100000; x++)
      if (tgX[x] > 0 && tgX[x] < wOrg - 1) {
        if (minX == 0) minX = (x - 50000.0f) / 10000.0f;
        maxX = (x - 50000.0f) / 10000.0f;
      }
    for (int y = 0; y < 100000; y++) { tgY[y] = (y - 50000.0f) / 10000.0f; tgX[y] = 0; }
    distort(M, Kd, tgX, tgY, tgX, tgY, 100000);
    for (int y = 0; y < 100000; y++)
      if (tgY[y] > 0 && tgY[y] < hOrg - 1) {
        if (minY == 0) minY = (y - 50000.0f) / 10000.0f;
        maxY = (y - 50000.0f) / 10000.0f;
      }
    free(tgX); free(tgY);
    minX *= 1.01; maxX *= 1.01; minY *= 1.01; maxY *= 1.01;
    int oobLeft = 1, oobRight = 1, oobTop = 1, oobBottom = 1, iteration = 0;
    while (oobLeft || oobRight || oobTop || oobBottom) {
      oobLeft = oobRight = oobTop = oobBottom = 0;
      for (int y = 0; y < h; y++) {
        remapX[y * 2] = minX;
        remapX[y * 2 + 1] = maxX;
        remapY[y * 2] = remapY[y * 2 + 1] = minY + (maxY - minY) * (float)y / ((float)h - 1.0f);
      }
      distort(M, Kd, remapX, remapY, remapX, remapY, 2 * h);
      for (int y = 0; y < h; y++) {
        if (!(remapX[2 * y] > 0 && remapX[2 * y] < wOrg - 1)) oobLeft = 1;
        if (!(remapX[2 * y + 1] > 0 && remapX[2 * y + 1] < wOrg - 1)) oobRight = 1;
      }
      for (int x = 0; x < w; x++) {
        remapY[x * 2] = minY;
        remapY[x * 2 + 1] = maxY;
        remapX[x * 2] = remapX[x * 2 + 1] = minX + (maxX - minX) * (float)x / ((float)w - 1.0f);
      }
      distort(M, Kd, remapX, remapY, remapX, remapY, 2 * w);
      for (int x = 0; x < w; x++) {
        if (!(remapY[2 * x] > 0 && remapY[2 * x] < hOrg - 1)) oobTop = 1;
        if (!(remapY[2 * x + 1] > 0 && remapY[2 * x + 1] < hOrg - 1)) oobBottom = 1;
      }
      if ((oobLeft || oobRight) && (oobTop || oobBottom)) {
        if ((maxX - minX) > (maxY - minY)) oobBottom = oobTop = 0;
        else oobLeft = oobRight = 0;
      }
      if (oobLeft) minX *= 0.995;
      if (oobRight) maxX *= 0.995;
      if (oobTop) minY *= 0.995;
      if (oobBottom) maxY *= 0.995;
      if (++iteration > 500) return -1;
    }
    Kd[0] = ((float)w - 1.0f) / (maxX - minX);
    Kd[1] = ((float)h - 1.0f) / (maxY - minY);
    Kd[2] = (double)(-minX) * Kd[0]; /* -minX * K(0,0): float times double; K(0,0) holds a float value */
    Kd[3] = (double)(-minY) * Kd[1];
  } else if (M->rect == SOS_RECT_NONE) { /* :826-837 */
    if (w != wOrg || h != hOrg) return -1;
    Kd[0] = M->pars[0]; Kd[1] = M->pars[1]; Kd[2] = M->pars[2]; Kd[3] = M->pars[3];
    *passthrough = 1;
  } else { /* :838-852 */
    Kd[0] = M->outCal[0] * w;
    Kd[1] = M->outCal[1] * h;
    Kd[2] = M->outCal[2] * w - 0.5;
    Kd[3] = M->outCal[3] * h - 0.5;
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) { remapX[x + y * w] = x; remapY[x + y * w] = y; }
  distort(M, Kd, remapX, remapY, remapX, remapY, h * w);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) { /* :867-885, with its two slips (ix assigned in the iy branch; iy tested against wOrg) */
      float ix = remapX[x + y * w], iy = remapY[x + y * w];
      if (ix == 0) ix = 0.001;
      if (iy == 0) iy = 0.001;
      if (ix == wOrg - 1) ix = wOrg - 1.001;
      if (iy == hOrg - 1) ix = hOrg - 1.001;
      /* deviation: the reference's test of iy against wOrg lets rows behind the input image through for landscape
       * images (it then reads out of bounds); those pixels are invalid here */
      if (ix > 0 && iy > 0 && ix < wOrg - 1 && iy < wOrg - 1 && iy < hOrg - 1) { remapX[x + y * w] = ix; remapY[x + y * w] = iy; }
      else { remapX[x + y * w] = -1; remapY[x + y * w] = -1; }
    }
  return 0;
}

/* PhotometricUndistorter constructor: G normalised in place (GDepth entries), vignetteInv from the raw vignette values.
 * Returns 1 when the calibration is valid. */
int orc_photometric_setup(float *G, int GDepth, const float *vignette, int n, int photometricMode, float *vignetteInv) {
  if (!G || GDepth < 256 || !vignette) return 0;
  for (int i = 0; i < GDepth - 1; i++)
    if (G[i + 1] <= G[i]) return 0;
  float min = G[0], max = G[GDepth - 1];
  for (int i = 0; i < GDepth; i++) G[i] = 255.0 * (G[i] - min) / (max - min);
  if (photometricMode == 0)
    for (int i = 0; i < GDepth; i++) G[i] = 255.0f * i / (float)(GDepth - 1);
  float maxV = 0;
  for (int i = 0; i < n; i++)
    if (vignette[i] > maxV) maxV = vignette[i];
  for (int i = 0; i < n; i++) vignetteInv[i] = 1.0f / (vignette[i] / maxV);
  return 1;
}

/* processFrame + undistort for one raw image (bpp 1 or 2); out = w*h floats */
void orc_undistort_frame(const sos_camera_model *M, const float *remapX, const float *remapY, int passthrough, const float *G, int valid,
                         const float *vignetteInv, int photometricMode, const void *raw, int bpp, float exposure, float factor,
                         float *out) {
  const int w = M->w, h = M->h, wOrg = M->wOrg, hOrg = M->hOrg, wh = wOrg * hOrg;
  float *data = (float *)malloc(sizeof(float) * (size_t)wh);
  const unsigned char *r8 = (const unsigned char *)raw;
  const unsigned short *r16 = (const unsigned short *)raw;
  if (!valid || exposure <= 0 || photometricMode == 0) { /* :203-209 */
    for (int i = 0; i < wh; i++) data[i] = factor * (bpp == 1 ? r8[i] : r16[i]);
  } else {
    for (int i = 0; i < wh; i++) data[i] = G[bpp == 1 ? r8[i] : r16[i]];
    if (photometricMode == 2)
      for (int i = 0; i < wh; i++) data[i] *= vignetteInv[i];
  }
  if (!passthrough) {
    for (int idx = w * h - 1; idx >= 0; idx--) { /* :399-448 */
      float xx = remapX[idx], yy = remapY[idx];
      if (xx < 0) out[idx] = 0;
      else {
        int xxi = xx, yyi = yy;
        xx -= xxi; yy -= yyi;
        float xxyy = xx * yy;
        const float *src = data + xxi + yyi * wOrg;
        out[idx] = xxyy * src[1 + wOrg] + (yy - xxyy) * src[wOrg] + (xx - xxyy) * src[1] + (1 - xx - yy + xxyy) * src[0];
      }
    }
  } else {
    memcpy(out, data, sizeof(float) * (size_t)w * h);
  }
  free(data);
}
