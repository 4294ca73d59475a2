// sos_undistort.hip -- image front-end (U/Undistort.cpp): the DSO camera file, the rectified camera matrix and remap
// table (host, once per camera: the transcendental functions of the distortion models come from libm exactly as in the
// reference), and per frame the photometric correction + bilinear remap on the device, written straight into the
// context's staging image and followed by the pyramid kernels -- a raw 8/16-bit frame is the only thing that crosses
// PCIe.
//   sos_camera_parse      getUndistorterForFile :240-351, readFromFile :679-800
//   CameraSetup           readFromFile :800-890, makeOptimalK_crop :557-672, distortCoordinates :902-1126
//   k_undistort<T>        PhotometricUndistorter::processFrame :194-227 fused into Undistort::undistort :361-458
// The reference's mixed float / double expressions are kept (parameters are doubles, narrowed to float per call; the
// literals 2.0 and 0.5 promote their sub-expressions).  benchmark_varNoise / varBlurNoise (off by default) are not
// implemented.
#include "sos_common.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct sos_undistort {
  sos_ctx *ctx = nullptr;
  sos_camera_model cam;
  double K[4] = {1, 1, 0, 0};
  bool passthrough = false, photoValid = false;
  int photometricMode = 2, GDepth = 0;
  std::vector<float> remapX, remapY;
  float *d_remapX = nullptr, *d_remapY = nullptr, *d_G = nullptr, *d_vinv = nullptr;
  void *d_raw = nullptr;
  size_t raw_cap = 0;
};

namespace {
// one line of the camera file against "<prefix>%lf ..." ; returns the number of parameters read
int scan_pars(const std::string &line, const char *prefix, int want, double *q) {
  std::string fmt = prefix;
  for (int i = 0; i < (want == 5 ? 5 : 10); i++) fmt += i ? " %lf" : "%lf";
  double t[10];
  const int got = std::sscanf(line.c_str(), fmt.c_str(), &t[0], &t[1], &t[2], &t[3], &t[4], &t[5], &t[6], &t[7], &t[8], &t[9]);
  for (int i = 0; i < want && i < got; i++) q[i] = t[i];
  return got;
}
bool scan_floats(const std::string &line, const char *prefix, int want) {
  std::string fmt = prefix;
  for (int i = 0; i < want; i++) fmt += i ? " %f" : "%f";
  float t[8];
  return std::sscanf(line.c_str(), fmt.c_str(), &t[0], &t[1], &t[2], &t[3], &t[4], &t[5], &t[6], &t[7]) == want;
}

// distortCoordinates of the five models over arrays (in place allowed)
struct Distorter {
  int model;
  float fx, fy, cx, cy, d[4], ofx, ofy, ocx, ocy;
  Distorter(const sos_camera_model &m, const double K[4]) : model(m.model) {
    fx = (float)m.pars[0]; fy = (float)m.pars[1]; cx = (float)m.pars[2]; cy = (float)m.pars[3];
    for (int i = 0; i < 4; i++) d[i] = (float)m.pars[4 + i];
    ofx = (float)K[0]; ofy = (float)K[1]; ocx = (float)K[2]; ocy = (float)K[3];
  }
  void operator()(const float *inx, const float *iny, float *outx, float *outy, int n) const {
    for (int i = 0; i < n; i++) {
      const float ix = (inx[i] - ocx) / ofx, iy = (iny[i] - ocy) / ofy;
      float ox, oy;
      if (model == SOS_CAM_RADTAN) {  // :945-984
        const float k1 = d[0], k2 = d[1], r1 = d[2], r2 = d[3];
        const float mx2 = ix * ix, my2 = iy * iy, mxy = ix * iy, rho2 = mx2 + my2;
        const float rad = k1 * rho2 + k2 * rho2 * rho2;
        const float xd = (float)(ix + ix * rad + 2.0 * r1 * mxy + r2 * (rho2 + 2.0 * mx2));
        const float yd = (float)(iy + iy * rad + 2.0 * r2 * mxy + r1 * (rho2 + 2.0 * my2));
        ox = fx * xd + cx;
        oy = fy * yd + cy;
      } else if (model == SOS_CAM_EQUIDISTANT) {  // :997-1037
        const float r = sqrtf(ix * ix + iy * iy);
        const float th = atanf(r), th2 = th * th, th4 = th2 * th2, th6 = th4 * th2, th8 = th4 * th4;
        const float thd = th * (1 + d[0] * th2 + d[1] * th4 + d[2] * th6 + d[3] * th8);
        const float scaling = (float)((r > 1e-8) ? thd / r : 1.0);
        ox = fx * ix * scaling + cx;
        oy = fy * iy * scaling + cy;
      } else if (model == SOS_CAM_KB) {  // :1049-1092
        const float s2 = ix * ix + iy * iy, s = sqrtf(s2);
        const float th = atan2f(s, 1), th2 = th * th, th3 = th2 * th, th5 = th3 * th2, th7 = th5 * th2, th9 = th7 * th2;
        const float r = th + d[0] * th3 + d[1] * th5 + d[2] * th7 + d[3] * th9;
        if (s < 1e-6) { ox = fx * ix + cx; oy = fy * iy + cy; }
        else { ox = (r / s) * fx * ix + cx; oy = (r / s) * fy * iy + cy; }
      } else if (model == SOS_CAM_FOV) {  // :902-933
        const float dist = d[0], d2t = 2.0f * tanf(dist / 2.0f);
        const float r = sqrtf(ix * ix + iy * iy);
        const float fac = (r == 0 || dist == 0) ? 1 : atanf(r * d2t) / (dist * r);
        ox = fx * fac * ix + cx;
        oy = fy * fac * iy + cy;
      } else {  // pinhole :1102-1126
        ox = fx * ix + cx;
        oy = fy * iy + cy;
      }
      outx[i] = ox;
      outy[i] = oy;
    }
  }
};

// K and the remap table; false where the reference exits
bool camera_setup(const sos_camera_model &m, double K[4], std::vector<float> &rx, std::vector<float> &ry, bool &passthrough) {
  const int w = m.w, h = m.h, wOrg = m.wOrg, hOrg = m.hOrg;
  rx.assign((size_t)w * h, 0.f);
  ry.assign((size_t)w * h, 0.f);
  passthrough = false;
  K[0] = K[1] = 1; K[2] = K[3] = 0;
  if (m.rect == SOS_RECT_CROP) {  // makeOptimalK_crop
    const int N = 100000;
    std::vector<float> a(N), b(N);
    float lim[4] = {0, 0, 0, 0};  // minX maxX minY maxY
    for (int axis = 0; axis < 2; axis++) {  // stretch the centre lines as far as they stay inside the input image
      for (int i = 0; i < N; i++) { a[i] = (i - 50000.0f) / 10000.0f; b[i] = 0; }
      const Distorter D(m, K);
      if (axis == 0) D(a.data(), b.data(), a.data(), b.data(), N);
      else D(b.data(), a.data(), b.data(), a.data(), N);
      const int org = axis == 0 ? wOrg : hOrg;
      for (int i = 0; i < N; i++)
        if (a[i] > 0 && a[i] < org - 1) {
          if (lim[2 * axis] == 0) lim[2 * axis] = (i - 50000.0f) / 10000.0f;
          lim[2 * axis + 1] = (i - 50000.0f) / 10000.0f;
        }
    }
    for (float &v : lim) v = (float)(v * 1.01);
    float &minX = lim[0], &maxX = lim[1], &minY = lim[2], &maxY = lim[3];
    bool oL = true, oR = true, oT = true, oB = true;
    int it = 0;
    while (oL || oR || oT || oB) {  // shrink the side that still maps outside, the wider dimension first
      oL = oR = oT = oB = false;
      const Distorter D(m, K);
      for (int y = 0; y < h; y++) {
        rx[2 * y] = minX;
        rx[2 * y + 1] = maxX;
        ry[2 * y] = ry[2 * y + 1] = minY + (maxY - minY) * (float)y / ((float)h - 1.0f);
      }
      D(rx.data(), ry.data(), rx.data(), ry.data(), 2 * h);
      for (int y = 0; y < h; y++) {
        if (!(rx[2 * y] > 0 && rx[2 * y] < wOrg - 1)) oL = true;
        if (!(rx[2 * y + 1] > 0 && rx[2 * y + 1] < wOrg - 1)) oR = true;
      }
      for (int x = 0; x < w; x++) {
        ry[2 * x] = minY;
        ry[2 * x + 1] = maxY;
        rx[2 * x] = rx[2 * x + 1] = minX + (maxX - minX) * (float)x / ((float)w - 1.0f);
      }
      D(rx.data(), ry.data(), rx.data(), ry.data(), 2 * w);
      for (int x = 0; x < w; x++) {
        if (!(ry[2 * x] > 0 && ry[2 * x] < hOrg - 1)) oT = true;
        if (!(ry[2 * x + 1] > 0 && ry[2 * x + 1] < hOrg - 1)) oB = true;
      }
      if ((oL || oR) && (oT || oB)) {
        if ((maxX - minX) > (maxY - minY)) oB = oT = false;
        else oL = oR = false;
      }
      if (oL) minX = (float)(minX * 0.995);
      if (oR) maxX = (float)(maxX * 0.995);
      if (oT) minY = (float)(minY * 0.995);
      if (oB) maxY = (float)(maxY * 0.995);
      if (++it > 500) return false;
    }
    K[0] = ((float)w - 1.0f) / (maxX - minX);
    K[1] = ((float)h - 1.0f) / (maxY - minY);
    K[2] = (double)(-minX) * K[0];
    K[3] = (double)(-minY) * K[1];
  } else if (m.rect == SOS_RECT_NONE) {
    if (w != wOrg || h != hOrg) return false;
    for (int i = 0; i < 4; i++) K[i] = m.pars[i];
    passthrough = true;
  } else {
    K[0] = m.outCal[0] * w;
    K[1] = m.outCal[1] * h;
    K[2] = m.outCal[2] * w - 0.5;
    K[3] = m.outCal[3] * h - 0.5;
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) { rx[x + y * w] = (float)x; ry[x + y * w] = (float)y; }
  const Distorter D(m, K);
  D(rx.data(), ry.data(), rx.data(), ry.data(), w * h);
  for (size_t i = 0; i < (size_t)w * h; i++) {  // :867-885 as written, including its two slips
    float ix = rx[i], iy = ry[i];
    if (ix == 0) ix = (float)0.001;
    if (iy == 0) iy = (float)0.001;
    if (ix == wOrg - 1) ix = (float)(wOrg - 1.001);
    if (iy == hOrg - 1) ix = (float)(hOrg - 1.001);
    // (iy is tested against wOrg in the reference; rows behind a landscape input image would be read out of bounds
    // there and are invalid here)
    const bool inside = ix > 0 && iy > 0 && ix < wOrg - 1 && iy < wOrg - 1 && iy < hOrg - 1;
    rx[i] = inside ? ix : -1.f;
    ry[i] = inside ? iy : -1.f;
  }
  return true;
}

// photometric correction of one raw pixel (processFrame): mode < 0 = plain factor * value
template <typename T>
__device__ __forceinline__ float photo(const T *__restrict__ raw, int i, const float *__restrict__ G, const float *__restrict__ vinv, int mode,
                                       float factor) {
  if (mode < 0) return factor * (float)raw[i];
  float v = G[raw[i]];
  if (mode == 2) v *= vinv[i];
  return v;
}
template <typename T>
__global__ void k_undistort(const T *__restrict__ raw, int wOrg, int npx, const float *__restrict__ remapX, const float *__restrict__ remapY,
                            const float *__restrict__ G, const float *__restrict__ vinv, int mode, float factor, int passthrough,
                            float *__restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npx) return;
  if (passthrough) {
    out[idx] = photo(raw, idx, G, vinv, mode, factor);
    return;
  }
  float xx = remapX[idx], yy = remapY[idx];
  if (xx < 0) {
    out[idx] = 0;
    return;
  }
  const int xxi = (int)xx, yyi = (int)yy;
  xx -= xxi;
  yy -= yyi;
  const float xxyy = xx * yy;
  const int s = xxi + yyi * wOrg;
  out[idx] = xxyy * photo(raw, s + 1 + wOrg, G, vinv, mode, factor) + (yy - xxyy) * photo(raw, s + wOrg, G, vinv, mode, factor) +
             (xx - xxyy) * photo(raw, s + 1, G, vinv, mode, factor) + (1 - xx - yy + xxyy) * photo(raw, s, G, vinv, mode, factor);
}
}  // namespace

extern "C" int sos_camera_parse(const char *text, sos_camera_model *out) {
  if (!text || !out) return SOS_ERR_ARG;
  memset(out, 0, sizeof(*out));
  std::string line[4];
  {
    const char *p = text;
    for (int k = 0; k < 4; k++) {
      while (*p && *p != '\n') line[k] += *p++;
      while (!line[k].empty() && line[k].back() == '\r') line[k].pop_back();
      if (*p == '\n') p++;
    }
  }
  // model selection in the reference's order of attempts (:261-345)
  const char *prefix = "";
  int nPars = 0;
  if (scan_floats(line[0], "", 8)) { out->model = SOS_CAM_RADTAN; nPars = 8; }
  else if (scan_floats(line[0], "", 5)) {
    double t[5];
    scan_pars(line[0], "", 5, t);
    float last = 0;
    std::sscanf(line[0].c_str(), "%*f %*f %*f %*f %f", &last);
    out->model = last == 0 ? SOS_CAM_PINHOLE : SOS_CAM_FOV;
    nPars = 5;
  } else if (scan_floats(line[0], "KannalaBrandt ", 8)) { out->model = SOS_CAM_KB; nPars = 8; prefix = "KannalaBrandt "; }
  else if (scan_floats(line[0], "RadTan ", 8)) { out->model = SOS_CAM_RADTAN; nPars = 8; prefix = "RadTan "; }
  else if (scan_floats(line[0], "EquiDistant ", 8)) { out->model = SOS_CAM_EQUIDISTANT; nPars = 8; prefix = "EquiDistant "; }
  else if (scan_floats(line[0], "FOV ", 5)) { out->model = SOS_CAM_FOV; nPars = 5; prefix = "FOV "; }
  else if (scan_floats(line[0], "Pinhole ", 5)) { out->model = SOS_CAM_PINHOLE; nPars = 5; prefix = "Pinhole "; }
  else return SOS_ERR_ARG;
  if (scan_pars(line[0], prefix, nPars, out->pars) != nPars) return SOS_ERR_ARG;
  if (std::sscanf(line[1].c_str(), "%d %d", &out->wOrg, &out->hOrg) != 2) return SOS_ERR_ARG;
  double *q = out->pars;
  if (q[2] < 1 && q[3] < 1) {  // "relative" calibration: scale by the image size, shift by half a pixel (:753-774)
    q[0] *= out->wOrg;
    q[1] *= out->hOrg;
    q[2] = q[2] * out->wOrg - 0.5;
    q[3] = q[3] * out->hOrg - 0.5;
  }
  if (line[2] == "crop") out->rect = SOS_RECT_CROP;
  else if (line[2] == "none") out->rect = SOS_RECT_NONE;
  else if (line[2] == "full") return SOS_ERR_ARG;  // makeOptimalK_full is assert(false)
  else if (std::sscanf(line[2].c_str(), "%f %f %f %f %f", &out->outCal[0], &out->outCal[1], &out->outCal[2], &out->outCal[3], &out->outCal[4]) == 5)
    out->rect = SOS_RECT_GIVEN;
  else return SOS_ERR_ARG;
  if (std::sscanf(line[3].c_str(), "%d %d", &out->w, &out->h) != 2) return SOS_ERR_ARG;
  return SOS_OK;
}

extern "C" int sos_undistort_destroy(sos_undistort *u) {
  if (!u) return SOS_OK;
  hipSetDevice(u->ctx->device);
  hipStreamSynchronize(u->ctx->stream);
  hipFree(u->d_remapX); hipFree(u->d_remapY); hipFree(u->d_G); hipFree(u->d_vinv); hipFree(u->d_raw);
  delete u;
  return SOS_OK;
}

extern "C" int sos_undistort_create(sos_ctx *c, const sos_camera_model *cam, const float *G, int GDepth, const float *vignette,
                                    int photometricMode, sos_undistort **out) {
  if (!c || !cam || !out || cam->w != c->w || cam->h != c->h || cam->wOrg < 2 || cam->hOrg < 2) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(c->device));
  sos_undistort *u = new sos_undistort();
  u->ctx = c;
  u->cam = *cam;
  u->photometricMode = photometricMode;
  if (!camera_setup(*cam, u->K, u->remapX, u->remapY, u->passthrough)) {
    delete u;
    return SOS_ERR_ARG;
  }
  const size_t npx = (size_t)cam->w * cam->h, nOrg = (size_t)cam->wOrg * cam->hOrg;
  // PhotometricUndistorter constructor, :38-161
  std::vector<float> g, vinv;
  if (G && vignette && GDepth >= 256) {
    bool inc = true;
    for (int i = 0; i + 1 < GDepth; i++) inc = inc && G[i + 1] > G[i];
    if (inc) {
      g.assign(G, G + GDepth);
      const float mn = g[0], mx = g[GDepth - 1];
      for (int i = 0; i < GDepth; i++) g[i] = (float)(255.0 * (g[i] - mn) / (mx - mn));
      if (photometricMode == 0)
        for (int i = 0; i < GDepth; i++) g[i] = 255.0f * i / (float)(GDepth - 1);
      float maxV = 0;
      for (size_t i = 0; i < nOrg; i++) maxV = vignette[i] > maxV ? vignette[i] : maxV;
      vinv.resize(nOrg);
      for (size_t i = 0; i < nOrg; i++) vinv[i] = 1.0f / (vignette[i] / maxV);
      u->photoValid = true;
      u->GDepth = GDepth;
    }
  }
  bool ok = hipMalloc(&u->d_remapX, sizeof(float) * npx) == hipSuccess && hipMalloc(&u->d_remapY, sizeof(float) * npx) == hipSuccess;
  if (ok && u->photoValid)
    ok = hipMalloc(&u->d_G, sizeof(float) * GDepth) == hipSuccess && hipMalloc(&u->d_vinv, sizeof(float) * nOrg) == hipSuccess;
  if (!ok) {
    sos_undistort_destroy(u);
    return SOS_ERR_NOMEM;
  }
  SOS_HIP(hipMemcpyAsync(u->d_remapX, u->remapX.data(), sizeof(float) * npx, hipMemcpyHostToDevice, c->stream));
  SOS_HIP(hipMemcpyAsync(u->d_remapY, u->remapY.data(), sizeof(float) * npx, hipMemcpyHostToDevice, c->stream));
  if (u->photoValid) {
    SOS_HIP(hipMemcpyAsync(u->d_G, g.data(), sizeof(float) * GDepth, hipMemcpyHostToDevice, c->stream));
    SOS_HIP(hipMemcpyAsync(u->d_vinv, vinv.data(), sizeof(float) * nOrg, hipMemcpyHostToDevice, c->stream));
  }
  SOS_HIP(hipStreamSynchronize(c->stream));
  *out = u;
  return SOS_OK;
}

extern "C" int sos_undistort_get(sos_undistort *u, float K[4], float *remapX, float *remapY, int32_t *passthrough) {
  if (!u) return SOS_ERR_ARG;
  if (K)
    for (int i = 0; i < 4; i++) K[i] = (float)u->K[i];
  if (remapX) memcpy(remapX, u->remapX.data(), sizeof(float) * u->remapX.size());
  if (remapY) memcpy(remapY, u->remapY.data(), sizeof(float) * u->remapY.size());
  if (passthrough) *passthrough = u->passthrough ? 1 : 0;
  return SOS_OK;
}

extern "C" int sos_undistort_frame(sos_undistort *u, const void *raw, int bytesPerPixel, float exposure, float factor, int slot,
                                   const float *gammaBgrad, float *image_out) {
  if (!u || !raw || (bytesPerPixel != 1 && bytesPerPixel != 2)) return SOS_ERR_ARG;
  sos_ctx *c = u->ctx;
  SOS_HIP(hipSetDevice(c->device));
  const size_t nOrg = (size_t)u->cam.wOrg * u->cam.hOrg, bytes = nOrg * bytesPerPixel;
  if (bytesPerPixel == 1 && u->photoValid && u->GDepth < 256) return SOS_ERR_STATE;
  if (bytesPerPixel == 2 && u->photoValid && u->GDepth < 65536 && exposure > 0 && u->photometricMode != 0) return SOS_ERR_STATE;
  if (bytes > u->raw_cap) {
    hipFree(u->d_raw);
    u->d_raw = nullptr;
    u->raw_cap = 0;
    if (hipMalloc(&u->d_raw, bytes) != hipSuccess) return SOS_ERR_NOMEM;
    u->raw_cap = bytes;
  }
  int rc = sos_ctx_ensure_slot(c, slot, true);
  if (rc) return rc;
  SOS_HIP(hipMemcpyAsync(u->d_raw, raw, bytes, hipMemcpyHostToDevice, c->stream));
  // processFrame: full photometric calibration only when it is valid, the exposure is known and the mode asks for it
  const int mode = (!u->photoValid || exposure <= 0 || u->photometricMode == 0) ? -1 : u->photometricMode;
  const int npx = c->w * c->h;
  if (bytesPerPixel == 1)
    k_undistort<unsigned char><<<(npx + 255) / 256, 256, 0, c->stream>>>(static_cast<const unsigned char *>(u->d_raw), u->cam.wOrg, npx, u->d_remapX,
                                                                         u->d_remapY, u->d_G, u->d_vinv, mode, factor, u->passthrough ? 1 : 0, c->d_img);
  else
    k_undistort<unsigned short><<<(npx + 255) / 256, 256, 0, c->stream>>>(static_cast<const unsigned short *>(u->d_raw), u->cam.wOrg, npx, u->d_remapX,
                                                                          u->d_remapY, u->d_G, u->d_vinv, mode, factor, u->passthrough ? 1 : 0, c->d_img);
  SOS_HIP(hipGetLastError());
  rc = sos_ctx_pyramid_from_staged(c, slot, gammaBgrad);
  if (rc) return rc;
  if (image_out) SOS_HIP(hipMemcpyAsync(image_out, c->d_img, sizeof(float) * (size_t)npx, hipMemcpyDeviceToHost, c->stream));
  SOS_HIP(hipStreamSynchronize(c->stream));
  return SOS_OK;
}
