// sos_ctx.hip -- device context, frame (image pyramid) store and the makeImages kernels (gfx950).
//
// Replaces FrameHessian::makeImages (FS/HessianBlocks.cpp:121-176): level 0 copy, 2x2 box mean for
// levels >= 1, central-difference gradients on the flat index range [w, w*(h-1)), absSquaredGrad with
// the optional gamma weighting.  The first and last rows of dx, dy and absSquaredGrad, which the
// reference leaves uninitialised, are zero here.
//
// Layout in HBM: per slot and level one AoS buffer (I, dx, dy) of wl*hl*3 floats, exactly the
// reference's Eigen::Vector3f array (FS/HessianBlocks.h:144-147), plus wl*hl floats of absSquaredGrad.
#include "sos_common.h"

static int pyr_levels(int w, int h) {  // util/globalCalib.cpp:39-47
  int wl = w, hl = h, lv = 1;
  while (wl % 2 == 0 && hl % 2 == 0 && wl * hl > 5000 && lv < SOS_PYR_LEVELS) {
    wl /= 2;
    hl /= 2;
    lv++;
  }
  return lv;
}

extern "C" const char *sos_backend_name(void) { return "hip-gfx950"; }

extern "C" int sos_ctx_create(int device, void *hip_stream, int w, int h, sos_ctx **out) {
  if (!out || w <= 0 || h <= 0) return SOS_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    fprintf(stderr, "[sos_slam_hip] no HIP device available (%s)\n", hipGetErrorString(e));
    return SOS_ERR_HIP;
  }
  if (device < 0 || device >= ndev) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(device));
  sos_ctx *c = new sos_ctx();
  c->device = device;
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    SOS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  c->w = w;
  c->h = h;
  c->levels = pyr_levels(w, h);
  for (int l = 0; l < c->levels; l++) {
    c->wl[l] = w >> l;
    c->hl[l] = h >> l;
  }
  memset(c->dI, 0, sizeof(c->dI));
  memset(c->absg, 0, sizeof(c->absg));
  memset(c->dIt, 0, sizeof(c->dIt));
  memset(c->has_pyr, 0, sizeof(c->has_pyr));
  SOS_HIP(hipMalloc(&c->d_img, sizeof(float) * (size_t)w * h));
  SOS_HIP(hipMalloc(&c->d_gammaB, sizeof(float) * 256));
  SOS_HIP(hipEventCreate(&c->ev0));
  SOS_HIP(hipEventCreate(&c->ev1));
  *out = c;
  return SOS_OK;
}

extern "C" int sos_ctx_destroy(sos_ctx *c) {
  if (!c) return SOS_OK;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  for (int s = 0; s < SOS_MAX_SLOTS; s++)
    for (int l = 0; l < SOS_PYR_LEVELS; l++) {
      if (c->dI[s][l]) hipFree(c->dI[s][l]);
      if (c->absg[s][l]) hipFree(c->absg[s][l]);
    }
  for (int s = 0; s < SOS_MAX_SLOTS; s++)
    if (c->dIt[s]) hipFree(c->dIt[s]);
  hipFree(c->d_img);
  hipFree(c->d_gammaB);
  hipEventDestroy(c->ev0);
  hipEventDestroy(c->ev1);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return SOS_OK;
}

extern "C" int sos_ctx_synchronize(sos_ctx *c) {
  if (!c) return SOS_ERR_ARG;
  SOS_HIP(hipStreamSynchronize(c->stream));
  return SOS_OK;
}

extern "C" int sos_ctx_pyr_levels(const sos_ctx *c) { return c ? c->levels : SOS_ERR_ARG; }

int sos_ctx_ensure_slot(sos_ctx *c, int slot, bool all_levels) {
  if (slot < 0 || slot >= SOS_MAX_SLOTS) return SOS_ERR_ARG;
  int nl = all_levels ? c->levels : 1;
  for (int l = 0; l < nl; l++) {
    size_t npx = (size_t)c->wl[l] * c->hl[l];
    if (!c->dI[slot][l]) SOS_HIP(hipMalloc(&c->dI[slot][l], sizeof(float) * 3 * npx));
    if (!c->absg[slot][l]) SOS_HIP(hipMalloc(&c->absg[slot][l], sizeof(float) * npx));
  }
  if (!c->dIt[slot]) SOS_HIP(hipMalloc(&c->dIt[slot], sizeof(float) * sos_tiled_floats(c->w, c->h)));
  return SOS_OK;
}

// ---------------------------------------------------------------------------------------------
// kernels.  One thread per pixel; a 256-thread block covers 256 consecutive pixels of a row-major
// level, so the (I,dx,dy) stores of a block form one contiguous 3 KiB span.
// ---------------------------------------------------------------------------------------------
__global__ void k_pyr_level0(const float *__restrict__ img, float *__restrict__ dI, int npx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npx) dI[3 * i] = img[i];
}

__global__ void k_pyr_down(const float *__restrict__ dIm, float *__restrict__ dI, int wl, int hl, int wlm1) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= wl * hl) return;
  int y = i / wl, x = i - y * wl;
  int b = 2 * x + 2 * y * wlm1;
  // FS/HessianBlocks.cpp:145-152, summed left to right
  dI[3 * i] = 0.25f * (dIm[3 * b] + dIm[3 * (b + 1)] + dIm[3 * (b + wlm1)] + dIm[3 * (b + 1 + wlm1)]);
}

__global__ void k_pyr_grad(float *__restrict__ dI, float *__restrict__ absg, int wl, int hl,
                           const float *__restrict__ gammaB) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int npx = wl * hl;
  if (idx >= npx) return;
  float dx = 0.f, dy = 0.f, ag = 0.f;
  if (idx >= wl && idx < wl * (hl - 1)) {  // FS/HessianBlocks.cpp:155-173
    dx = 0.5f * (dI[3 * (idx + 1)] - dI[3 * (idx - 1)]);
    dy = 0.5f * (dI[3 * (idx + wl)] - dI[3 * (idx - wl)]);
    if (!isfinite(dx)) dx = 0;
    if (!isfinite(dy)) dy = 0;
    ag = dx * dx + dy * dy;
    if (gammaB) {
      int c = (int)(dI[3 * idx] + 0.5f);
      if (c < 5) c = 5;
      if (c > 250) c = 250;
      float gw = gammaB[c + 1] - gammaB[c];
      ag *= gw * gw;
    }
  }
  dI[3 * idx + 1] = dx;
  dI[3 * idx + 2] = dy;
  absg[idx] = ag;
}

// level 0 -> tiled copy: one thread per float of the tiled buffer (coalesced stores; the loads of a block fall into
// two image rows)
__global__ void k_tile_level0(const float *__restrict__ dI, float *__restrict__ dIt, int w, int h, int tpr, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t tile = i / SOS_TLINE;
  const int f = (int)(i - tile * SOS_TLINE);
  const int ty = (int)(tile / tpr), tx = (int)(tile - (size_t)ty * tpr);
  float v = 0.f;
  if (f < 3 * SOS_TW * SOS_TH) {
    const int t = f / 3, ch = f - 3 * t;
    const int x = tx * SOS_TW + (t % SOS_TW), y = ty * SOS_TH + (t / SOS_TW);
    if (x < w && y < h) v = dI[3 * ((size_t)y * w + x) + ch];
  }
  dIt[i] = v;
}
static int launch_tile_level0(sos_ctx *c, int slot) {
  const size_t total = sos_tiled_floats(c->w, c->h);
  k_tile_level0<<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(c->dI[slot][0], c->dIt[slot], c->w, c->h,
                                                                          sos_tiles_per_row(c->w), total);
  return hipGetLastError() == hipSuccess ? SOS_OK : SOS_ERR_HIP;
}

// FrameHessian::makeImages from the float image already in c->d_img (sos_make_pyramid, sos_undistort_frame)
int sos_ctx_pyramid_from_staged(sos_ctx *c, int slot, const float *gammaB) {
  int rc = sos_ctx_ensure_slot(c, slot, true);
  if (rc) return rc;
  size_t npx0 = (size_t)c->w * c->h;
  if (gammaB) SOS_HIP(hipMemcpyAsync(c->d_gammaB, gammaB, sizeof(float) * 256, hipMemcpyHostToDevice, c->stream));
  const int B = 256;
  k_pyr_level0<<<(unsigned)((npx0 + B - 1) / B), B, 0, c->stream>>>(c->d_img, c->dI[slot][0], (int)npx0);
  for (int l = 0; l < c->levels; l++) {
    int npx = c->wl[l] * c->hl[l];
    if (l > 0)
      k_pyr_down<<<(npx + B - 1) / B, B, 0, c->stream>>>(c->dI[slot][l - 1], c->dI[slot][l], c->wl[l], c->hl[l],
                                                        c->wl[l - 1]);
    k_pyr_grad<<<(npx + B - 1) / B, B, 0, c->stream>>>(c->dI[slot][l], c->absg[slot][l], c->wl[l], c->hl[l],
                                                      gammaB ? c->d_gammaB : nullptr);
  }
  SOS_HIP(hipGetLastError());
  rc = launch_tile_level0(c, slot);
  if (rc) return rc;
  c->has_pyr[slot] = true;
  return SOS_OK;
}

extern "C" int sos_make_pyramid(sos_ctx *c, int slot, const float *img, const float *gammaB) {
  if (!c || !img) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(c->device));
  int rc = sos_ctx_ensure_slot(c, slot, true);
  if (rc) return rc;
  SOS_HIP(hipMemcpyAsync(c->d_img, img, sizeof(float) * (size_t)c->w * c->h, hipMemcpyHostToDevice, c->stream));
  rc = sos_ctx_pyramid_from_staged(c, slot, gammaB);
  if (rc) return rc;
  SOS_HIP(hipStreamSynchronize(c->stream));  // `img` is a caller-owned pageable buffer
  return SOS_OK;
}

extern "C" int sos_frame_upload_dI(sos_ctx *c, int slot, const float *dI) {
  if (!c || !dI) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(c->device));
  int rc = sos_ctx_ensure_slot(c, slot, false);
  if (rc) return rc;
  SOS_HIP(hipMemcpyAsync(c->dI[slot][0], dI, sizeof(float) * 3 * (size_t)c->w * c->h, hipMemcpyHostToDevice,
                         c->stream));
  rc = launch_tile_level0(c, slot);
  if (rc) return rc;
  SOS_HIP(hipStreamSynchronize(c->stream));
  return SOS_OK;
}

extern "C" int sos_frame_download_level(sos_ctx *c, int slot, int lvl, float *dI_out, float *absgrad_out) {
  if (!c || slot < 0 || slot >= SOS_MAX_SLOTS || lvl < 0 || lvl >= c->levels) return SOS_ERR_ARG;
  if (!c->dI[slot][lvl]) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(c->device));
  size_t npx = (size_t)c->wl[lvl] * c->hl[lvl];
  if (dI_out)
    SOS_HIP(hipMemcpyAsync(dI_out, c->dI[slot][lvl], sizeof(float) * 3 * npx, hipMemcpyDeviceToHost, c->stream));
  if (absgrad_out)
    SOS_HIP(hipMemcpyAsync(absgrad_out, c->absg[slot][lvl], sizeof(float) * npx, hipMemcpyDeviceToHost, c->stream));
  SOS_HIP(hipStreamSynchronize(c->stream));
  return SOS_OK;
}

extern "C" int sos_frame_release(sos_ctx *c, int slot) {
  if (!c || slot < 0 || slot >= SOS_MAX_SLOTS) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(c->device));
  SOS_HIP(hipStreamSynchronize(c->stream));
  for (int l = 0; l < SOS_PYR_LEVELS; l++) {
    if (c->dI[slot][l]) { hipFree(c->dI[slot][l]); c->dI[slot][l] = nullptr; }
    if (c->absg[slot][l]) { hipFree(c->absg[slot][l]); c->absg[slot][l] = nullptr; }
  }
  if (c->dIt[slot]) { hipFree(c->dIt[slot]); c->dIt[slot] = nullptr; }
  c->has_pyr[slot] = false;
  return SOS_OK;
}
