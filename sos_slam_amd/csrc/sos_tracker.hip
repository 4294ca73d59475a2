// sos_tracker.hip -- device side of CoarseTracker / ScaleOptimizer for gfx950 / MI355X.
//
//   set_ref      ScaleOptimizer::makeK + CoarseTracker::makeCoarseDepthL0   FS/ScaleOptimizer.cpp:95-118,
//                                                                           FS/CoarseTracker.cpp:56-230
//   calc_res     CoarseTracker::calcResPose / ScaleOptimizer::calcResScale  FS/CoarseTracker.cpp:612-764,
//                                                                           FS/ScaleOptimizer.cpp:273-437
//   calc_gs      calcGSSSEPose / calcGSSSEScale                             FS/CoarseTracker.cpp:554-610,
//                                                                           FS/ScaleOptimizer.cpp:232-271
//
// Design notes (DESIGN.md "tracker"):
//   * the splat of the <= P points is combined per pixel on the host in point order (float sums of
//     colliding points keep the reference's order), the device scatters the unique pixels and does the
//     pyramid sums, the dilation (race-free: reads cells with bak > 0, writes cells with bak <= 0),
//     the normalisation and an ORDER-PRESERVING compaction (raster order, as the reference's loop).
//   * calc_res does not compact its survivors (the reference's compaction is an SSE artefact): it
//     writes the 8 warp values of every template pixel in place with weight 0 for dropped pixels, so
//     calc_gs sums exact zeros for them.  n = survivors rounded up to a multiple of 4 as in the
//     reference (FS/CoarseTracker.cpp:736-747) is kept for the 1/n normalisation.
//   * all reductions are fixed-shape trees: results are run-to-run deterministic.
#include "sos_common.h"

#include <chrono>

#include <algorithm>
#include <cmath>
#include <unordered_map>
#include <vector>

struct sos_tracker {
  sos_ctx *ctx = nullptr;
  sos_params prm;
  int levels = 1;
  int w[SOS_PYR_LEVELS], h[SOS_PYR_LEVELS];
  float fx[SOS_PYR_LEVELS], fy[SOS_PYR_LEVELS], cx[SOS_PYR_LEVELS], cy[SOS_PYR_LEVELS];
  float *idepth[SOS_PYR_LEVELS], *wsum[SOS_PYR_LEVELS], *wbak[SOS_PYR_LEVELS];
  float *pc_u[SOS_PYR_LEVELS], *pc_v[SOS_PYR_LEVELS], *pc_idepth[SOS_PYR_LEVELS], *pc_color[SOS_PYR_LEVELS];
  int pc_n[SOS_PYR_LEVELS];
  float *buf[8];  // idepth|rx1, u|rx2, v|rx3, dx, dy, residual, weight, refColor
  int buf_count = 0, buf_n = 0, buf_lvl = -1;
  int *d_counts = nullptr;   // per-block counts / offsets
  float *d_part = nullptr;   // per-block partial sums
  double *d_out = nullptr;   // final reduced values
  int *d_pix = nullptr;      // splat staging
  float *d_pixv = nullptr;
  int maxblk = 0;
  bool have_ref = false;
  // results come back through device-mapped pinned memory (no copy command per call): [0,8) calcRes sums,
  // [8,53) calcGSSSE sums
  double *pin_o = nullptr, *pin_o_dev = nullptr;  // [60] doubles as the completion flag (int)
  int seq = 0;
  float *d_part2 = nullptr;  // per-block partials of the speculative calcGSSSE
  int *d_fusectr = nullptr;  // arrival counter of the in-kernel final sums (zero between launches)
  // Speculation: the LM loops call calcRes and, when the step is accepted (the common case), calcGSSSE on the same
  // buffers.  With a hint for b0 (sos_tracker_set_gs_hint) calcRes also runs calcGSSSE behind itself, and the
  // following sos_tracker_calc_gs with the same (lvl, a, b0) is answered from the host copy: one round trip, not two.
  bool hint_on = false;
  float hint_b0 = 0;
  bool gs_cached = false;
  int gs_lvl = -1;
  float gs_a = 0, gs_b0 = 0;
  bool gss_cached = false;  // scale variant: (lvl, t, K1, scale)
  // loop-closure aligner (PoseEstimator, src/LoopClosure/PoseEstimator.cpp): 3-D points with one colour per level
  bool loop_mode = false;
  float *l_xyz[3] = {nullptr, nullptr, nullptr}, *l_col[SOS_PYR_LEVELS] = {nullptr};
  int l_n = 0, l_cap = 0;
  float gss_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // device-resident LM loop (sos_tracker_lm.inc): granule buffers, mapped result block, launch counter
  unsigned long long *lm_part = nullptr;
  double *lm_out = nullptr, *lm_out_dev = nullptr;
  int lm_seq = 0;
  unsigned lm_spin_limit = 1u << 18;  // LM_SPIN_LIMIT of sos_tracker_lm.inc
};

static inline int divup(int a, int b) { return (a + b - 1) / b; }

extern "C" int sos_tracker_create(sos_ctx *ctx, const sos_params *prm, sos_tracker **out) {
  if (!ctx || !prm || !out) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(ctx->device));
  sos_tracker *T = new sos_tracker();
  T->ctx = ctx;
  T->prm = *prm;
  T->levels = ctx->levels;
  for (int l = 0; l < T->levels; l++) {
    T->w[l] = ctx->wl[l];
    T->h[l] = ctx->hl[l];
    size_t n = (size_t)T->w[l] * T->h[l];
    SOS_HIP(hipMalloc(&T->idepth[l], sizeof(float) * n));
    SOS_HIP(hipMalloc(&T->wsum[l], sizeof(float) * n));
    SOS_HIP(hipMalloc(&T->wbak[l], sizeof(float) * n));
    SOS_HIP(hipMalloc(&T->pc_u[l], sizeof(float) * n));
    SOS_HIP(hipMalloc(&T->pc_v[l], sizeof(float) * n));
    SOS_HIP(hipMalloc(&T->pc_idepth[l], sizeof(float) * n));
    SOS_HIP(hipMalloc(&T->pc_color[l], sizeof(float) * n));
    T->pc_n[l] = 0;
  }
  size_t n0 = (size_t)ctx->w * ctx->h;
  for (int k = 0; k < 8; k++) SOS_HIP(hipMalloc(&T->buf[k], sizeof(float) * (n0 + 4)));
  T->maxblk = divup((int)n0, 256) + 1;
  SOS_HIP(hipMalloc(&T->d_counts, sizeof(int) * 2 * (T->maxblk + 8)));  // all levels side by side
  SOS_HIP(hipMalloc(&T->d_part, sizeof(float) * 48 * T->maxblk));
  SOS_HIP(hipMalloc(&T->d_out, sizeof(double) * 64));
  SOS_HIP(hipMalloc(&T->d_part2, sizeof(float) * 48 * T->maxblk));
  SOS_HIP(hipMalloc(&T->d_fusectr, sizeof(int) * 4));
  SOS_HIP(hipMemset(T->d_fusectr, 0, sizeof(int) * 4));
  SOS_HIP(hipHostMalloc((void **)&T->pin_o, sizeof(double) * 64, hipHostMallocMapped));
  SOS_HIP(hipHostGetDevicePointer((void **)&T->pin_o_dev, T->pin_o, 0));
  memset(T->pin_o, 0, sizeof(double) * 64);
  SOS_HIP(hipMalloc(&T->d_pix, sizeof(int) * n0));
  SOS_HIP(hipMalloc(&T->d_pixv, sizeof(float) * 2 * n0));
  *out = T;
  return SOS_OK;
}

extern "C" int sos_tracker_destroy(sos_tracker *T) {
  if (!T) return SOS_OK;
  hipSetDevice(T->ctx->device);
  hipStreamSynchronize(T->ctx->stream);
  for (int l = 0; l < T->levels; l++) {
    hipFree(T->idepth[l]); hipFree(T->wsum[l]); hipFree(T->wbak[l]);
    hipFree(T->pc_u[l]); hipFree(T->pc_v[l]); hipFree(T->pc_idepth[l]); hipFree(T->pc_color[l]);
  }
  for (int k = 0; k < 8; k++) hipFree(T->buf[k]);
  for (int k = 0; k < 3; k++) hipFree(T->l_xyz[k]);
  for (int l = 0; l < SOS_PYR_LEVELS; l++) hipFree(T->l_col[l]);
  hipFree(T->d_part2);
  hipFree(T->d_fusectr);
  hipFree(T->lm_part);
  if (T->lm_out) hipHostFree(T->lm_out);
  if (T->pin_o) hipHostFree(T->pin_o);
  hipFree(T->d_counts); hipFree(T->d_part); hipFree(T->d_out); hipFree(T->d_pix); hipFree(T->d_pixv);
  delete T;
  return SOS_OK;
}

// ------------------------------------------------------------------------------------------------
// makeCoarseDepthL0 kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_splat(const int *__restrict__ pix, const float *__restrict__ val, int n, float *__restrict__ idepth,
                        float *__restrict__ wsum) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idepth[pix[i]] = val[2 * i];
  wsum[pix[i]] = val[2 * i + 1];
}
__global__ void k_depth_down(const float *__restrict__ im, const float *__restrict__ wm, float *__restrict__ il,
                             float *__restrict__ wl_, int wl, int hl, int wlm1) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= wl * hl) return;
  int y = i / wl, x = i - y * wl;
  int b = 2 * x + 2 * y * wlm1;
  il[i] = im[b] + im[b + 1] + im[b + wlm1] + im[b + wlm1 + 1];  // FS/CoarseTracker.cpp:93-99
  wl_[i] = wm[b] + wm[b + 1] + wm[b + wlm1] + wm[b + wlm1 + 1];
}
__global__ void k_dilate(float *__restrict__ idepth, float *__restrict__ wsum, const float *__restrict__ bak, int wl,
                         int hl, int diag) {  // FS/CoarseTracker.cpp:105-190
  int i = blockIdx.x * blockDim.x + threadIdx.x + wl;
  if (i >= wl * hl - wl) return;
  if (bak[i] > 0) return;
  const int o0 = diag ? 1 + wl : 1, o1 = diag ? -1 - wl : -1, o2 = diag ? wl - 1 : wl, o3 = diag ? -wl + 1 : -wl;
  // The reference's loop touches index -1 (first cell, diagonal pattern) and index w*h (last cell): one element
  // outside the map.  Both cells lie in the 2-pixel border that never reaches the template, so the taps
  // are treated as empty here instead of reading outside the allocation.
  const int last = wl * hl - 1;
  float sum = 0, num = 0, numn = 0;
  if (i + o0 <= last && bak[i + o0] > 0) { sum += idepth[i + o0]; num += bak[i + o0]; numn++; }
  if (i + o1 >= 0 && bak[i + o1] > 0) { sum += idepth[i + o1]; num += bak[i + o1]; numn++; }
  if (bak[i + o2] > 0) { sum += idepth[i + o2]; num += bak[i + o2]; numn++; }
  if (bak[i + o3] > 0) { sum += idepth[i + o3]; num += bak[i + o3]; numn++; }
  if (numn > 0) {
    idepth[i] = sum / numn;
    wsum[i] = num / numn;
  }
}
// normalise + classify (FS/CoarseTracker.cpp:193-229); pass 0 counts per block, pass 1 writes compacted
__global__ __launch_bounds__(256) void k_normalize(float *__restrict__ idepth, float *__restrict__ wsum,
                                                   const float *__restrict__ dIref, int wl, int hl, int pass,
                                                   int *__restrict__ counts, float *__restrict__ pu, float *__restrict__ pv,
                                                   float *__restrict__ pid, float *__restrict__ pcol) {
  __shared__ int swave[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool keep = false;
  float idv = 0, col = 0;
  int x = 0, y = 0;
  if (i < wl * hl) {
    y = i / wl;
    x = i - y * wl;
    if (x >= 2 && x < wl - 2 && y >= 2 && y < hl - 2) {
      const float ws = wsum[i];
      if (ws > 0) {
        idv = pass == 0 ? idepth[i] / ws : idepth[i];
        col = dIref[3 * i];
        keep = isfinite(col) && (idv > 0);
        if (pass == 0) {
          idepth[i] = keep ? idv : -1.f;
          if (keep) wsum[i] = 1.f;
        }
      } else if (pass == 0) {
        idepth[i] = -1.f;
        wsum[i] = 1.f;
      }
    }
  }
  // pass 1 re-derives `keep` from the maps written by pass 0: kept cells have idepth > 0 and wsum == 1
  if (pass == 1) {
    keep = false;
    if (i < wl * hl && x >= 2 && x < wl - 2 && y >= 2 && y < hl - 2) {
      idv = idepth[i];
      col = dIref[3 * i];
      keep = (idv > 0) && isfinite(col);
    }
  }
  const unsigned long long m = __ballot(keep);
  const int rank = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) swave[wave] = __popcll(m);
  __syncthreads();
  int base = 0;
  for (int k = 0; k < wave; k++) base += swave[k];
  if (pass == 0) {
    if (threadIdx.x == 0) counts[blockIdx.x] = swave[0] + swave[1] + swave[2] + swave[3];
  } else if (keep) {
    const int o = counts[blockIdx.x] + base + rank;
    pu[o] = (float)x;
    pv[o] = (float)y;
    pid[o] = idv;
    pcol[o] = col;
  }
}
// exclusive scan of the per-block counts in place (order-preserving compaction offsets); counts[n] = total, which
// also goes to `total_out` (device-mapped host memory).  One block of 1024 threads, chunks of 1024 with a carry.
__global__ __launch_bounds__(1024) void k_scan_counts(int *counts, int n, int *total_out) {
  __shared__ int sm[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? counts[i] : 0;
    sm[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
      const int t = tid >= o ? sm[tid - o] : 0;
      __syncthreads();
      sm[tid] += t;
      __syncthreads();
    }
    const int c0 = carry;
    if (i < n) counts[i] = c0 + sm[tid] - v;  // exclusive
    __syncthreads();
    if (tid == 1023) carry = c0 + sm[1023];
    __syncthreads();
  }
  if (tid == 0) {
    counts[n] = carry;
    if (total_out) *total_out = carry;
  }
}
__global__ void k_scale_depth(float *__restrict__ pid, int n, float scale) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pid[i] /= scale;
}

extern "C" int sos_tracker_set_ref(sos_tracker *T, const sos_calib *calib, int refSlot, int npts, const float *u,
                                   const float *v, const float *idepth, const float *hdi, int32_t *pc_n_out) {
  if (!T || !calib || npts < 0 || (npts && (!u || !v || !idepth || !hdi))) return SOS_ERR_ARG;
  sos_ctx *c = T->ctx;
  if (refSlot < 0 || refSlot >= SOS_MAX_SLOTS || !c->has_pyr[refSlot]) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  T->loop_mode = false;
  // makeK, FS/ScaleOptimizer.cpp:95-118
  T->fx[0] = calib->fxl; T->fy[0] = calib->fyl; T->cx[0] = calib->cxl; T->cy[0] = calib->cyl;
  for (int l = 1; l < T->levels; l++) {
    T->fx[l] = T->fx[l - 1] * 0.5;
    T->fy[l] = T->fy[l - 1] * 0.5;
    T->cx[l] = (T->cx[0] + 0.5) / ((int)1 << l) - 0.5;
    T->cy[l] = (T->cy[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  // per-pixel sums in point order (FS/CoarseTracker.cpp:62-79)
  const int w0 = T->w[0], h0 = T->h[0];
  std::vector<int> pix;
  std::vector<float> val;
  std::unordered_map<int, int> where;
  pix.reserve(npts);
  val.reserve(2 * (size_t)npts);
  for (int i = 0; i < npts; i++) {
    int ui = u[i] + 0.5f;
    int vi = v[i] + 0.5f;
    if (ui < 0 || ui >= w0 || vi < 0 || vi >= h0) return SOS_ERR_ARG;
    float new_idepth = idepth[i];
    float weight = sqrtf(1e-3 / (hdi[i] + 1e-12));
    int key = ui + w0 * vi;
    auto it = where.find(key);
    if (it == where.end()) {
      where[key] = (int)pix.size();
      pix.push_back(key);
      val.push_back(0.0f + new_idepth * weight);
      val.push_back(0.0f + weight);
    } else {
      val[2 * (size_t)it->second] += new_idepth * weight;
      val[2 * (size_t)it->second + 1] += weight;
    }
  }
  const size_t n0 = (size_t)w0 * h0;
  SOS_HIP(hipMemsetAsync(T->idepth[0], 0, sizeof(float) * n0, st));
  SOS_HIP(hipMemsetAsync(T->wsum[0], 0, sizeof(float) * n0, st));
  const int nu = (int)pix.size();
  if (nu > 0) {
    SOS_HIP(hipMemcpyAsync(T->d_pix, pix.data(), sizeof(int) * nu, hipMemcpyHostToDevice, st));
    SOS_HIP(hipMemcpyAsync(T->d_pixv, val.data(), sizeof(float) * 2 * nu, hipMemcpyHostToDevice, st));
    k_splat<<<divup(nu, 256), 256, 0, st>>>(T->d_pix, T->d_pixv, nu, T->idepth[0], T->wsum[0]);
  }
  for (int l = 1; l < T->levels; l++) {
    const int npx = T->w[l] * T->h[l];
    k_depth_down<<<divup(npx, 256), 256, 0, st>>>(T->idepth[l - 1], T->wsum[l - 1], T->idepth[l], T->wsum[l], T->w[l],
                                                  T->h[l], T->w[l - 1]);
  }
  for (int l = 0; l < T->levels; l++) {
    const int npx = T->w[l] * T->h[l];
    SOS_HIP(hipMemcpyAsync(T->wbak[l], T->wsum[l], sizeof(float) * npx, hipMemcpyDeviceToDevice, st));
    const int span = npx - 2 * T->w[l];
    if (span > 0) k_dilate<<<divup(span, 256), 256, 0, st>>>(T->idepth[l], T->wsum[l], T->wbak[l], T->w[l], T->h[l], l < 2);
  }
  {
    int *tot_dev = reinterpret_cast<int *>(T->pin_o_dev + 61);  // 6 ints in the tail of the mapped result block
    int off = 0;
    for (int l = 0; l < T->levels; l++) {  // every level has its own slice of the count buffer: one sync for all levels
      const int npx = T->w[l] * T->h[l];
      const int nb = divup(npx, 256);
      const float *ref = c->dI[refSlot][l];
      int *cnt = T->d_counts + off;
      k_normalize<<<nb, 256, 0, st>>>(T->idepth[l], T->wsum[l], ref, T->w[l], T->h[l], 0, cnt, nullptr, nullptr, nullptr, nullptr);
      k_scan_counts<<<1, 1024, 0, st>>>(cnt, nb, tot_dev + l);
      k_normalize<<<nb, 256, 0, st>>>(T->idepth[l], T->wsum[l], ref, T->w[l], T->h[l], 1, cnt, T->pc_u[l], T->pc_v[l],
                                      T->pc_idepth[l], T->pc_color[l]);
      off += nb + 1;
    }
    SOS_HIP(hipStreamSynchronize(st));
    const int *tot = reinterpret_cast<const int *>(T->pin_o + 61);
    for (int l = 0; l < T->levels; l++) T->pc_n[l] = tot[l];
  }
  SOS_HIP(hipGetLastError());
  if (pc_n_out)
    for (int l = 0; l < T->levels; l++) pc_n_out[l] = T->pc_n[l];
  T->have_ref = true;
  T->buf_lvl = -1;
  return SOS_OK;
}

// PoseEstimator::makeK + pts = matched_frame->pts_dso (src/LoopClosure/PoseEstimator.cpp:129-148, 296-297): the tracker
// object becomes the loop-closure aligner -- its template is n 3-D points of the matched keyframe, each with one
// reference colour per pyramid level (colors[l * n + i]).  calc_res / calc_gs then run PoseEstimator::calcRes / calcGSSSE.
extern "C" int sos_tracker_set_points3d(sos_tracker *T, const sos_calib *calib, int n, const float *xyz, const float *colors) {
  if (!T || !calib || n < 0 || (n && (!xyz || !colors))) return SOS_ERR_ARG;
  sos_ctx *c = T->ctx;
  if ((size_t)n > (size_t)T->w[0] * T->h[0]) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  T->fx[0] = calib->fxl; T->fy[0] = calib->fyl; T->cx[0] = calib->cxl; T->cy[0] = calib->cyl;
  for (int l = 1; l < T->levels; l++) {
    T->fx[l] = T->fx[l - 1] * 0.5;
    T->fy[l] = T->fy[l - 1] * 0.5;
    T->cx[l] = (T->cx[0] + 0.5) / ((int)1 << l) - 0.5;
    T->cy[l] = (T->cy[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  if (n > T->l_cap) {
    for (int k = 0; k < 3; k++) { hipFree(T->l_xyz[k]); T->l_xyz[k] = nullptr; }
    for (int l = 0; l < SOS_PYR_LEVELS; l++) { hipFree(T->l_col[l]); T->l_col[l] = nullptr; }
    const int cap = n + n / 4 + 64;
    for (int k = 0; k < 3; k++) SOS_HIP(hipMalloc(&T->l_xyz[k], sizeof(float) * cap));
    for (int l = 0; l < T->levels; l++) SOS_HIP(hipMalloc(&T->l_col[l], sizeof(float) * cap));
    T->l_cap = cap;
  }
  std::vector<float> soa((size_t)3 * n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) soa[(size_t)k * n + i] = xyz[3 * (size_t)i + k];
  for (int k = 0; k < 3 && n; k++) SOS_HIP(hipMemcpyAsync(T->l_xyz[k], soa.data() + (size_t)k * n, sizeof(float) * n, hipMemcpyHostToDevice, st));
  for (int l = 0; l < T->levels && n; l++)
    SOS_HIP(hipMemcpyAsync(T->l_col[l], colors + (size_t)l * n, sizeof(float) * n, hipMemcpyHostToDevice, st));
  SOS_HIP(hipStreamSynchronize(st));
  for (int l = 0; l < T->levels; l++) T->pc_n[l] = n;
  T->l_n = n;
  T->loop_mode = true;
  T->have_ref = true;
  T->gs_cached = T->gss_cached = false;
  T->buf_lvl = -1;
  return SOS_OK;
}

extern "C" int sos_tracker_scale_depth(sos_tracker *T, float scale) {  // FS/CoarseTracker.cpp:244-251
  if (!T || !T->have_ref) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(T->ctx->device));
  for (int l = 0; l < T->levels; l++)
    if (T->pc_n[l] > 0) k_scale_depth<<<divup(T->pc_n[l], 256), 256, 0, T->ctx->stream>>>(T->pc_idepth[l], T->pc_n[l], scale);
  SOS_HIP(hipGetLastError());
  return SOS_OK;
}

extern "C" int sos_tracker_get_pc(sos_tracker *T, int lvl, float *pu, float *pv, float *pid, float *pcol) {
  if (!T || !T->have_ref || lvl < 0 || lvl >= T->levels) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(T->ctx->device));
  hipStream_t st = T->ctx->stream;
  const size_t n = (size_t)T->pc_n[lvl];
  if (n) {
    if (pu) SOS_HIP(hipMemcpyAsync(pu, T->pc_u[lvl], sizeof(float) * n, hipMemcpyDeviceToHost, st));
    if (pv) SOS_HIP(hipMemcpyAsync(pv, T->pc_v[lvl], sizeof(float) * n, hipMemcpyDeviceToHost, st));
    if (pid) SOS_HIP(hipMemcpyAsync(pid, T->pc_idepth[lvl], sizeof(float) * n, hipMemcpyDeviceToHost, st));
    if (pcol) SOS_HIP(hipMemcpyAsync(pcol, T->pc_color[lvl], sizeof(float) * n, hipMemcpyDeviceToHost, st));
  }
  SOS_HIP(hipStreamSynchronize(st));
  return SOS_OK;
}

// ------------------------------------------------------------------------------------------------
// calcResPose / calcResScale: one thread per template pixel
// ------------------------------------------------------------------------------------------------
struct ResArgs {
  const float *pu, *pv, *pid, *pcol;
  const float *dINew;
  float *buf[8];
  int n, lvl, wl, hl;
  float M[9], RKi[9], KiS[9], t[3];
  float fxl, fyl, cxl, cyl;
  float aff0, aff1, huber, cutoff, maxEnergy;
};

// Sum over the 64 lanes of a wave with DPP adds (VALU rate; the shuffle form is one LDS-crossbar round trip per step,
// 6 x NV dependent ds_bpermute per wave, about 2 us per LM step with the 53 sums of a residual launch): inclusive scan inside
// each 16-lane row (row_shr 1, 2, 4, 8), then lane 15 of a row is added into the next row (row_bcast:15 on rows 1 and
// 3, row_bcast:31 on rows 2 and 3).  Fixed order; lane 63 holds the sum.
#define SOS_DPP_ADD(v, ctrl, rmask, bound) \
  (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, (bound)))
__device__ __forceinline__ float wave_sum63(float a) {
  SOS_DPP_ADD(a, 0x111, 0xf, true);   // row_shr:1
  SOS_DPP_ADD(a, 0x112, 0xf, true);   // row_shr:2
  SOS_DPP_ADD(a, 0x114, 0xf, true);   // row_shr:4
  SOS_DPP_ADD(a, 0x118, 0xf, true);   // row_shr:8
  SOS_DPP_ADD(a, 0x142, 0xa, false);  // row_bcast:15 -> rows 1, 3
  SOS_DPP_ADD(a, 0x143, 0xc, false);  // row_bcast:31 -> rows 2, 3
  return a;
}
// block-level fixed-tree sum of NV floats held per thread; result in sm[0..NV) of thread 0's view
template <int NV>
__device__ __forceinline__ float block_sum_val(float *v, float *sm /* NV*4 */) {  // thread k < NV returns sum k
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = wave_sum63(v[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < NV; k++) sm[k * 4 + wave] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < NV) return (sm[threadIdx.x * 4] + sm[threadIdx.x * 4 + 1]) + (sm[threadIdx.x * 4 + 2] + sm[threadIdx.x * 4 + 3]);
  return 0.f;
}
template <int NV>
__device__ __forceinline__ void block_sum(float *v, float *sm /* NV*4 */, float *out) {
  const float r = block_sum_val<NV>(v, sm);
  if (threadIdx.x < NV) out[threadIdx.x] = r;
}

// parameters of the calcGSSSE that may ride inside k_calc_res (speculation, see sos_tracker)
struct GsFuse {
  float fxl, fyl, a, b0;   // pose variant: fx, fy of the level, affLL[0], b0
  float s, tx, ty, tz;     // scale variant
};
// MODE 0: CoarseTracker::calcRes, 1: ScaleOptimizer::calcResScale, 2: PoseEstimator::calcRes (3-D points x y z in pu pv pid)
// Final sums inside the producing kernel: the block whose arrival completes the grid adds the per-block partials in block
// order (exactly what k_sum_parts / k_sum_parts2 do in a second launch) and publishes the result to the polling host.
// One release fence per block (wave 0 wrote the partials), one acquire in the last block.
struct FuseSum {
  int *ctr;         // arrival counter, zero between launches (nullptr: the partials are summed by a second kernel)
  double *o1, *o2;  // device-mapped host destinations of the 8 residual sums and of the 45 / 3 Hessian sums
  int *flag;
  int seq;
};
template <int NV2>
__device__ __forceinline__ void fused_final_sum(const FuseSum &fs, const float *__restrict__ part, const float *__restrict__ part_gs) {
  if (!fs.ctr || threadIdx.x >= 64) return;  // wave 0 only: it stored this block's partials
  const int k = threadIdx.x;
  __threadfence();
  int old = 0;
  if (k == 0) old = atomicAdd(fs.ctr, 1);
  old = __shfl(old, 0, 64);
  if (old != (int)gridDim.x - 1) return;
  __threadfence();
  if (k == 0) *fs.ctr = 0;
  const int nblk = gridDim.x;
  if (k < 8) {
    double acc = 0;
#pragma unroll 8
    for (int b = 0; b < nblk; b++) acc += (double)part[(size_t)b * 8 + k];
    fs.o1[k] = acc;
  } else if (k < 8 + NV2) {
    const int q = k - 8;
    double acc = 0;
#pragma unroll 8
    for (int b = 0; b < nblk; b++) acc += (double)part_gs[(size_t)b * NV2 + q];
    fs.o2[q] = acc;
  }
  __threadfence_system();
  if (k == 0) __hip_atomic_store(fs.flag, fs.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one template pixel of calcRes: v = (E, numTermsInE, numWarped, numSaturated, flowT, flowRT, flowNum, -), gsb = this pixel's
// warp-buffer entries as the reference stores them (zero weight for dropped pixels)
// (x, y, id, refColor) = the template pixel's record (pc_u, pc_v, pc_idepth, pc_color; MODE 2: a 3-D point, id holds z)
template <int MODE>
__device__ __forceinline__ void res_pixel(const ResArgs &a, int i, float x, float y, float id, float refColor, float *v, float *gsb) {
  {
    float pt0, pt1, pt2;
    if (MODE == 2) {  // pt = R (x, y, z) + t, src/LoopClosure/PoseEstimator.cpp:181-183
      pt0 = a.M[0] * x + a.M[1] * y + a.M[2] * id + a.t[0];
      pt1 = a.M[3] * x + a.M[4] * y + a.M[5] * id + a.t[1];
      pt2 = a.M[6] * x + a.M[7] * y + a.M[8] * id + a.t[2];
    } else {
      pt0 = a.M[0] * x + a.M[1] * y + a.M[2] + a.t[0] * id;
      pt1 = a.M[3] * x + a.M[4] * y + a.M[5] + a.t[1] * id;
      pt2 = a.M[6] * x + a.M[7] * y + a.M[8] + a.t[2] * id;
    }
    const float u = pt0 / pt2, vv = pt1 / pt2;
    const float Ku = a.fxl * u + a.cxl, Kv = a.fyl * vv + a.cyl;
    const float new_idepth = MODE == 2 ? 1 / pt2 : id / pt2;
    float o0 = new_idepth, o1 = u, o2 = vv;
    if (MODE == 1) {  // FS/ScaleOptimizer.cpp:333
      o0 = (a.RKi[0] * x + a.RKi[1] * y + a.RKi[2]) / id;
      o1 = (a.RKi[3] * x + a.RKi[4] * y + a.RKi[5]) / id;
      o2 = (a.RKi[6] * x + a.RKi[7] * y + a.RKi[8]) / id;
    }
    if (MODE == 2 && a.lvl == 0 && (i & 31) == 0) {  // flow indicators of PoseEstimator::calcRes, :190-223
      const float Ku0 = a.fxl * (x / id) + a.cxl, Kv0 = a.fyl * (y / id) + a.cyl;
      const float pT0 = x + a.t[0], pT1 = y + a.t[1], pT2 = 1 + a.t[2];
      const float KuT = a.fxl * (pT0 / pT2) + a.cxl, KvT = a.fyl * (pT1 / pT2) + a.cyl;
      const float qT0 = x - a.t[0], qT1 = y - a.t[1], qT2 = 1 - a.t[2];
      const float KuT2 = a.fxl * (qT0 / qT2) + a.cxl, KvT2 = a.fyl * (qT1 / qT2) + a.cyl;
      const float p30 = a.M[0] * x + a.M[1] * y + a.M[2] - a.t[0], p31 = a.M[3] * x + a.M[4] * y + a.M[5] - a.t[1],
                  p32 = a.M[6] * x + a.M[7] * y + a.M[8] - a.t[2];
      const float Ku3 = a.fxl * (p30 / p32) + a.cxl, Kv3 = a.fyl * (p31 / p32) + a.cyl;
      float sT = 0, sRT = 0;
      sT += (KuT - Ku0) * (KuT - Ku0) + (KvT - Kv0) * (KvT - Kv0);
      sT += (KuT2 - Ku0) * (KuT2 - Ku0) + (KvT2 - Kv0) * (KvT2 - Kv0);
      sRT += (Ku - Ku0) * (Ku - Ku0) + (Kv - Kv0) * (Kv - Kv0);
      sRT += (Ku3 - Ku0) * (Ku3 - Ku0) + (Kv3 - Kv0) * (Kv3 - Kv0);
      v[4] = sT;
      v[5] = sRT;
      v[6] = 2.f;
    }
    if (MODE != 2 && a.lvl == 0 && (i & 31) == 0) {  // flow indicators, FS/CoarseTracker.cpp:666-696
      const float a0 = a.KiS[0] * x + a.KiS[1] * y + a.KiS[2], a1 = a.KiS[3] * x + a.KiS[4] * y + a.KiS[5],
                  a2 = a.KiS[6] * x + a.KiS[7] * y + a.KiS[8];
      const float pT0 = a0 + a.t[0] * id, pT1 = a1 + a.t[1] * id, pT2 = a2 + a.t[2] * id;
      const float KuT = a.fxl * (pT0 / pT2) + a.cxl, KvT = a.fyl * (pT1 / pT2) + a.cyl;
      const float qT0 = a0 - a.t[0] * id, qT1 = a1 - a.t[1] * id, qT2 = a2 - a.t[2] * id;
      const float KuT2 = a.fxl * (qT0 / qT2) + a.cxl, KvT2 = a.fyl * (qT1 / qT2) + a.cyl;
      const float m0 = a.M[0] * x + a.M[1] * y + a.M[2], m1 = a.M[3] * x + a.M[4] * y + a.M[5],
                  m2 = a.M[6] * x + a.M[7] * y + a.M[8];
      const float p30 = m0 - a.t[0] * id, p31 = m1 - a.t[1] * id, p32 = m2 - a.t[2] * id;
      const float Ku3 = a.fxl * (p30 / p32) + a.cxl, Kv3 = a.fyl * (p31 / p32) + a.cyl;
      float sT = 0, sRT = 0;
      sT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      sRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      v[4] = sT;
      v[5] = sRT;
      v[6] = 2.f;
    }
    float b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
    bool warped = false;
    if (Ku > 2 && Kv > 2 && Ku < (float)(a.wl - 3) && Kv < (float)(a.hl - 3) && new_idepth > 0) {
      int ix = (int)Ku, iy = (int)Kv;
      const float fdx = Ku - (float)ix, fdy = Kv - (float)iy, dxdy = fdx * fdy;
      const float *bp = a.dINew + 3 * (ix + iy * a.wl);
      const float *bq = bp + 3 * a.wl;
      const float w11 = dxdy, w01 = fdy - dxdy, w10 = fdx - dxdy, w00 = 1 - fdx - fdy + dxdy;
      const float hit0 = w11 * bq[3] + w01 * bq[0] + w10 * bp[3] + w00 * bp[0];
      const float hit1 = w11 * bq[4] + w01 * bq[1] + w10 * bp[4] + w00 * bp[1];
      const float hit2 = w11 * bq[5] + w01 * bq[2] + w10 * bp[5] + w00 * bp[2];
      if (isfinite(hit0)) {
        const float residual = MODE == 1 ? hit0 - refColor : hit0 - (float)(a.aff0 * refColor + a.aff1);
        const float hw = fabsf(residual) < a.huber ? 1 : a.huber / fabsf(residual);
        if (fabsf(residual) > a.cutoff) {
          v[0] = a.maxEnergy;
          v[1] = 1.f;
          v[3] = 1.f;
        } else {
          v[0] = hw * residual * residual * (2 - hw);
          v[1] = 1.f;
          v[2] = 1.f;
          warped = true;
          b3 = hit1; b4 = hit2; b5 = residual; b6 = hw; b7 = refColor;
        }
      }
    }
    gsb[0] = warped ? o0 : 0.f; gsb[1] = warped ? o1 : 0.f; gsb[2] = warped ? o2 : 0.f;
    gsb[3] = b3; gsb[4] = b4; gsb[5] = b5; gsb[6] = b6; gsb[7] = b7;
  }
}
// the calcGSSSE products of one pixel from its warp-buffer entries: the arithmetic of k_calc_gs / k_calc_gs_scale
__device__ __forceinline__ void gs_pixel_scale(const float *gsb, const GsFuse &gf, float *g3) {
  {
      {
        const float dxfx = gsb[3] * gf.fxl, dyfy = gsb[4] * gf.fyl;
        const float rx1 = gsb[0], rx2 = gsb[1], rx3 = gsb[2];
        const float deno_sqrt = gf.s * rx3 + gf.tz;
        const float deno = 1.0f / (deno_sqrt * deno_sqrt);
        const float xno = rx1 * gf.tz - rx3 * gf.tx, yno = rx2 * gf.tz - rx3 * gf.ty;
        const float J0 = dxfx * (deno * xno) + dyfy * (deno * yno);
        const float J1 = gsb[5], w = gsb[6];
        if (w != 0.f) {
          const float J0w = J0 * w, J1w = J1 * w;
          g3[0] = J0w * J0;
          g3[1] = J0w * J1;
          g3[2] = J1w * J1;
        }
      }
  }
}
__device__ __forceinline__ void gs_pixel_pose(const float *gsb, const GsFuse &gf, float *g45) {
  {
      {
        const float dx = gsb[3] * gf.fxl, dy = gsb[4] * gf.fyl;
        const float u = gsb[1], vv = gsb[2], id = gsb[0];
        float J[9];
        J[0] = id * dx;
        J[1] = id * dy;
        J[2] = 0 - id * (u * dx + vv * dy);
        J[3] = 0 - (u * vv * dx + dy * (1 + vv * vv));
        J[4] = u * vv * dy + dx * (1 + u * u);
        J[5] = u * dy - vv * dx;
        J[6] = gf.a * (gf.b0 - gsb[7]);
        J[7] = -1;
        J[8] = gsb[5];
        const float w = gsb[6];
        int idx = 0;
#pragma unroll
        for (int r = 0; r < 9; r++) {
          const float Jw = J[r] * w;
#pragma unroll
          for (int cc = r; cc < 9; cc++) g45[idx++] = Jw * J[cc];
        }
      }
  }
}
template <int MODE, bool GS>
__global__ __launch_bounds__(256) void k_calc_res(ResArgs a, float *__restrict__ part /* nblk*8 */, GsFuse gf,
                                                  float *__restrict__ part_gs /* nblk*45 | nblk*3 */, FuseSum fs) {
  __shared__ float sm[8 * 4];
  __shared__ float smg[(MODE == 1 ? 3 : 45) * 4];
  float gsb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < a.n) {
    res_pixel<MODE>(a, i, a.pu[i], a.pv[i], a.pid[i], a.pcol[i], v, gsb);
#pragma unroll
    for (int k = 0; k < 8; k++) a.buf[k][i] = gsb[k];
  }
  block_sum<8>(v, sm, part + 8 * (size_t)blockIdx.x);
  if (GS) {  // the same arithmetic as k_calc_gs / k_calc_gs_scale on the values just stored
    if (MODE == 1) {
      float g3[3] = {0, 0, 0};
      if (i < a.n) gs_pixel_scale(gsb, gf, g3);
      block_sum<3>(g3, smg, part_gs + 3 * (size_t)blockIdx.x);
    } else {
      float g45[45];
#pragma unroll
      for (int k = 0; k < 45; k++) g45[k] = 0.f;
      if (i < a.n) gs_pixel_pose(gsb, gf, g45);
      block_sum<45>(g45, smg, part_gs + 45 * (size_t)blockIdx.x);
    }
  }
  fused_final_sum<GS ? (MODE == 1 ? 3 : 45) : 0>(fs, part, part_gs);
}

// both final sums of a speculative call in one launch
// These two kernels are ONE wave each; their results go to device-mapped host memory.  A system-scope fence executed by
// the whole wave orders every lane's stores before lane 0 publishes the sequence number the host is polling for.
__global__ void k_sum_parts2(const float *__restrict__ p1, int nv1, double *__restrict__ o1, const float *__restrict__ p2, int nv2,
                             double *__restrict__ o2, int nblk, int *flag, int seq) {
  const int k = threadIdx.x;
  if (k < nv1) {
    double a = 0;
#pragma unroll 8
    for (int b = 0; b < nblk; b++) a += (double)p1[(size_t)b * nv1 + k];
    o1[k] = a;
  } else if (k < nv1 + nv2) {
    const int q = k - nv1;
    double a = 0;
#pragma unroll 8
    for (int b = 0; b < nblk; b++) a += (double)p2[(size_t)b * nv2 + q];
    o2[q] = a;
  }
  if (flag) {
    __threadfence_system();
    if (k == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// final fixed-order sum of per-block partials in double
__global__ void k_sum_parts(const float *__restrict__ part, int nblk, int nv, double *__restrict__ out, int *flag = nullptr,
                            int seq = 0) {
  const int k = threadIdx.x;
  if (k < nv) {
    double a = 0;
#pragma unroll 8
    for (int b = 0; b < nblk; b++) a += (double)part[(size_t)b * nv + k];
    out[k] = a;
  }
  if (flag) {
    __threadfence_system();
    if (k == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static void fill_common(sos_tracker *T, ResArgs &a, int lvl, const float *RKi, const float *t, float scale, bool scaleMode,
                        float cutoffTH) {
  a.pu = T->pc_u[lvl]; a.pv = T->pc_v[lvl]; a.pid = T->pc_idepth[lvl]; a.pcol = T->pc_color[lvl];
  if (T->loop_mode) { a.pu = T->l_xyz[0]; a.pv = T->l_xyz[1]; a.pid = T->l_xyz[2]; a.pcol = T->l_col[lvl]; }
  for (int k = 0; k < 8; k++) a.buf[k] = T->buf[k];
  a.n = T->pc_n[lvl]; a.lvl = lvl; a.wl = T->w[lvl]; a.hl = T->h[lvl];
  float Ki[9] = {1.0f / T->fx[lvl], 0, -T->cx[lvl] / T->fx[lvl], 0, 1.0f / T->fy[lvl], -T->cy[lvl] / T->fy[lvl], 0, 0, 1};
  for (int i = 0; i < 9; i++) {
    a.RKi[i] = RKi[i];
    a.M[i] = scaleMode ? scale * RKi[i] : RKi[i];
    a.KiS[i] = scaleMode ? scale * Ki[i] : Ki[i];
  }
  for (int i = 0; i < 3; i++) a.t[i] = t[i];
  a.huber = T->prm.huberTH;
  a.cutoff = cutoffTH;
  a.maxEnergy = 2 * a.huber * cutoffTH - a.huber * a.huber;
}

struct GsArgs;
static void enqueue_gs(sos_tracker *T, int lvl, float a, float b0, int nblk);
static void enqueue_gs_scale(sos_tracker *T, int lvl, const float *t, const float *K1, float scale, int nblk);

static int finish_res(sos_tracker *T, int nblk, double *rs) {
  hipStream_t st = T->ctx->stream;
  SOS_HIP(hipGetLastError());
  if (nblk > 0) {  // poll the flag the last (single-wave) kernel publishes; stream sync as the 50 ms fallback
    int *flag = reinterpret_cast<int *>(T->pin_o + 60);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != T->seq) {
      __builtin_ia32_pause();
      if ((++spins & 4095u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) {
        SOS_HIP(hipStreamSynchronize(st));
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != T->seq) return SOS_ERR_HIP;
        break;
      }
    }
  } else {
    SOS_HIP(hipStreamSynchronize(st));
  }
  double o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (nblk > 0) memcpy(o, T->pin_o, sizeof(o));
  const int numTermsInE = (int)o[1], numWarped = (int)o[2], numSaturated = (int)o[3];
  T->buf_count = numWarped;
  T->buf_n = (numWarped + 3) / 4 * 4;
  const float sumT = (float)o[4], sumRT = (float)o[5], sumNum = (float)o[6];
  rs[0] = (double)(float)o[0];
  rs[1] = numTermsInE;
  rs[2] = sumT / (sumNum + 0.1);
  rs[3] = 0;
  rs[4] = sumRT / (sumNum + 0.1);
  rs[5] = numSaturated / (float)numTermsInE;
  return SOS_OK;
}

extern "C" int sos_tracker_set_gs_hint(sos_tracker *T, int on, float b0) {
  if (!T) return SOS_ERR_ARG;
  T->hint_on = on != 0;
  T->hint_b0 = b0;
  T->gs_cached = false;
  return SOS_OK;
}

// destinations of the final sums of a calc_res launch.  The in-kernel form pays one device-scope fence per block
// (~0.13 us each, serialised per XCD) to save a launch: measured faster up to about 30 blocks (W7: trackNewestCoarse
// 0.45 -> 0.42 ms, optimizeScale 0.20 -> 0.18 ms), even at 40-55 blocks; above the threshold the second kernel stays
static FuseSum fuse_sum(sos_tracker *T, int nblk) {
  static const int maxFused = getenv("SOS_TRACKER_FUSE_MAX") ? atoi(getenv("SOS_TRACKER_FUSE_MAX")) : 32;
  FuseSum fs;
  fs.ctr = nblk <= maxFused ? T->d_fusectr : nullptr;
  fs.o1 = T->pin_o_dev;
  fs.o2 = T->pin_o_dev + 8;
  fs.flag = reinterpret_cast<int *>(T->pin_o_dev + 60);
  fs.seq = ++T->seq;
  return fs;
}

extern "C" int sos_tracker_calc_res(sos_tracker *T, int lvl, int newSlot, const float *RKi, const float *t, const float *affLL,
                                    float cutoffTH, double *rs) {
  if (!T || !T->have_ref || lvl < 0 || lvl >= T->levels || !RKi || !t || !affLL || !rs) return SOS_ERR_ARG;
  sos_ctx *c = T->ctx;
  if (newSlot < 0 || newSlot >= SOS_MAX_SLOTS || !c->dI[newSlot][lvl]) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(c->device));
  ResArgs a;
  fill_common(T, a, lvl, RKi, t, 1.0f, false, cutoffTH);
  a.dINew = c->dI[newSlot][lvl];
  a.fxl = T->fx[lvl]; a.fyl = T->fy[lvl]; a.cxl = T->cx[lvl]; a.cyl = T->cy[lvl];
  a.aff0 = affLL[0]; a.aff1 = affLL[1];
  const int nblk = divup(a.n, 256);
  T->gs_cached = T->gss_cached = false;
  if (nblk > 0) {
    GsFuse gf = {T->fx[lvl], T->fy[lvl], affLL[0], T->hint_b0, 1.f, 0.f, 0.f, 0.f};
    if (T->loop_mode) {  // PoseEstimator::calcRes: RKi is the plain rotation here
      const FuseSum fs = fuse_sum(T, nblk);
      if (T->hint_on) k_calc_res<2, true><<<nblk, 256, 0, c->stream>>>(a, T->d_part, gf, T->d_part2, fs);
      else k_calc_res<2, false><<<nblk, 256, 0, c->stream>>>(a, T->d_part, gf, nullptr, fs);
      if (T->hint_on) {
        if (!fs.ctr)
          k_sum_parts2<<<1, 64, 0, c->stream>>>(T->d_part, 8, T->pin_o_dev, T->d_part2, 45, T->pin_o_dev + 8, nblk, fs.flag, fs.seq);
        T->gs_cached = true; T->gs_lvl = lvl; T->gs_a = affLL[0]; T->gs_b0 = T->hint_b0;
      } else if (!fs.ctr) {
        k_sum_parts<<<1, 64, 0, c->stream>>>(T->d_part, nblk, 8, T->pin_o_dev, fs.flag, fs.seq);
      }
    } else if (T->hint_on) {  // speculative calcGSSSE for this pose inside the same kernel (see sos_tracker)
      const FuseSum fs = fuse_sum(T, nblk);
      k_calc_res<0, true><<<nblk, 256, 0, c->stream>>>(a, T->d_part, gf, T->d_part2, fs);
      if (!fs.ctr) k_sum_parts2<<<1, 64, 0, c->stream>>>(T->d_part, 8, T->pin_o_dev, T->d_part2, 45, T->pin_o_dev + 8, nblk, fs.flag, fs.seq);
      T->gs_cached = true; T->gs_lvl = lvl; T->gs_a = affLL[0]; T->gs_b0 = T->hint_b0;
    } else {
      const FuseSum fs = fuse_sum(T, nblk);
      k_calc_res<0, false><<<nblk, 256, 0, c->stream>>>(a, T->d_part, gf, nullptr, fs);
      if (!fs.ctr) k_sum_parts<<<1, 64, 0, c->stream>>>(T->d_part, nblk, 8, T->pin_o_dev, fs.flag, fs.seq);
    }
  }
  T->buf_lvl = lvl;
  return finish_res(T, nblk, rs);
}

extern "C" int sos_tracker_calc_res_scale(sos_tracker *T, int lvl, int stereoSlot, const float *RKi, const float *t,
                                          const float *K1, float scale, float cutoffTH, double *rs) {
  if (!T || !T->have_ref || lvl < 0 || lvl >= T->levels || !RKi || !t || !K1 || !rs) return SOS_ERR_ARG;
  sos_ctx *c = T->ctx;
  if (stereoSlot < 0 || stereoSlot >= SOS_MAX_SLOTS || !c->dI[stereoSlot][lvl]) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(c->device));
  ResArgs a;
  fill_common(T, a, lvl, RKi, t, scale, true, cutoffTH);
  a.dINew = c->dI[stereoSlot][lvl];
  a.fxl = K1[0]; a.fyl = K1[1]; a.cxl = K1[2]; a.cyl = K1[3];
  a.aff0 = 1; a.aff1 = 0;
  const int nblk = divup(a.n, 256);
  T->gs_cached = T->gss_cached = false;
  if (nblk > 0) {
    GsFuse gf = {K1[0], K1[1], 0.f, 0.f, scale, t[0], t[1], t[2]};
    if (!T->hint_on) {
      const FuseSum fs = fuse_sum(T, nblk);
      k_calc_res<1, false><<<nblk, 256, 0, c->stream>>>(a, T->d_part, gf, nullptr, fs);
      if (!fs.ctr) k_sum_parts<<<1, 64, 0, c->stream>>>(T->d_part, nblk, 8, T->pin_o_dev, fs.flag, fs.seq);
    } else {  // calcGSSSEScale needs nothing beyond what calcResScale was given: same kernel
      const FuseSum fs = fuse_sum(T, nblk);
      k_calc_res<1, true><<<nblk, 256, 0, c->stream>>>(a, T->d_part, gf, T->d_part2, fs);
      if (!fs.ctr) k_sum_parts2<<<1, 64, 0, c->stream>>>(T->d_part, 8, T->pin_o_dev, T->d_part2, 3, T->pin_o_dev + 8, nblk, fs.flag, fs.seq);
      T->gss_cached = true;
      T->gss_key[0] = (float)lvl; T->gss_key[1] = t[0]; T->gss_key[2] = t[1]; T->gss_key[3] = t[2];
      T->gss_key[4] = K1[0]; T->gss_key[5] = K1[1]; T->gss_key[6] = scale;
    }
  }
  T->buf_lvl = lvl;
  return finish_res(T, nblk, rs);
}

// ------------------------------------------------------------------------------------------------
// calcGSSSEPose: 45 uniques of the weighted 9x9 J^T W J per template pixel, block tree sums
// ------------------------------------------------------------------------------------------------
struct GsArgs {
  const float *buf[8];
  int n;
  float fxl, fyl, a, b0;
  float s, tx, ty, tz;
};

__global__ __launch_bounds__(256) void k_calc_gs(GsArgs g, float *__restrict__ part /* nblk*45 */) {
  __shared__ float sm[45 * 4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v[45];
#pragma unroll
  for (int k = 0; k < 45; k++) v[k] = 0.f;
  if (i < g.n) {
    const float dx = g.buf[3][i] * g.fxl, dy = g.buf[4][i] * g.fyl;
    const float u = g.buf[1][i], vv = g.buf[2][i], id = g.buf[0][i];
    float J[9];
    J[0] = id * dx;  // FS/CoarseTracker.cpp:571-591
    J[1] = id * dy;
    J[2] = 0 - id * (u * dx + vv * dy);
    J[3] = 0 - (u * vv * dx + dy * (1 + vv * vv));
    J[4] = u * vv * dy + dx * (1 + u * u);
    J[5] = u * dy - vv * dx;
    J[6] = g.a * (g.b0 - g.buf[7][i]);
    J[7] = -1;
    J[8] = g.buf[5][i];
    const float w = g.buf[6][i];
    int idx = 0;
#pragma unroll
    for (int r = 0; r < 9; r++) {
      const float Jw = J[r] * w;  // OB/MatrixAccumulators.h:1320-1425
#pragma unroll
      for (int cc = r; cc < 9; cc++) v[idx++] = Jw * J[cc];
    }
  }
  block_sum<45>(v, sm, part + 45 * (size_t)blockIdx.x);
}

__global__ __launch_bounds__(256) void k_calc_gs_scale(GsArgs g, float *__restrict__ part /* nblk*3 */) {
  __shared__ float sm[3 * 4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v[3] = {0, 0, 0};
  if (i < g.n) {  // FS/ScaleOptimizer.cpp:246-263
    const float dxfx = g.buf[3][i] * g.fxl, dyfy = g.buf[4][i] * g.fyl;
    const float rx1 = g.buf[0][i], rx2 = g.buf[1][i], rx3 = g.buf[2][i];
    const float deno_sqrt = g.s * rx3 + g.tz;
    const float deno = 1.0f / (deno_sqrt * deno_sqrt);
    const float xno = rx1 * g.tz - rx3 * g.tx, yno = rx2 * g.tz - rx3 * g.ty;
    const float J0 = dxfx * (deno * xno) + dyfy * (deno * yno);
    const float J1 = g.buf[5][i], w = g.buf[6][i];
    if (w != 0.f) {  // dropped pixels carry rx = 0 (0/0 guards not needed) and weight 0
      const float J0w = J0 * w, J1w = J1 * w;
      v[0] = J0w * J0;
      v[1] = J0w * J1;
      v[2] = J1w * J1;
    }
  }
  block_sum<3>(v, sm, part + 3 * (size_t)blockIdx.x);
}

static void enqueue_gs(sos_tracker *T, int lvl, float a, float b0, int nblk) {
  GsArgs g;
  for (int k = 0; k < 8; k++) g.buf[k] = T->buf[k];
  g.n = T->pc_n[lvl];
  g.fxl = T->fx[lvl]; g.fyl = T->fy[lvl]; g.a = a; g.b0 = b0;
  g.s = 1; g.tx = g.ty = g.tz = 0;
  hipStream_t st = T->ctx->stream;
  k_calc_gs<<<nblk, 256, 0, st>>>(g, T->d_part2);
  k_sum_parts<<<1, 64, 0, st>>>(T->d_part2, nblk, 45, T->pin_o_dev + 8);
}
static void enqueue_gs_scale(sos_tracker *T, int lvl, const float *t, const float *K1, float scale, int nblk) {
  GsArgs g;
  for (int k = 0; k < 8; k++) g.buf[k] = T->buf[k];
  g.n = T->pc_n[lvl];
  g.fxl = K1[0]; g.fyl = K1[1]; g.a = 0; g.b0 = 0;
  g.s = scale; g.tx = t[0]; g.ty = t[1]; g.tz = t[2];
  hipStream_t st = T->ctx->stream;
  k_calc_gs_scale<<<nblk, 256, 0, st>>>(g, T->d_part2);
  k_sum_parts<<<1, 64, 0, st>>>(T->d_part2, nblk, 3, T->pin_o_dev + 8);
}

extern "C" int sos_tracker_calc_gs(sos_tracker *T, int lvl, float a, float b0, double *H_out, double *b_out) {
  if (!T || !T->have_ref || lvl != T->buf_lvl || !H_out || !b_out) return SOS_ERR_STATE;
  sos_ctx *c = T->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  double o[45];
  for (int k = 0; k < 45; k++) o[k] = 0;
  const int nblk = divup(T->pc_n[lvl], 256);
  if (nblk > 0) {
    if (!(T->gs_cached && T->gs_lvl == lvl && T->gs_a == a && T->gs_b0 == b0)) {  // not speculated: run it now
      enqueue_gs(T, lvl, a, b0, nblk);
      SOS_HIP(hipGetLastError());
      SOS_HIP(hipStreamSynchronize(st));
      T->gs_cached = true; T->gs_lvl = lvl; T->gs_a = a; T->gs_b0 = b0;
    }
    memcpy(o, T->pin_o + 8, sizeof(o));
  }
  float Hf[81];
  int idx = 0;
  for (int r = 0; r < 9; r++)
    for (int cc = r; cc < 9; cc++) {
      Hf[9 * r + cc] = Hf[9 * cc + r] = (float)o[idx];
      idx++;
    }
  const int n = T->buf_n;
  const double inv = 1.0f / n;  // FS/CoarseTracker.cpp:595-596
  const double sc[8] = {SOS_SCALE_XI_ROT, SOS_SCALE_XI_ROT, SOS_SCALE_XI_ROT, SOS_SCALE_XI_TRANS,
                        SOS_SCALE_XI_TRANS, SOS_SCALE_XI_TRANS, SOS_SCALE_A, SOS_SCALE_B};
  for (int r = 0; r < 8; r++) {
    for (int cc = 0; cc < 8; cc++) H_out[8 * r + cc] = (double)Hf[9 * r + cc] * inv * (sc[r] * sc[cc]);
    b_out[r] = (double)Hf[9 * r + 8] * inv * sc[r];
  }
  return SOS_OK;
}

extern "C" int sos_tracker_calc_gs_scale(sos_tracker *T, int lvl, const float *t, const float *K1, float scale, float *H_out,
                                         float *b_out) {
  if (!T || !T->have_ref || lvl != T->buf_lvl || !t || !K1 || !H_out || !b_out) return SOS_ERR_STATE;
  sos_ctx *c = T->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  double o[3] = {0, 0, 0};
  const int nblk = divup(T->pc_n[lvl], 256);
  if (nblk > 0) {
    const float key[7] = {(float)lvl, t[0], t[1], t[2], K1[0], K1[1], scale};
    if (!(T->gss_cached && memcmp(key, T->gss_key, sizeof(key)) == 0)) {
      enqueue_gs_scale(T, lvl, t, K1, scale, nblk);
      SOS_HIP(hipGetLastError());
      SOS_HIP(hipStreamSynchronize(st));
      T->gss_cached = true;
      memcpy(T->gss_key, key, sizeof(key));
    }
    memcpy(o, T->pin_o + 8, sizeof(o));
  }
  const int n = T->buf_n;
  *H_out = (float)o[0] * (1.0f / n);
  *b_out = (float)o[1] * (1.0f / n);
  return SOS_OK;
}

#include "sos_tracker_lm.inc"
