// sos_pixsel.hip -- PixelSelector (FS/PixelSelector2.cpp) on gfx950: the candidate-pixel selection in front of the
// ImmaturePoint constructor (FullSystem::makeNewTraces, FS/FullSystem.cpp:1071-1097).
//   k_pixsel_hist / k_pixsel_smooth   makeHists   :69-155
//   k_pixsel_cells .. k_pixsel_lvl4   select      :292-424
//   sos_pixsel_make_maps              makeMaps    :157-290 (re-selection recursion on the host, sub-sampling on the device)
// The reference's select is one nested loop whose only loop-carried state is n2, the running count of level-0
// selections: it picks the direction (randomPattern[n2] & 0xF) each pot / 2 pot / 4 pot cell ranks its gradients
// against.  Whether a cell selects at all depends on that direction only when every passing gradient is exactly
// orthogonal to it, so the device (1) computes per cell a 16-bit mask "selects under direction d", (2) turns the masks
// into every cell's n2 by a scan in the reference's traversal order -- a plain prefix sum wherever the masks are all
// 0x0000 / 0xFFFF, a serial walk over the rare chunk that holds a direction-dependent cell -- and (3) picks the pixels
// per cell, then per 2 pot block, then per 4 pot block, each level looking at the trigger flags of the level below
// exactly as bestIdx3 / bestIdx4 = -2 do in the reference.  Same fp32 convention as the rest (no FMA contraction):
// maps, counts and thresholds are bit-identical to oracle/orc_pixsel.c.
#include "sos_common.h"

#include <vector>

struct sos_pixsel {
  sos_ctx *ctx = nullptr;
  sos_pixsel_params prm;
  int w = 0, h = 0, w32 = 0, h32 = 0;
  uint8_t *d_pattern = nullptr;
  float *d_ths = nullptr, *d_thsSm = nullptr, *d_map = nullptr;
  unsigned short *d_mask = nullptr;  // per cell slot
  int *d_n2start = nullptr;          // per cell slot
  unsigned char *d_trig = nullptr;   // per cell slot: level-0 trigger; then per 2 pot block (4 per 4 pot block): level-1 trigger
  int *d_counts = nullptr;           // n2 n3 n4 | scan scratch
  int *d_blockcnt = nullptr;
  int2 *d_list = nullptr;            // (pixel index, map value bits) of the selected pixels in row-major order
  size_t cap_slots = 0;
  int hist_slot = -1;
};

namespace {
__constant__ float c_dirs[16][2] = {{0, 1.0000f},        {0.3827f, 0.9239f},  {0.1951f, 0.9808f},  {0.9239f, 0.3827f},
                                    {0.7071f, 0.7071f},  {0.3827f, -0.9239f}, {0.8315f, 0.5556f},  {0.8315f, -0.5556f},
                                    {0.5556f, -0.8315f}, {0.9808f, 0.1951f},  {0.9239f, -0.3827f}, {0.7071f, -0.7071f},
                                    {0.5556f, 0.8315f},  {0.9808f, -0.1951f}, {1.0000f, 0.0000f},  {0.1951f, -0.9808f}};

struct SelArgs {
  const float *dI, *absg0, *absg1, *absg2, *thsSm;
  const uint8_t *pattern;
  int w, h, w1, w2, thsStep, thsN, pot, nbx, nby;  // nbx x nby blocks of 4 pot
  float thFactor, dw1, dw2;
  int useDir;
};

// ---- makeHists ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pixsel_hist(const float *__restrict__ absg0, int w, int h, int w32, float below, float add,
                                                     float *__restrict__ ths) {
  __shared__ int hist[50];
  const int x = blockIdx.x % w32, y = blockIdx.x / w32;
  if (threadIdx.x < 50) hist[threadIdx.x] = 0;
  __syncthreads();
  for (int q = threadIdx.x; q < 1024; q += 256) {
    const int i = q & 31, j = q >> 5;
    const int it = i + 32 * x, jt = j + 32 * y;
    if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
    int g = (int)sqrtf(absg0[it + jt * w]);
    if (g > 48) g = 48;
    atomicAdd(&hist[g + 1], 1);
    atomicAdd(&hist[0], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // computeHistQuantil, :59-67
    int th = (int)(hist[0] * below + 0.5f);
    int res = 90;
    for (int i = 0; i < 90; i++) {
      th -= (i + 1 < 50) ? hist[i + 1] : 0;
      if (th < 0) { res = i; break; }
    }
    ths[x + y * w32] = res + add;
  }
}
__global__ void k_pixsel_smooth(const float *__restrict__ ths, int w32, int h32, float *__restrict__ thsSm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w32 * h32) return;
  const int x = i % w32, y = i / w32;
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
  thsSm[i] = (sum / num) * (sum / num);
}

// ---- select ------------------------------------------------------------------------------------------------------------
// Cell slots: every 4 pot block owns 16 slots, slot = 4 * (2 pot sub-block, row-major 2x2) + (cell inside it, row-major 2x2)
// -- the reference's traversal order; cells cut off by the image border stay empty.
struct Cell {
  int x0, y0, mx, my;  // pixel origin and extent (0 extent = empty slot)
};
__device__ __forceinline__ Cell cell_of_slot(const SelArgs &a, int slot) {
  const int blk = slot >> 4, sub = (slot >> 2) & 3, cc = slot & 3;
  const int x4 = (blk % a.nbx) * 4 * a.pot, y4 = (blk / a.nbx) * 4 * a.pot;
  const int x234 = x4 + (sub & 1) * 2 * a.pot + (cc & 1) * a.pot, y234 = y4 + (sub >> 1) * 2 * a.pot + (cc >> 1) * a.pot;
  Cell c;
  c.x0 = x234;
  c.y0 = y234;
  c.mx = max(0, min(a.pot, a.w - x234));
  c.my = max(0, min(a.pot, a.h - y234));
  return c;
}
__device__ __forceinline__ bool pix_valid(const SelArgs &a, int xf, int yf) { return !(xf < 4 || xf >= a.w - 5 || yf < 4 || yf > a.h - 4); }
__device__ __forceinline__ float pix_th0(const SelArgs &a, int xf, int yf) {
  const int ti = (xf >> 5) + (yf >> 5) * a.thsStep;
  return ti < a.thsN ? a.thsSm[ti] : 0.0f;
}

__global__ void k_pixsel_cells(SelArgs a, int nslots, unsigned short *__restrict__ mask) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  const Cell c = cell_of_slot(a, slot);
  unsigned m = 0;
  for (int y1 = 0; y1 < c.my; y1++)
    for (int x1 = 0; x1 < c.mx; x1++) {
      const int xf = c.x0 + x1, yf = c.y0 + y1;
      if (!pix_valid(a, xf, yf)) continue;
      const int idx = xf + a.w * yf;
      const float ag0 = a.absg0[idx];
      if (!(ag0 > pix_th0(a, xf, yf) * a.thFactor)) continue;
      if (!a.useDir) {
        if (ag0 > 0) m = 0xffffu;
        continue;
      }
      const float gx = a.dI[3 * idx + 1], gy = a.dI[3 * idx + 2];
#pragma unroll
      for (int d = 0; d < 16; d++)
        if (fabsf(gx * c_dirs[d][0] + gy * c_dirs[d][1]) > 0) m |= 1u << d;
    }
  mask[slot] = (unsigned short)m;
}

// n2 at the start of every cell slot, in slot order.  One block; chunks of 1024 slots.
__global__ __launch_bounds__(1024) void k_pixsel_scan(const unsigned short *__restrict__ mask, const uint8_t *__restrict__ pattern,
                                                      int nslots, int *__restrict__ n2start, int *__restrict__ counts) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nslots; c0 += 1024) {
    const int i = c0 + tid;
    const unsigned m = i < nslots ? mask[i] : 0u;
    const int dep = (m != 0u && m != 0xffffu) ? 1 : 0;
    const int anyDep = __syncthreads_or(dep);
    const int base = s_base;
    if (!anyDep) {
      const unsigned long long b = __ballot(m != 0u);
      const int excl = __popcll(b & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wv] = __popcll(b);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if (k < wv) off += s_wave[k];
        tot += s_wave[k];
      }
      if (i < nslots) n2start[i] = base + off + excl;
      __syncthreads();
      if (tid == 0) s_base = base + tot;
    } else if (tid == 0) {  // a direction-dependent cell in this chunk: the reference's serial recurrence
      int n2 = base;
      const int e = min(nslots, c0 + 1024);
      for (int k = c0; k < e; k++) {
        n2start[k] = n2;
        n2 += (mask[k] >> (pattern[n2] & 0xF)) & 1;
      }
      s_base = n2;
    }
    __syncthreads();
  }
  if (tid == 0) counts[0] = s_base;
}

// level 0: the pixel of every pot cell
__global__ void k_pixsel_pick(SelArgs a, int nslots, const int *__restrict__ n2start, float *__restrict__ map, unsigned char *__restrict__ trig) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  const Cell c = cell_of_slot(a, slot);
  const int d = a.pattern[n2start[slot]] & 0xF;
  const float dx = c_dirs[d][0], dy = c_dirs[d][1];
  int bestIdx = -1;
  float bestVal = 0;
  for (int y1 = 0; y1 < c.my; y1++)
    for (int x1 = 0; x1 < c.mx; x1++) {
      const int xf = c.x0 + x1, yf = c.y0 + y1;
      if (!pix_valid(a, xf, yf)) continue;
      const int idx = xf + a.w * yf;
      const float ag0 = a.absg0[idx];
      if (ag0 > pix_th0(a, xf, yf) * a.thFactor) {
        float dirNorm = fabsf((float)(a.dI[3 * idx + 1] * dx + a.dI[3 * idx + 2] * dy));
        if (!a.useDir) dirNorm = ag0;
        if (dirNorm > bestVal) { bestVal = dirNorm; bestIdx = idx; }
      }
    }
  trig[slot] = bestIdx >= 0 ? 1 : 0;  // bestIdx3 = bestIdx4 = -2 in the reference
  if (bestIdx > 0) map[bestIdx] = 1.f;
}

// first maximum in traversal order over the lanes of a wave: larger value wins, equal values -> smaller order index
__device__ __forceinline__ void wave_first_max(float &val, int &ord, int &idx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(val, o, 64);
    const int o2 = __shfl_xor(ord, o, 64), i2 = __shfl_xor(idx, o, 64);
    if (v2 > val || (v2 == val && o2 < ord)) { val = v2; ord = o2; idx = i2; }
  }
}

// level 1: the pixel of every 2 pot block none of whose cells triggered.  One wave per block; lanes stride over the
// block's pixels in the reference's traversal order (cell by cell, row-major inside a cell).
__global__ __launch_bounds__(256) void k_pixsel_lvl3(SelArgs a, int nsub, const int *__restrict__ n2start, const unsigned char *__restrict__ trig,
                                                     float *__restrict__ map, unsigned char *__restrict__ trig3, int *__restrict__ counts) {
  const int sb = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // (4 pot block, sub-block)
  if (sb >= nsub) return;
  const int slot0 = sb << 2;
  if (lane == 0) trig3[sb] = 0;
  const Cell c0 = cell_of_slot(a, slot0);
  if (c0.mx == 0 || c0.my == 0) return;
  if (trig[slot0] | trig[slot0 + 1] | trig[slot0 + 2] | trig[slot0 + 3]) return;
  const int d = a.pattern[n2start[slot0]] & 0xF;
  const float dx = c_dirs[d][0], dy = c_dirs[d][1];
  const int pp = a.pot * a.pot;
  int bestIdx = -1, bestOrd = 0x7fffffff;
  float bestVal = 0;
  for (int q = lane; q < 4 * pp; q += 64) {
    const int cc = q / pp, r = q - cc * pp, y1 = r / a.pot, x1 = r - y1 * a.pot;
    const Cell c = cell_of_slot(a, slot0 + cc);
    if (x1 >= c.mx || y1 >= c.my) continue;
    const int xf = c.x0 + x1, yf = c.y0 + y1;
    if (!pix_valid(a, xf, yf)) continue;
    const int idx = xf + a.w * yf;
    const float pixelTH1 = pix_th0(a, xf, yf) * a.dw1;
    const float ag1 = a.absg1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * a.w1];
    if (ag1 > pixelTH1 * a.thFactor) {
      float dirNorm = fabsf((float)(a.dI[3 * idx + 1] * dx + a.dI[3 * idx + 2] * dy));
      if (!a.useDir) dirNorm = ag1;
      if (dirNorm > bestVal) { bestVal = dirNorm; bestIdx = idx; bestOrd = q; }  // q ascends per lane: first maximum
    }
  }
  wave_first_max(bestVal, bestOrd, bestIdx);
  if (lane == 0) {
    if (bestIdx >= 0) trig3[sb] = 1;
    if (bestIdx > 0) {
      map[bestIdx] = 2.f;
      atomicAdd(&counts[1], 1);
    }
  }
}

// level 2: the pixel of every 4 pot block without any trigger below; one wave per block
__global__ __launch_bounds__(256) void k_pixsel_lvl4(SelArgs a, int nblk, const int *__restrict__ n2start, const unsigned char *__restrict__ trig,
                                                     const unsigned char *__restrict__ trig3, float *__restrict__ map, int *__restrict__ counts) {
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blk >= nblk) return;
  const int slot0 = blk << 4;
  const bool any = (lane < 16 && trig[slot0 + lane]) || (lane >= 16 && lane < 20 && trig3[4 * blk + lane - 16]);
  if (__ballot(any) != 0ull) return;
  const int d = a.pattern[n2start[slot0]] & 0xF;
  const float dx = c_dirs[d][0], dy = c_dirs[d][1];
  const int pp = a.pot * a.pot;
  int bestIdx = -1, bestOrd = 0x7fffffff;
  float bestVal = 0;
  for (int q = lane; q < 16 * pp; q += 64) {
    const int k = q / pp, r = q - k * pp, y1 = r / a.pot, x1 = r - y1 * a.pot;
    const Cell c = cell_of_slot(a, slot0 + k);
    if (x1 >= c.mx || y1 >= c.my) continue;
    const int xf = c.x0 + x1, yf = c.y0 + y1;
    if (!pix_valid(a, xf, yf)) continue;
    const int idx = xf + a.w * yf;
    const float pixelTH1 = pix_th0(a, xf, yf) * a.dw1;
    const float pixelTH2 = pixelTH1 * a.dw2;
    const float ag2 = a.absg2[(int)(xf * 0.25f + 0.125) + (int)(yf * 0.25f + 0.125) * a.w2];
    if (ag2 > pixelTH2 * a.thFactor) {
      float dirNorm = fabsf((float)(a.dI[3 * idx + 1] * dx + a.dI[3 * idx + 2] * dy));
      if (!a.useDir) dirNorm = ag2;
      if (dirNorm > bestVal) { bestVal = dirNorm; bestIdx = idx; bestOrd = q; }
    }
  }
  wave_first_max(bestVal, bestOrd, bestIdx);
  if (lane == 0 && bestIdx > 0) {
    map[bestIdx] = 4.f;
    atomicAdd(&counts[2], 1);
  }
}

// ---- row-major list of the selected pixels (sub-sampling of makeMaps :262-275, the loop of makeNewTraces) ---------------
__global__ __launch_bounds__(1024) void k_pixsel_count(const float *__restrict__ map, int npx, int *__restrict__ blockcnt) {
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const int c = __syncthreads_count(i < npx && map[i] != 0.f);
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = c;
}
__global__ __launch_bounds__(1024) void k_pixsel_offsets(int *__restrict__ blockcnt, int nb, int *__restrict__ total) {
  __shared__ int s[1024];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nb; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int v = i < nb ? blockcnt[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
      const int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) blockcnt[i] = base + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) base += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = base;
}
// rank of every selected pixel in row-major order; charTH < 0: only list; else drop the pixels whose pattern byte exceeds it
__global__ __launch_bounds__(1024) void k_pixsel_rank(float *__restrict__ map, int npx, const int *__restrict__ blockoff,
                                                      const uint8_t *__restrict__ pattern, int charTH, int2 *__restrict__ list,
                                                      int *__restrict__ dropped) {
  __shared__ int s_wave[16];
  const int i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool sel = i < npx && map[i] != 0.f;
  const unsigned long long b = __ballot(sel);
  if (lane == 0) s_wave[wv] = __popcll(b);
  __syncthreads();
  int off = 0;
  for (int k = 0; k < wv; k++) off += s_wave[k];
  if (!sel) return;
  const int rn = blockoff[blockIdx.x] + off + __popcll(b & ((1ull << lane) - 1ull));
  if (charTH >= 0) {
    if ((int)pattern[rn] > charTH) {
      map[i] = 0.f;
      atomicAdd(dropped, 1);
    }
  } else if (list) {
    list[rn] = make_int2(i, __float_as_int(map[i]));
  }
}

int ensure_slots(sos_pixsel *ps, size_t nslots) {
  if (nslots <= ps->cap_slots) return SOS_OK;
  hipFree(ps->d_mask); hipFree(ps->d_n2start); hipFree(ps->d_trig);
  ps->d_mask = nullptr; ps->d_n2start = nullptr; ps->d_trig = nullptr;
  ps->cap_slots = 0;
  const size_t want = nslots + nslots / 4 + 64;
  if (hipMalloc(&ps->d_mask, sizeof(unsigned short) * want) != hipSuccess) return SOS_ERR_NOMEM;
  if (hipMalloc(&ps->d_n2start, sizeof(int) * want) != hipSuccess) return SOS_ERR_NOMEM;
  if (hipMalloc(&ps->d_trig, want + want / 4 + 16) != hipSuccess) return SOS_ERR_NOMEM;
  ps->cap_slots = want;
  return SOS_OK;
}

int do_hists(sos_pixsel *ps, int slot) {
  sos_ctx *c = ps->ctx;
  if (slot < 0 || slot >= SOS_MAX_SLOTS || !c->has_pyr[slot]) return SOS_ERR_STATE;
  if (ps->w32 * ps->h32 > 0) {
    k_pixsel_hist<<<ps->w32 * ps->h32, 256, 0, c->stream>>>(c->absg[slot][0], ps->w, ps->h, ps->w32, ps->prm.minGradHistCut,
                                                           ps->prm.minGradHistAdd, ps->d_ths);
    k_pixsel_smooth<<<(ps->w32 * ps->h32 + 255) / 256, 256, 0, c->stream>>>(ps->d_ths, ps->w32, ps->h32, ps->d_thsSm);
  }
  SOS_HIP(hipGetLastError());
  ps->hist_slot = slot;
  return SOS_OK;
}

int do_select(sos_pixsel *ps, int slot, int pot, float thFactor, int32_t n[3]) {
  sos_ctx *c = ps->ctx;
  if (slot < 0 || slot >= SOS_MAX_SLOTS || !c->has_pyr[slot] || c->levels < 3) return SOS_ERR_STATE;
  if (pot < 1) return SOS_ERR_ARG;
  SelArgs a;
  a.dI = c->dI[slot][0]; a.absg0 = c->absg[slot][0]; a.absg1 = c->absg[slot][1]; a.absg2 = c->absg[slot][2];
  a.thsSm = ps->d_thsSm; a.pattern = ps->d_pattern;
  a.w = ps->w; a.h = ps->h; a.w1 = c->wl[1]; a.w2 = c->wl[2];
  a.thsStep = ps->w32; a.thsN = ps->w32 * ps->h32; a.pot = pot;
  a.nbx = (ps->w + 4 * pot - 1) / (4 * pot); a.nby = (ps->h + 4 * pot - 1) / (4 * pot);
  a.thFactor = thFactor; a.dw1 = ps->prm.gradDownweightPerLevel; a.dw2 = a.dw1 * a.dw1;
  a.useDir = ps->prm.selectDirectionDistribution ? 1 : 0;
  const int nblk = a.nbx * a.nby, nslots = nblk * 16, nsub = nblk * 4;
  int rc = ensure_slots(ps, (size_t)nslots);
  if (rc) return rc;
  hipStream_t st = c->stream;
  SOS_HIP(hipMemsetAsync(ps->d_map, 0, sizeof(float) * (size_t)ps->w * ps->h, st));
  SOS_HIP(hipMemsetAsync(ps->d_counts, 0, sizeof(int) * 8, st));
  k_pixsel_cells<<<(nslots + 255) / 256, 256, 0, st>>>(a, nslots, ps->d_mask);
  k_pixsel_scan<<<1, 1024, 0, st>>>(ps->d_mask, ps->d_pattern, nslots, ps->d_n2start, ps->d_counts);
  k_pixsel_pick<<<(nslots + 255) / 256, 256, 0, st>>>(a, nslots, ps->d_n2start, ps->d_map, ps->d_trig);
  unsigned char *trig3 = ps->d_trig + ps->cap_slots;
  k_pixsel_lvl3<<<(nsub + 3) / 4, 256, 0, st>>>(a, nsub, ps->d_n2start, ps->d_trig, ps->d_map, trig3, ps->d_counts);
  k_pixsel_lvl4<<<(nblk + 3) / 4, 256, 0, st>>>(a, nblk, ps->d_n2start, ps->d_trig, trig3, ps->d_map, ps->d_counts);
  SOS_HIP(hipGetLastError());
  int hc[3];
  SOS_HIP(hipMemcpyAsync(hc, ps->d_counts, sizeof(hc), hipMemcpyDeviceToHost, st));
  SOS_HIP(hipStreamSynchronize(st));
  n[0] = hc[0]; n[1] = hc[1]; n[2] = hc[2];
  return SOS_OK;
}

// row-major ranks of the selected pixels; charTH >= 0: sub-sample in place, returns the number dropped; else fill d_list
int do_rank(sos_pixsel *ps, int charTH, int *result) {
  hipStream_t st = ps->ctx->stream;
  const int npx = ps->w * ps->h, nb = (npx + 1023) / 1024;
  k_pixsel_count<<<nb, 1024, 0, st>>>(ps->d_map, npx, ps->d_blockcnt);
  k_pixsel_offsets<<<1, 1024, 0, st>>>(ps->d_blockcnt, nb, ps->d_counts + 4);
  SOS_HIP(hipMemsetAsync(ps->d_counts + 5, 0, sizeof(int), st));
  k_pixsel_rank<<<nb, 1024, 0, st>>>(ps->d_map, npx, ps->d_blockcnt, ps->d_pattern, charTH, ps->d_list, ps->d_counts + 5);
  SOS_HIP(hipGetLastError());
  int hc[2];
  SOS_HIP(hipMemcpyAsync(hc, ps->d_counts + 4, sizeof(hc), hipMemcpyDeviceToHost, st));
  SOS_HIP(hipStreamSynchronize(st));
  *result = charTH >= 0 ? hc[1] : hc[0];
  return SOS_OK;
}
}  // namespace

extern "C" int sos_pixsel_create(sos_ctx *c, const sos_pixsel_params *prm, const uint8_t *randomPattern, sos_pixsel **out) {
  if (!c || !prm || !randomPattern || !out) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(c->device));
  sos_pixsel *ps = new sos_pixsel();
  ps->ctx = c;
  ps->prm = *prm;
  ps->w = c->w; ps->h = c->h; ps->w32 = c->w / 32; ps->h32 = c->h / 32;
  const size_t npx = (size_t)c->w * c->h;
  const size_t nth = (size_t)ps->w32 * ps->h32 + 1;
  if (hipMalloc(&ps->d_pattern, npx) != hipSuccess || hipMalloc(&ps->d_ths, sizeof(float) * nth) != hipSuccess ||
      hipMalloc(&ps->d_thsSm, sizeof(float) * nth) != hipSuccess || hipMalloc(&ps->d_map, sizeof(float) * npx) != hipSuccess ||
      hipMalloc(&ps->d_counts, sizeof(int) * 8) != hipSuccess || hipMalloc(&ps->d_blockcnt, sizeof(int) * ((npx + 1023) / 1024 + 1)) != hipSuccess ||
      hipMalloc(&ps->d_list, sizeof(int2) * npx) != hipSuccess) {
    sos_pixsel_destroy(ps);
    return SOS_ERR_NOMEM;
  }
  SOS_HIP(hipMemcpyAsync(ps->d_pattern, randomPattern, npx, hipMemcpyHostToDevice, c->stream));
  SOS_HIP(hipMemsetAsync(ps->d_map, 0, sizeof(float) * npx, c->stream));
  SOS_HIP(hipStreamSynchronize(c->stream));
  *out = ps;
  return SOS_OK;
}

extern "C" int sos_pixsel_destroy(sos_pixsel *ps) {
  if (!ps) return SOS_OK;
  hipSetDevice(ps->ctx->device);
  hipStreamSynchronize(ps->ctx->stream);
  hipFree(ps->d_pattern); hipFree(ps->d_ths); hipFree(ps->d_thsSm); hipFree(ps->d_map); hipFree(ps->d_mask);
  hipFree(ps->d_n2start); hipFree(ps->d_trig); hipFree(ps->d_counts); hipFree(ps->d_blockcnt); hipFree(ps->d_list);
  delete ps;
  return SOS_OK;
}

extern "C" int sos_pixsel_make_hists(sos_pixsel *ps, int slot, float *ths, float *thsSmoothed) {
  if (!ps) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(ps->ctx->device));
  int rc = do_hists(ps, slot);
  if (rc) return rc;
  const size_t n = (size_t)ps->w32 * ps->h32;
  if (ths && n) SOS_HIP(hipMemcpyAsync(ths, ps->d_ths, sizeof(float) * n, hipMemcpyDeviceToHost, ps->ctx->stream));
  if (thsSmoothed && n) SOS_HIP(hipMemcpyAsync(thsSmoothed, ps->d_thsSm, sizeof(float) * n, hipMemcpyDeviceToHost, ps->ctx->stream));
  SOS_HIP(hipStreamSynchronize(ps->ctx->stream));
  return SOS_OK;
}

extern "C" int sos_pixsel_select(sos_pixsel *ps, int slot, int pot, float thFactor, float *map_out, int32_t n[3]) {
  if (!ps || !n) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(ps->ctx->device));
  if (ps->hist_slot != slot) return SOS_ERR_STATE;
  int rc = do_select(ps, slot, pot, thFactor, n);
  if (rc) return rc;
  if (map_out) {
    SOS_HIP(hipMemcpyAsync(map_out, ps->d_map, sizeof(float) * (size_t)ps->w * ps->h, hipMemcpyDeviceToHost, ps->ctx->stream));
    SOS_HIP(hipStreamSynchronize(ps->ctx->stream));
  }
  return SOS_OK;
}

extern "C" int sos_pixsel_make_maps(sos_pixsel *ps, int slot, float density, int recursionsLeft, float thFactor,
                                    int32_t *currentPotential, float *map_out, int32_t *numSelected) {
  if (!ps || !currentPotential || !numSelected || *currentPotential < 1) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(ps->ctx->device));
  int rc;
  if (ps->hist_slot != slot && (rc = do_hists(ps, slot))) return rc;  // if (fh != gradHistFrame) makeHists(fh), :195-196
  int pot = *currentPotential, idealPotential = pot;
  float numHave = 0, quotia = 0;
  const float numWant = density;
  for (;;) {  // the tail recursion of :217-243
    int32_t n[3];
    if ((rc = do_select(ps, slot, pot, thFactor, n))) return rc;
    numHave = (float)(n[0] + n[1] + n[2]);
    quotia = numWant / numHave;
    const float K = numHave * (pot + 1) * (pot + 1);
    idealPotential = (int)(sqrtf(K / numWant) - 1);
    if (idealPotential < 1) idealPotential = 1;
    if (recursionsLeft > 0 && quotia > 1.25 && pot > 1) {
      if (idealPotential >= pot) idealPotential = pot - 1;
      pot = idealPotential;
      recursionsLeft--;
      continue;
    }
    if (recursionsLeft > 0 && quotia < 0.25) {
      if (idealPotential <= pot) idealPotential = pot + 1;
      pot = idealPotential;
      recursionsLeft--;
      continue;
    }
    break;
  }
  int numHaveSub = (int)numHave;
  if (quotia < 0.95) {  // :258-275
    const unsigned char charTH = (unsigned char)(255 * quotia);
    int dropped = 0;
    if ((rc = do_rank(ps, (int)charTH, &dropped))) return rc;
    numHaveSub -= dropped;
  }
  *currentPotential = idealPotential;
  *numSelected = numHaveSub;
  if (map_out) {
    SOS_HIP(hipMemcpyAsync(map_out, ps->d_map, sizeof(float) * (size_t)ps->w * ps->h, hipMemcpyDeviceToHost, ps->ctx->stream));
    SOS_HIP(hipStreamSynchronize(ps->ctx->stream));
  }
  return SOS_OK;
}

extern "C" int sos_pixsel_list(sos_pixsel *ps, int patternPadding, int capacity, int32_t *u, int32_t *v, float *type, int32_t *count) {
  if (!ps || !count || capacity < 0 || (capacity && (!u || !v || !type))) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(ps->ctx->device));
  int total = 0;
  int rc = do_rank(ps, -1, &total);
  if (rc) return rc;
  std::vector<int2> lst((size_t)total);
  if (total) SOS_HIP(hipMemcpy(lst.data(), ps->d_list, sizeof(int2) * (size_t)total, hipMemcpyDeviceToHost));
  int k = 0;
  for (int i = 0; i < total; i++) {
    const int x = lst[i].x % ps->w, y = lst[i].x / ps->w;
    if (x < patternPadding + 1 || x >= ps->w - patternPadding - 2 || y < patternPadding + 1 || y >= ps->h - patternPadding - 2) continue;
    if (k < capacity) {
      u[k] = x;
      v[k] = y;
      memcpy(&type[k], &lst[i].y, sizeof(float));
    }
    k++;
  }
  *count = k;
  return SOS_OK;
}
