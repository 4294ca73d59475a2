// sos_common.h -- internal declarations shared by the HIP translation units of libsos_slam_hip.so.
// Target: gfx950 (MI355X, CDNA4) only.  wave = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/sos_slam.h"

// SOS_POISON=1 (debugging): every device allocation of the library is filled with 0xFF bytes (NaN as float / double, -1 as int) before it is
// used, so a kernel that reads something nobody wrote shows up in the parity tests instead of being hidden by whatever a fresh hipMalloc holds.
static inline bool sos_poison_on() {
  static const bool on = getenv("SOS_POISON") != nullptr;
  return on;
}
template <typename T>
static inline hipError_t sos_malloc_poisoned(T **p, size_t bytes) {
  const hipError_t e = (hipMalloc)(p, bytes);   // (parenthesised: not the macro below)
  if (e == hipSuccess && sos_poison_on() && bytes) {
    (void)hipMemset(*p, 0xFF, bytes);
    (void)hipDeviceSynchronize();
  }
  return e;
}
#define hipMalloc(p, n) sos_malloc_poisoned((p), (n))

#define SOS_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "[sos_slam_hip] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr,            \
              hipGetErrorString(e_));                                                          \
      return SOS_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

#define SOS_TILE 32          // residuals per Jacobian tile (one 256-thread linearize block)
#define SOS_JPLANES 72       // unique floats of a RawResidualJacobian (symmetric 2x2s stored once)
#define SOS_TILE_FLOATS (SOS_JPLANES * SOS_TILE)
#define SOS_TOPN 96          // 91 uniques of the 13x13 block, padded for the butterfly reduction

// plane numbers inside a Jacobian tile  J[tile][plane][32]
#define JP_RESF 0      // 8
#define JP_JIDX0 8     // 8
#define JP_JIDX1 16    // 8
#define JP_JAB0 24     // 8
#define JP_JAB1 32     // 8
#define JP_DXI0 40     // 6
#define JP_DXI1 46     // 6
#define JP_DC0 52      // 4
#define JP_DC1 56      // 4
#define JP_DD 60       // 2
#define JP_JIDX2 62    // 3: 00 01 11
#define JP_JABJIDX 65  // 4: 00 01 10 11
#define JP_JAB2 69     // 3: 00 01 11

// internal residual flag bits (device side, one byte per sorted residual)
#define DF_ACTIVE 1u
#define DF_LINEARIZED 2u
#define DF_ISNEW 4u
#define DF_VALID 8u  // a real residual (not tile padding)

// Tiled copy of the level-0 (I,dx,dy) image: 5 x 2 texels (120 B) per 128-byte line.  The 8-pixel pattern of a residual
// with its bilinear neighbours touches 5.7 such lines on average, against 8.0 lines of the row-major image.
#define SOS_TW 5
#define SOS_TH 2
#define SOS_TLINE 32  // floats per tile
static inline int sos_tiles_per_row(int w) { return (w + SOS_TW - 1) / SOS_TW; }
static inline size_t sos_tiled_floats(int w, int h) { return (size_t)sos_tiles_per_row(w) * ((h + SOS_TH - 1) / SOS_TH) * SOS_TLINE; }

struct sos_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int w = 0, h = 0, levels = 1;
  int wl[SOS_PYR_LEVELS] = {0}, hl[SOS_PYR_LEVELS] = {0};
  float *dI[SOS_MAX_SLOTS][SOS_PYR_LEVELS];
  float *dIt[SOS_MAX_SLOTS];  // level 0 again in SOS_TW x SOS_TH texel tiles of one 128 B line each (gather layout of the backend)
  float *absg[SOS_MAX_SLOTS][SOS_PYR_LEVELS];
  bool has_pyr[SOS_MAX_SLOTS];
  float *d_img = nullptr;     // staging for the raw image
  float *d_gammaB = nullptr;  // 256 floats
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

int sos_ctx_ensure_slot(sos_ctx *ctx, int slot, bool all_levels);
int sos_ctx_pyramid_from_staged(sos_ctx *ctx, int slot, const float *gammaB);  // makeImages from the float image in ctx->d_img

// RCCL communicator (sos_comm.hip): collectives enqueued on the caller's stream
int sos_comm_allreduce_sum_f32(sos_comm *c, float *buf, size_t count, hipStream_t st);
int sos_comm_allreduce_max_i32(sos_comm *c, int *buf, size_t count, hipStream_t st);
int sos_comm_allreduce_sum_f64(sos_comm *c, double *buf, size_t count, hipStream_t st);
int sos_comm_allgather_f32(sos_comm *c, const float *send, float *recv, size_t sendcount, hipStream_t st);
