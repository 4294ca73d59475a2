// sos_ba.hip -- device side of the OptimizationBackend (EnergyFunctional) for gfx950 / MI355X.
//
// Kernels (one per reference loop):
//   k_linearize       PointFrameResidual::linearize           FS/Residuals.cpp:77-271
//   k_apply_res       applyRes(true) (+ takeDataF, folded)    FS/Residuals.cpp:304-321
//   k_top_accumulate  AccumulatedTopHessianSSE::addPoint<0|1> OB/AccumulatedTopHessian.cpp:35-147
//   k_point_prep      per-point sums + Hdi/bdSum              OB/AccumulatedTopHessian.cpp:124-146,
//                                                             OB/AccumulatedSCHessian.cpp:34-55
//   k_sc_gram         AccumulatedSCHessianSSE::addPoint       OB/AccumulatedSCHessian.cpp:57-78  (f32 MFMA)
//   k_reduce_*        sum of the partial accumulators         OB/AccumulatedTopHessian.cpp:252-259
//   k_stitch_top/_sc  stitchDoubleInternal (fp64)             OB/AccumulatedTopHessian.cpp:231-301,
//                                                             OB/AccumulatedSCHessian.cpp:80-158
//   k_resubstitute    resubstituteFPt                         OB/EnergyFunctional.cpp:526-551
//   k_fix_lin         EFResidual::fixLinearizationF           OB/EnergyFunctionalStructs.cpp:75-103
//   k_lenergy         calcLEnergyPt                           OB/EnergyFunctional.cpp:563-624
//
// Data layout in HBM (DESIGN.md "data layout"):
//   residuals are sorted by (isLinearized, pair = host + n*target) and padded per pair to tiles of 32;
//   J[tile][72 planes][32] floats (9216 B per tile, written by ONE linearize block as one contiguous
//   span); JpJd[s][8]; everything else SoA over the sorted index s.  The fused Gauss-Newton calls
//   (linearize + applyRes in one launch, the default setting_forceAceptStep path, util/settings.cpp:117)
//   write EFResidual::J directly.  The two-step protocol -- sos_ba_linearize, then sos_ba_apply_res or
//   not (a rejected step, FS/FullSystemOptimize.cpp:387-413) -- keeps the reference's two Jacobians per
//   residual: the linearisation fills PointFrameResidual::J (d_Jnew, with its provisional JpJdF / point
//   terms) and k_commit_new moves it into EFResidual::J (d_J) for the residuals that end up IN, the swap
//   of FS/Residuals.cpp:304-321.
//
// fp32 arithmetic convention: no FMA contraction (-ffp-contract=off), correctly rounded / and sqrt
// (hipcc default), sums in the order of the reference source; the 8-pixel pattern sums are evaluated
// sequentially pixel 0..7 through a DPP row_shr chain so that IN/OOB/OUTLIER sets are bit-exact.
#include "sos_common.h"
#include "sos_devmath.h"

#include <algorithm>
#include <numeric>
#include <string>
#include <chrono>

// ------------------------------------------------------------------------------------------------
// device-visible description of a packed window
// ------------------------------------------------------------------------------------------------
struct BaDev {
  int n, P, R, Rpad, ntiles, ntilesA;
  int lin_nd;  // k_linearize2: number of leading two-tile blocks (set per launch, lin_grid)
  int w, h;
  float wM3G, hM3G;
  sos_calib calib;      // host-side copy (kernels read calibp: a device-resident Gauss-Newton loop moves the calibration itself)
  const float *calibp;  // fxl fyl cxl cyl fxli fyli cxli cyli in the staging buffer
  float huberTH, outlierTH, modeA, modeB;
  const float *img[SOS_MAX_FRAMES];
  const float *imgT[SOS_MAX_FRAMES];  // tiled level-0 copies (sos_common.h), read by k_linearize2
  int tpr;                            // tiles per image row
  const sos_precalc *precalc;
  const float *adHTdelta;
  const float *cdelta;
  sos_point *pts;
  const int *s_point, *s_orig;
  uint8_t *s_flags, *s_state, *s_newstate;
  float *s_energy, *s_newenergy, *s_newenergywo, *s_ret, *s_center, *s_rtz, *s_pterm;  // s_pterm: 8 floats / residual
  const int *t_pair;
  const float4 *t_pre;             // per tile: the sos_precalc record of its pair (7 float4 + 1 pad), refreshed with precalc
  const float *const *t_img;       // per tile: tiled level-0 image of the target frame
  const int *t_ht;                 // per tile: host idx | target idx << 16
  float *J, *JpJd;
  const int *p_begin, *p_list, *p_res_t;
  float *p_out;  // 16 floats per point
  uint8_t *o_newstate;
  float *o_newenergy, *o_newenergywo, *o_center;
  // packed per-iteration outputs (sos_ba_gn_step): per-tile energy sums, energies of the residuals that
  // target the newest frame (contiguous in the sorted order), point steps
  double *tile_esum;
  float *o_newest;
  int newest_begin, newest_count;
  const int2 *p_list2;  // per point residual list: (sorted index, xAd index = n*h + t)
  const int2 *p_list16; // the first 16 entries of every point's list at a fixed stride (x = -1: none): no p_begin lookup in front
  float4 *r_geo;        // per sorted residual: u, v, idepth_scaled, idepth_zero_scaled of its point
  const float *r_cw;    // per sorted residual: (color, weight) of its point's 8 pattern pixels, interleaved
};

#define PO_HDD_A 0
#define PO_BD_A 1
#define PO_HCD_A 2
#define PO_HDD_L 6
#define PO_BD_L 7
#define PO_HCD_L 8
#define PO_HDI 12
#define PO_BDSUM 13
#define PO_IDH 14
#define PO_STEP 15

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// ((((((v0+v1)+v2)+v3)+v4)+v5)+v6)+v7 over the 8 lanes of a pattern group (the reference's sequential
// pixel loop, FS/Residuals.cpp:177-243); valid in lane 7 of the group.  One fused v_add_f32_dpp per term:
// lane i adds lane (i-k)'s value of its 16-lane row (row_shr:k, 0 beyond the row start).  hipcc does not fuse
// __builtin_amdgcn_update_dpp into the add here (it emits mov 0 + mov_dpp + add), hence the asm block; the
// leading s_nop covers the VALU-write -> DPP-read hazard on `v` (2 wait states).
__device__ __forceinline__ float seqsum8(float v) {
  float s;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %2 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_e32 %0, %0, %1"
      : "=&v"(s)
      : "v"(v), "v"(0.0f));
  return s;
}

// ================================================================================================
// k_linearize: one 256-thread block = one tile of 32 residuals of ONE (host,target) pair; 8 lanes per
// residual, one lane per pattern pixel; lane 7 of each group additionally does the per-residual part
// (FEJ centre projection, geometric Jacobians, classification, JpJdF).  The kernel is latency-bound at
// window sizes of 10^4..10^5 residuals (a handful of blocks per CU), so the design minimises the number
// of DEPENDENT memory levels: the point data is replicated per residual in sorted order (r_geo / r_cw,
// exactly the 80 B "point record" of the algorithmic byte count), so the chain is
//   {r_geo[s], r_cw[s]}  ||  {tile -> pair -> precalc (scalar loads)}  ->  projection -> 4 image taps.
// The staged 72 x 32 tile leaves as one contiguous 9216-byte span of J.
// ================================================================================================
#define SJ_STRIDE 40  // LDS row stride (floats): == 8 mod 32 -> the 8x8 (pixel, residual) stores are 2-way at worst

// ------------------------------------------------------------------------------------------------
// Completion signal of a kernel to the host without a stream marker: every block, after its last host-visible
// store, fences at system scope and bumps a device counter; the block that completes the launch (the counter is
// cumulative: target = launches so far x blocks) stores the launch sequence number into a flag in device-mapped
// pinned host memory.  The host polls that flag instead of waiting on an event / the stream, which removes the
// marker packet (~6 us of bubble in front of the next kernel) and the runtime's wake-up latency.
// ------------------------------------------------------------------------------------------------
// k_publish: the one-thread kernel variant of the same idea -- queued behind the producing kernel, it starts only
// after that kernel's end-of-kernel release, so its store of the sequence number tells the polling host that the
// producer's results are visible (no per-block fences inside the producer).
__global__ void k_publish(int *flag, int seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
struct DoneSignal {
  int *ctr;    // device counter (nullptr = no signalling)
  int *flag;   // device address of the mapped host flag
  int target;  // counter value that completes this launch
  int seq;     // value to publish
};
__device__ __forceinline__ void signal_block_done(const DoneSignal &sg) {  // call from ONE thread of the block
  __threadfence_system();
  const int old = atomicAdd(sg.ctr, 1);
  if (old == sg.target - 1) {
    __threadfence_system();
    __hip_atomic_store(sg.flag, sg.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------------
// The 96 per-residual values of AccumulatedTopHessianSSE::addPoint (OB/AccumulatedTopHessian.cpp:112-122 ->
// AccumulatorApprox::update / updateTopRight / updateBotRight, OB/MatrixAccumulators.h:928-1112) in the order of
// the packed 13x13 block: 55 uniques of the 10x10 [C xi] block, 30 top-right entries, 6 bottom-right, the
// residual count, 4 zeros.  top_value<K> is the K-th of them, with the same expression shapes as
// top_accumulate_body so that both produce identical per-residual terms.
// ------------------------------------------------------------------------------------------------
struct TopIn {
  float x[10], y[10];  // Jpdc[0..3] | Jpdxi[0..5], rows 0 and 1
  float a, b, c;       // JIdx2 00 01 11
  float jab00, jab01, jab10, jab11, ab00, ab01, ab11;
  float JI_r0, JI_r1, Jab_r0, Jab_r1, rr;
};
__host__ __device__ constexpr int top_row_of(int k) {
  int r = 0, rem = k;
  while (rem >= 10 - r) { rem -= 10 - r; r++; }
  return r;
}
__host__ __device__ constexpr int top_col_of(int k) {
  int r = 0, rem = k;
  while (rem >= 10 - r) { rem -= 10 - r; r++; }
  return r + rem;
}
template <int K>
__device__ __forceinline__ float top_value(const TopIn &in) {
  if constexpr (K < 55) {
    constexpr int rr_ = top_row_of(K), cc = top_col_of(K);
    return in.a * in.x[cc] * in.x[rr_] + in.c * in.y[cc] * in.y[rr_] + in.b * (in.x[cc] * in.y[rr_] + in.y[cc] * in.x[rr_]);
  } else if constexpr (K < 85) {
    constexpr int i = (K - 55) / 3, w = (K - 55) % 3;
    if constexpr (w == 0) return in.x[i] * in.jab00 + in.y[i] * in.jab01;
    else if constexpr (w == 1) return in.x[i] * in.jab10 + in.y[i] * in.jab11;
    else return in.x[i] * in.JI_r0 + in.y[i] * in.JI_r1;
  } else if constexpr (K == 85) return in.ab00;
  else if constexpr (K == 86) return in.ab01;
  else if constexpr (K == 87) return in.Jab_r0;
  else if constexpr (K == 88) return in.ab11;
  else if constexpr (K == 89) return in.Jab_r1;
  else if constexpr (K == 90) return in.rr;
  else if constexpr (K == 91) return 1.f;
  else return 0.f;
}
template <int P>
__device__ __forceinline__ void top_values12(const TopIn &in, float *v) {
  v[0] = top_value<12 * P + 0>(in); v[1] = top_value<12 * P + 1>(in); v[2] = top_value<12 * P + 2>(in);
  v[3] = top_value<12 * P + 3>(in); v[4] = top_value<12 * P + 4>(in); v[5] = top_value<12 * P + 5>(in);
  v[6] = top_value<12 * P + 6>(in); v[7] = top_value<12 * P + 7>(in); v[8] = top_value<12 * P + 8>(in);
  v[9] = top_value<12 * P + 9>(in); v[10] = top_value<12 * P + 10>(in); v[11] = top_value<12 * P + 11>(in);
}

#ifndef SOS_LIN_WAVES
#define SOS_LIN_WAVES 4
#endif
#ifdef SOS_LIN_PROFILE  // developer build: per-block phase timestamps (s_memtime) of k_linearize
__device__ unsigned long long g_lin_prof[8192 * 8];
#ifdef SOS_LIN_PROFILE_WALL  // 100 MHz constant clock, common to all XCDs: block start / end spread over the chip
#define LIN_CLOCK() wall_clock64()
#else
#define LIN_CLOCK() clock64()
#endif
#define LIN_STAMP(i) do { if (tid == 0 && blockIdx.x < 8192) g_lin_prof[blockIdx.x * 8 + (i)] = LIN_CLOCK(); } while (0)
extern "C" int sos_debug_lin_prof(unsigned long long *out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin_prof), sizeof(unsigned long long) * 8 * nblocks) == hipSuccess ? 0 : -2;
}
#else
#define LIN_STAMP(i)
#endif
template <int HALF>
__device__ __forceinline__ void bfly_step(float *v, int m, bool hi);

// ================================================================================================
// k_linearize2: the production form of the kernel above, restructured around the finding that the kernel is bound
// by VALU issue, not by memory (rocprof: every wave of k_linearize spends a third of its instructions in the
// per-residual "leader" part with 1 of 8 lanes active).  One 512-thread block = two tiles; three phases:
//   1  all 8 waves, lane = pattern pixel: projection, image taps, photometric rows, the 8-pixel sequential sums
//      (DPP chains); lane 7 of each group leaves the 17 sums of its residual in LDS
//   2  ONE wave (rotating with the block index so the work spreads over the SIMDs), lane = residual, all 64 lanes
//      busy: FEJ centre projection, geometric Jacobians, classification, applyRes, JpJdF, point terms and -- when
//      fuse_top is given -- the 13x13 block sums of AccumulatedTopHessianSSE::addPoint<0> over each tile
//      (transposed butterfly over the 32 lanes of a tile, 24 values at a time)
//   3  all waves: coalesced store of the two staged tiles (skipped when fused: the tiles never leave the chip)
// Per-element arithmetic is the same expression for expression as in k_linearize: outputs are bit-identical
// (tests/test_gpu_backend.py), only the tile sums of the fused mode use a different (still fixed) summation tree.
// ================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define L2_TILES 2
#define L2_NS 18  // 17 sums + group-out-of-bounds flag
#define L2_LR_PST 68                          // row pitch of an operand plane: 64 rows (row type x residual) + 4
#define L2_LR_TILE (2 * 16 * L2_LR_PST + 32)  // L and R planes of one tile; + 32 floats so the two tiles hit different banks
// FUSED selects the pipelined form (fuse_top given, doApply = 1, tiles never stored) at compile time: two kernels with
// their own names in a profile and their own register allocation
// SCALAR1: per-tile constants of phase 1 through the scalar cache (below); chosen per launch
template <bool FUSED, bool SCALAR1>
__global__ __launch_bounds__(256 * L2_TILES, 6) void k_linearize2(const float4 *__restrict__ a_r_geo, const float *__restrict__ a_r_cw,
                                                               const float4 *__restrict__ a_t_pre, const float *const *__restrict__ a_t_img,
                                                               int a_ntilesA, int a_lin_nd_x, BaDev d, const float *__restrict__ frameTH,
                                                               int doApply, float *__restrict__ fuse_top_) {
  // the leading scalar arguments (what the first loads of a block need) arrive preloaded in SGPRs
  // (-amdgpu-kernarg-preload-count): the block does not wait for the kernel-argument segment before its first requests
  float *__restrict__ const fuse_top = FUSED ? fuse_top_ : nullptr;
  __builtin_assume(!FUSED || fuse_top != nullptr);
  if (FUSED) doApply = 1;
  // one LDS arena: the tile staging [2][72 planes][40], and -- in fused mode, before phase 2 touches the tiles -- the
  // transposition buffer of the 17 x 8 addends per residual: [residual 0..63][sum 0..16][pixel 0..7]
  __shared__ __attribute__((aligned(16))) float sBig[32 * L2_TILES * 17 * 8];
  static_assert(32 * L2_TILES * 17 * 8 >= L2_TILES * SOS_JPLANES * SJ_STRIDE, "arena holds the tile staging");
  static_assert(32 * L2_TILES * 17 * 8 >= L2_TILES * L2_LR_TILE, "arena holds the MFMA operand planes");
  float(*sJ2)[SOS_JPLANES * SJ_STRIDE] = reinterpret_cast<float(*)[SOS_JPLANES * SJ_STRIDE]>(sBig);
  __shared__ float sS[L2_NS][32 * L2_TILES];
  __shared__ unsigned int sLin2[L2_TILES];
  const int tid = threadIdx.x;
  const int tloc = tid >> 8, t256 = tid & 255;
  // blocks [0, lin_nd) own two tiles, the blocks behind them one (the second half idles): with an odd number of tiles per
  // CU the grid is cut so that every CU gets the same number of tiles instead of whole two-tile blocks
  // XCD-contiguous tile ranges (SOS_LIN_XCD=1, off by default): the dispatcher deals block b to XCD b % 8 (observed, MI355X_MICROARCH.md; a speed
  // assumption only), and the tiles are sorted by (target, host) -- dealt round-robin, the hosts of one target image land on all eight
  // XCDs and each L2 fetches its own copy of the texel lines they share (a fifth of the texel traffic at W12).  With the switch XCD x
  // owns the CONTIGUOUS tile range [start_x, start_x + 2 nd_x + ns_x): its two-tile blocks first, then its one-tile blocks -- the same
  // number of either as under the round-robin deal, so every CU still gets the tiles lin_grid meant for it.  Which block computes a
  // tile changes, nothing a tile or a residual computes does: results are bit-identical (tests/test_gpu_variants.py).
  // (the switch rides in bit 30 of the preloaded block count: nothing of this waits for the kernel-argument segment)
  const int a_lin_nd = a_lin_nd_x & 0x3fffffff;
  int tbase_;
  if (a_lin_nd_x >> 30) {
    const int b = (int)blockIdx.x, nb = a_ntilesA - a_lin_nd, x = b & 7, k = b >> 3;  // (lin_grid: tiles = 2 nd + (nb - nd); no gridDim load)
    const int Dx = (a_lin_nd >> 3) * x + min(a_lin_nd & 7, x), Nx = (nb >> 3) * x + min(nb & 7, x);  // two-tile / all blocks of the XCDs before x
    const int ndx = a_lin_nd > x ? (a_lin_nd - x + 7) >> 3 : 0;                                         // two-tile blocks of XCD x
    tbase_ = Dx + Nx + (b < a_lin_nd ? 2 * k : 2 * ndx + (k - ndx));
  } else {
    tbase_ = (int)blockIdx.x < a_lin_nd ? (int)blockIdx.x * L2_TILES : a_lin_nd + (int)blockIdx.x;
  }
  const int tbase = tbase_;
  const int tlim = (int)blockIdx.x < a_lin_nd ? a_ntilesA : min(a_ntilesA, tbase + 1);
  const int tile = tbase + tloc;
  const bool tile_ok = tile < tlim;
  const int rl = t256 >> 3, idx = t256 & 7;
  const int lane = tid & 63, wave = tid >> 6;
  const int w2 = blockIdx.x & 7;  // the wave of this block that runs phase 2
  float *sJ = sJ2[tloc];
  if (tid < L2_TILES) sLin2[tid] = 0;
  LIN_STAMP(0);

  // ---- phase-1 first-level loads are issued FIRST: memory returns are counted in issue order, so whatever is requested
  // ahead of them delays the projection (measured: 3.0 k -> 1.1 k cycles to the first use when nothing precedes them)
  const int tile1 = tile_ok ? tile : 0;
  const int s1 = tile1 * SOS_TILE + rl;
  const float4 geo = a_r_geo[s1];
  const float2 cw = reinterpret_cast<const float2 *>(a_r_cw)[8 * (size_t)s1 + idx];  // one 8-byte load per pixel lane
  const float color = cw.x, pweight = cw.y;
  // the tile is the same for all lanes of a wave (256 threads per tile): made explicit, its precalc record and image
  // pointer come through the scalar cache instead of as 64-lane broadcasts through the vector memory pipeline
  // (7 -> 3 vector loads per wave in front of the projection).  Measured: -4 % when the grid is one round of resident
  // blocks that all start together and crowd the vector memory pipeline (W12: 10.0 -> 9.6 us), +6 % when blocks start
  // one by one as others retire (W16: 21.7 -> 23.1 us, the scalar cache's miss path is the slower one then) -- hence a
  // per-launch choice.  The same for the phase-2 operands, two tiles per wave selected per lane, gave nothing.
  const int tile1u = SCALAR1 ? __builtin_amdgcn_readfirstlane(tile1) : tile1;
  const sos_precalc *pc = reinterpret_cast<const sos_precalc *>(a_t_pre + 8 * (size_t)tile1u);  // tile-indexed: no dependent load
  const float *__restrict__ img = a_t_img[tile1u];
  float krk[9], ktt[3];
#pragma unroll
  for (int i = 0; i < 9; i++) krk[i] = pc->PRE_KRKiTll[i];
#pragma unroll
  for (int i = 0; i < 3; i++) ktt[i] = pc->PRE_KtTll[i];

  // ---- phase-2 operands of this lane's residual are requested next so their latency hides behind phase 1
  const int r64 = lane, tl2 = r64 >> 5, rr2 = r64 & 31;
  const int tile2 = tbase + tl2;
  const int role = (wave - w2) & 7;  // phase-2 role of this wave: 0 in stored-tile mode, 0..3 in fused mode
  const bool p2 = role < (fuse_top ? 4 : 1) && tile2 < tlim;
  const int s2 = (p2 ? tile2 : 0) * SOS_TILE + rr2;
  float4 geo2 = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned flags2 = 0;
  int st2 = 0, pair2 = 0;
  float R0[9], t0[3], eOld2 = 0.f, neOld2 = 0.f, th2 = 0.f;
  int orig2 = -1;
  if (p2) {
    geo2 = a_r_geo[s2];
    flags2 = d.s_flags[s2];
    st2 = d.s_state[s2];
    pair2 = d.t_ht[tile2];  // host idx | target idx << 16
    eOld2 = d.s_energy[s2];
    neOld2 = d.s_newenergy[s2];
    if (role == 0) orig2 = d.s_orig[s2];
    const sos_precalc *pc2 = reinterpret_cast<const sos_precalc *>(a_t_pre + 8 * (size_t)tile2);
#pragma unroll
    for (int i = 0; i < 9; i++) R0[i] = pc2->PRE_RTll_0[i];
#pragma unroll
    for (int i = 0; i < 3; i++) t0[i] = pc2->PRE_tTll_0[i];
  }

  // =============================== phase 1: lane = pattern pixel ===============================
  if (tile_ok) {
    const float pu = geo.x, pv = geo.y, id = geo.z;

    // this lane's pattern pixel with the current pose / idepth (FS/ResidualProjections.h:43-50)
    const int px = (int)((0x21420312u >> (4 * idx)) & 0xf) - 2;  // {0,-1,1,-2,0,2,-1,0}
    const int py = (int)((0x43222110u >> (4 * idx)) & 0xf) - 2;  // {-2,-1,-1,0,0,0,1,2}
    const float u_pt = pu + (float)px, v_pt = pv + (float)py;
    const float q0 = krk[0] * u_pt + krk[1] * v_pt + krk[2] + ktt[0] * id;
    const float q1 = krk[3] * u_pt + krk[4] * v_pt + krk[5] + ktt[1] * id;
    const float q2 = krk[6] * u_pt + krk[7] * v_pt + krk[8] + ktt[2] * id;
    const float Ku = q0 / q2, Kv = q1 / q2;
#ifdef SOS_LIN_PROFILE_HWID  // where the block runs: HW_ID (cu 11:8, sh 12, se 15:13) | XCC_ID << 32
    if (tid == 0 && blockIdx.x < 8192)
      g_lin_prof[blockIdx.x * 8 + 1] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);
#else
    LIN_STAMP(1);
#endif
    const bool inb = Ku > 1.1f && Kv > 1.1f && Ku < d.wM3G && Kv < d.hM3G;

    // bilinear (I,dx,dy) tap (util/globalFuncs.h:68-82); addresses clamped so the loads are always legal
    int ix = (int)Ku, iy = (int)Kv;
    const float fdx = Ku - (float)ix, fdy = Kv - (float)iy;
    ix = min(max(ix, 0), d.w - 2);
    iy = min(max(iy, 0), d.h - 2);
    // the four texels in the tiled copy: tile (ix / 5, iy / 2), 128 B per tile, (I,dx,dy) of texel (x, y) at 3 (5 y + x)
    const unsigned qx0 = (unsigned)ix / SOS_TW, rx0 = (unsigned)ix - SOS_TW * qx0;
    const bool wrap = rx0 == SOS_TW - 1;
    const unsigned qx1 = wrap ? qx0 + 1 : qx0, rx1 = wrap ? 0u : rx0 + 1;
    const unsigned ty0 = (unsigned)iy >> 1, ry0 = (unsigned)iy & 1u;
    const unsigned ty1 = ty0 + ry0, ry1 = ry0 ^ 1u;
    const unsigned rowA = ty0 * (unsigned)d.tpr, rowC = ty1 * (unsigned)d.tpr;
    const float *pa = img + (size_t)(rowA + qx0) * SOS_TLINE + 3 * (SOS_TW * ry0 + rx0);
    const float *pb = img + (size_t)(rowA + qx1) * SOS_TLINE + 3 * (SOS_TW * ry0 + rx1);
    const float *pc_ = img + (size_t)(rowC + qx0) * SOS_TLINE + 3 * (SOS_TW * ry1 + rx0);
    const float *pd = img + (size_t)(rowC + qx1) * SOS_TLINE + 3 * (SOS_TW * ry1 + rx1);
    const float a0 = pa[0], a1 = pa[1], a2 = pa[2], b0_ = pb[0], b1_ = pb[1], b2_ = pb[2];
    const float c0 = pc_[0], c1 = pc_[1], c2 = pc_[2], d0 = pd[0], d1 = pd[1], d2 = pd[2];
    const float dxdy = fdx * fdy;
    const float w11 = dxdy, w01 = fdy - dxdy, w10 = fdx - dxdy, w00 = 1 - fdx - fdy + dxdy;
    const float hit0 = w11 * d0 + w01 * c0 + w10 * b0_ + w00 * a0;
    float hit1 = w11 * d1 + w01 * c1 + w10 * b1_ + w00 * a1;
    float hit2 = w11 * d2 + w01 * c2 + w10 * b2_ + w00 * a2;

    LIN_STAMP(2);
    const bool lane_oob = !inb || !isfinite(hit0);
    const unsigned long long oobmask = __ballot(lane_oob);
    const bool grp_oob = ((oobmask >> (lane & 56)) & 0xffull) != 0;

    // photometric residual, weights (FS/Residuals.cpp:189-241)
    const float affLL0 = pc->PRE_aff_mode[0], affLL1 = pc->PRE_aff_mode[1], b0 = pc->PRE_b0_mode;
    const float residual = hit0 - (float)(affLL0 * color + affLL1);
    const float drdA = color - b0;
#ifdef SOS_EXP_CHEAP
    float wgt = __builtin_amdgcn_rsqf((d.outlierTH + (hit1 * hit1 + hit2 * hit2)) * (1.0f / 2500.f));
    wgt = 0.5f * (wgt + pweight);
    float hw = fabsf(residual) < d.huberTH ? 1 : d.huberTH * __builtin_amdgcn_rcpf(fabsf(residual));
    const float e_i = wgt * wgt * hw * residual * residual * (2 - hw);
    if (hw < 1) hw = __builtin_amdgcn_sqrtf(hw);
#else
    float wgt = sqrtf(d.outlierTH / (d.outlierTH + (hit1 * hit1 + hit2 * hit2)));
    wgt = 0.5f * (wgt + pweight);
    float hw = fabsf(residual) < d.huberTH ? 1 : d.huberTH / fabsf(residual);
    const float e_i = wgt * wgt * hw * residual * residual * (2 - hw);
    if (hw < 1) hw = sqrtf(hw);
#endif
    hw = hw * wgt;
    hit1 *= hw;
    hit2 *= hw;
    const float jabF0 = d.modeA < 0 ? 0.0f : drdA * hw, jabF1 = d.modeB < 0 ? 0.0f : hw;

    if (!fuse_top) {  // per-pixel rows of the Jacobian -> LDS staging (only needed when the tile is stored)
      sJ[(JP_RESF + idx) * SJ_STRIDE + rl] = residual * hw;
      sJ[(JP_JIDX0 + idx) * SJ_STRIDE + rl] = hit1;
      sJ[(JP_JIDX1 + idx) * SJ_STRIDE + rl] = hit2;
      sJ[(JP_JAB0 + idx) * SJ_STRIDE + rl] = jabF0;
      sJ[(JP_JAB1 + idx) * SJ_STRIDE + rl] = jabF1;
    }

    if (fuse_top) {
      // Fused mode: the seventeen 8-pixel sums are not formed by DPP chains (7 dependent adds x 17 per wave, a third of
      // this phase's instructions) but transposed through LDS: every pixel lane drops its 17 addends, then one thread
      // per (residual, sum) adds the 8 addends in pattern order -- the same left-to-right sum, bit for bit.
      float *ad = sBig + ((size_t)(tloc * 32 + rl) * 17) * 8 + idx;  // bank = (8 c + idx) mod 64: conflict-free
      ad[0 * 8] = e_i;
      ad[1 * 8] = hit1 * hit1;
      ad[2 * 8] = hit2 * hit2;
      ad[3 * 8] = hit1 * hit2;
      ad[4 * 8] = drdA * hw * hit1;
      ad[5 * 8] = drdA * hw * hit2;
      ad[6 * 8] = hw * hit1;
      ad[7 * 8] = hw * hit2;
      ad[8 * 8] = drdA * drdA * hw * hw;
      ad[9 * 8] = drdA * hw * hw;
      ad[10 * 8] = hw * hw;
      ad[11 * 8] = hw * hw * (hit1 * hit1 + hit2 * hit2);
      ad[12 * 8] = residual * hw * hit1;
      ad[13 * 8] = residual * hw * hit2;
      ad[14 * 8] = residual * hw * jabF0;
      ad[15 * 8] = residual * hw * jabF1;
      ad[16 * 8] = residual * hw * (residual * hw);
      if (idx == 7) sS[17][tloc * 32 + rl] = grp_oob ? 1.f : 0.f;
    } else {
    float sm[17];
    sm[0] = seqsum8(e_i);                                  // energyLeft
    sm[1] = seqsum8(hit1 * hit1);                          // JIdxJIdx_00
    sm[2] = seqsum8(hit2 * hit2);                          // JIdxJIdx_11
    sm[3] = seqsum8(hit1 * hit2);                          // JIdxJIdx_10
    sm[4] = seqsum8(drdA * hw * hit1);                     // JabJIdx_00
    sm[5] = seqsum8(drdA * hw * hit2);                     // JabJIdx_01
    sm[6] = seqsum8(hw * hit1);                            // JabJIdx_10
    sm[7] = seqsum8(hw * hit2);                            // JabJIdx_11
    sm[8] = seqsum8(drdA * drdA * hw * hw);                // JabJab_00
    sm[9] = seqsum8(drdA * hw * hw);                       // JabJab_01
    sm[10] = seqsum8(hw * hw);                             // JabJab_11
    sm[11] = seqsum8(hw * hw * (hit1 * hit1 + hit2 * hit2));  // wJI2_sum
    sm[12] = seqsum8(residual * hw * hit1);                // JI_r0 (OB/AccumulatedTopHessian.cpp:101-110)
    sm[13] = seqsum8(residual * hw * hit2);                // JI_r1
    sm[14] = seqsum8(residual * hw * jabF0);               // Jab_r0
    sm[15] = seqsum8(residual * hw * jabF1);               // Jab_r1
    sm[16] = seqsum8(residual * hw * (residual * hw));     // rr
    if (idx == 7) {
      const int c = tloc * 32 + rl;
#pragma unroll
      for (int k = 0; k < 17; k++) sS[k][c] = sm[k];
      sS[17][c] = grp_oob ? 1.f : 0.f;
    }
    }
  }
  if (p2) th2 = fmaxf(frameTH[pair2 & 0xffff], frameTH[pair2 >> 16]);  // second-level load: returns under the barrier and the sums
  LIN_STAMP(3);
  __syncthreads();
  LIN_STAMP(4);
  if (fuse_top) {  // the sums: thread = (residual c, sum k), eight addends in pattern order 0..7
    for (int item = tid; item < 32 * L2_TILES * 17; item += 256 * L2_TILES) {
      const float4 a0 = *reinterpret_cast<const float4 *>(sBig + (size_t)item * 8);
      const float4 a1 = *reinterpret_cast<const float4 *>(sBig + (size_t)item * 8 + 4);
      float acc = 0.0f + a0.x;  // seqsum8 starts from 0 + p0 as well
      acc = acc + a0.y; acc = acc + a0.z; acc = acc + a0.w;
      acc = acc + a1.x; acc = acc + a1.y; acc = acc + a1.z; acc = acc + a1.w;
      const int c = item / 17, k = item - c * 17;
      sS[k][c] = acc;
    }
    __syncthreads();
  }

  LIN_STAMP(5);
  // =============================== phase 2: lane = residual ===============================
  // Stored-tile mode: one wave does everything.  Fused mode: four waves on different SIMDs share the work by ROLE
  // (each repeats the short centre projection / classification it needs, so none waits for another):
  //   role 0  classification, commit of states / energies / centre, host outputs, tile energy sums
  //   role 1  geometric Jacobians -> JpJdF and the per-point terms
  //   role 2  geometric Jacobians -> L operand rows of the block sums, residual count
  //   role 3  geometric Jacobians -> R operand rows
  if (p2) {
    float *sJr = sJ2[tl2];
    const int s = s2, rl_ = rr2, c = r64;
    const unsigned flags = flags2;
    const int st = st2;
    const int tIdx = pair2 >> 16;
    const bool fused = fuse_top != nullptr;
    const bool doCommit = role == 0, doJp = fused ? role == 1 : true, doL = fused && role == 2, doR = fused && role == 3;
    const bool valid = (flags & DF_VALID) != 0;
    const bool isLin = valid && (flags & DF_LINEARIZED);
    const float pu = geo2.x, pv = geo2.y, idz = geo2.w;
    const float energyLeft0 = sS[0][c], JIdxJIdx_00 = sS[1][c], JIdxJIdx_11 = sS[2][c], JIdxJIdx_10 = sS[3][c];
    const float wJI2_sum = sS[11][c];
    const bool grp_oob = sS[17][c] != 0.f;

    const float fxl = d.calibp[0], fyl = d.calibp[1], cxl = d.calibp[2], cyl = d.calibp[3];
    const float fxli = d.calibp[4], fyli = d.calibp[5];
    // ---- centre projection with the FEJ pose / idepth (FS/ResidualProjections.h:52-73)
    const float KliP0 = (pu - cxl) * fxli;
    const float KliP1 = (pv - cyl) * fyli;
    const float ptp0 = R0[0] * KliP0 + R0[1] * KliP1 + R0[2] + t0[0] * idz;
    const float ptp1 = R0[3] * KliP0 + R0[4] * KliP1 + R0[5] + t0[1] * idz;
    const float ptp2 = R0[6] * KliP0 + R0[7] * KliP1 + R0[8] + t0[2] * idz;
    const float drescale = 1.0f / ptp2;
    const float new_idepth = idz * drescale;
    const float cu = ptp0 * drescale, cv = ptp1 * drescale;
    const float cKu = cu * fxl + cxl, cKv = cv * fyl + cyl;
    const bool center_ok = (drescale > 0) && cKu > 1.1f && cKv > 1.1f && cKu < d.wM3G && cKv < d.hM3G;

    // ---- classification (FS/Residuals.cpp:78-83,107-112,258-270)
    int newState;
    float newEnergy = 0.f, newEnergyWO = -1.f, ret;
    if (!valid) {
      newState = SOS_RES_OOB;
      ret = 0.f;
    } else if (isLin) {  // not in activeResiduals (FS/FullSystemOptimize.cpp:321)
      newState = st;
      ret = 0.f;
      newEnergy = neOld2;
      if (doCommit) atomicOr(&sLin2[tl2], 1u << rl_);
    } else if (st == SOS_RES_OOB || !center_ok || grp_oob) {
      newState = SOS_RES_OOB;
      ret = eOld2;
      newEnergy = neOld2;
    } else {
      float energyLeft = energyLeft0;
      newEnergyWO = energyLeft;
      const float th = th2;  // max(frameEnergyTH[host], frameEnergyTH[target]), requested up front
      if (energyLeft > th || wJI2_sum < 2) {
        energyLeft = th;
        newState = SOS_RES_OUTLIER;
      } else {
        newState = SOS_RES_IN;
      }
      newEnergy = energyLeft;
      ret = energyLeft;
    }
    const bool wr = doApply != 2;  // 2 = refresh: recompute the tile at the unchanged state and store nothing but J
    bool activeAfter = (flags & DF_ACTIVE) != 0;
    const bool applies = doApply == 1 && valid && !isLin && st != SOS_RES_OOB;  // applyRes(true), FS/Residuals.cpp:304-321
    if (applies) activeAfter = newState == SOS_RES_IN;
    const bool use = valid && !isLin && activeAfter;

    if (doCommit) {
      if (wr) {
        d.s_newstate[s] = (uint8_t)newState;
        d.s_newenergy[s] = newEnergy;
        d.s_newenergywo[s] = newEnergyWO;
        d.s_ret[s] = ret;
        if (d.o_newest && tIdx == d.n - 1) d.o_newest[s - d.newest_begin] = newEnergyWO;
      }
      if (applies) {
        d.s_flags[s] = (uint8_t)(activeAfter ? (flags | DF_ACTIVE) : (flags & ~DF_ACTIVE));
        d.s_state[s] = (uint8_t)newState;
        d.s_energy[s] = newEnergy;
      }
      const bool wrote_center = wr && valid && !isLin && st != SOS_RES_OOB && center_ok;
      if (wrote_center) {
        d.s_center[3 * s + 0] = cKu;
        d.s_center[3 * s + 1] = cKv;
        d.s_center[3 * s + 2] = new_idepth;
      }
      const int orig = orig2;
      if (wr && orig >= 0) {
        if (d.o_newstate) d.o_newstate[orig] = (uint8_t)newState;
        if (d.o_newenergy) d.o_newenergy[orig] = newEnergy;
        if (d.o_newenergywo) d.o_newenergywo[orig] = newEnergyWO;
        if (d.o_center && wrote_center) {
          d.o_center[3 * orig + 0] = cKu;
          d.o_center[3 * orig + 1] = cKv;
          d.o_center[3 * orig + 2] = new_idepth;
        }
      }
      if (d.tile_esum && wr) {  // returned energies of the tile: fp64 butterfly over its 32 residuals
        double a = (double)ret;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (rl_ == 0) d.tile_esum[tile2] = a;
      }
    }

    if (doJp || doL || doR) {
      // ---- geometric Jacobians (FS/Residuals.cpp:116-157)
      const float d_d_x = drescale * (t0[0] - t0[2] * cu) * SOS_SCALE_IDEPTH * fxl;
      const float d_d_y = drescale * (t0[1] - t0[2] * cv) * SOS_SCALE_IDEPTH * fyl;
      float dCx2 = drescale * (R0[6] * cu - R0[0]);
      float dCx3 = fxl * drescale * (R0[7] * cu - R0[1]) * fyli;
      float dCx0 = KliP0 * dCx2;
      float dCx1 = KliP1 * dCx3;
      float dCy2 = fyl * drescale * (R0[6] * cv - R0[3]) * fxli;
      float dCy3 = drescale * (R0[7] * cv - R0[4]);
      float dCy0 = KliP0 * dCy2;
      float dCy1 = KliP1 * dCy3;
      dCx0 = (dCx0 + cu) * SOS_SCALE_F;
      dCx1 *= SOS_SCALE_F;
      dCx2 = (dCx2 + 1) * SOS_SCALE_C;
      dCx3 *= SOS_SCALE_C;
      dCy0 *= SOS_SCALE_F;
      dCy1 = (dCy1 + cv) * SOS_SCALE_F;
      dCy2 *= SOS_SCALE_C;
      dCy3 = (dCy3 + 1) * SOS_SCALE_C;
      const float dxi_x[6] = {new_idepth * fxl, 0.0f, -new_idepth * cu * fxl, -cu * cv * fxl, (1 + cu * cu) * fxl, -cv * fxl};
      const float dxi_y[6] = {0.0f, new_idepth * fyl, -new_idepth * cv * fyl, -(1 + cv * cv) * fyl, cu * cv * fyl, cu * fyl};

      if (!fused) {  // per-residual planes of the stored tile
#pragma unroll
        for (int i = 0; i < 6; i++) {
          sJr[(JP_DXI0 + i) * SJ_STRIDE + rl_] = dxi_x[i];
          sJr[(JP_DXI1 + i) * SJ_STRIDE + rl_] = dxi_y[i];
        }
        sJr[(JP_DC0 + 0) * SJ_STRIDE + rl_] = dCx0;
        sJr[(JP_DC0 + 1) * SJ_STRIDE + rl_] = dCx1;
        sJr[(JP_DC0 + 2) * SJ_STRIDE + rl_] = dCx2;
        sJr[(JP_DC0 + 3) * SJ_STRIDE + rl_] = dCx3;
        sJr[(JP_DC1 + 0) * SJ_STRIDE + rl_] = dCy0;
        sJr[(JP_DC1 + 1) * SJ_STRIDE + rl_] = dCy1;
        sJr[(JP_DC1 + 2) * SJ_STRIDE + rl_] = dCy2;
        sJr[(JP_DC1 + 3) * SJ_STRIDE + rl_] = dCy3;
        sJr[(JP_DD + 0) * SJ_STRIDE + rl_] = d_d_x;
        sJr[(JP_DD + 1) * SJ_STRIDE + rl_] = d_d_y;
        sJr[(JP_JIDX2 + 0) * SJ_STRIDE + rl_] = JIdxJIdx_00;
        sJr[(JP_JIDX2 + 1) * SJ_STRIDE + rl_] = JIdxJIdx_10;
        sJr[(JP_JIDX2 + 2) * SJ_STRIDE + rl_] = JIdxJIdx_11;
        sJr[(JP_JABJIDX + 0) * SJ_STRIDE + rl_] = sS[4][c];
        sJr[(JP_JABJIDX + 1) * SJ_STRIDE + rl_] = sS[5][c];
        sJr[(JP_JABJIDX + 2) * SJ_STRIDE + rl_] = sS[6][c];
        sJr[(JP_JABJIDX + 3) * SJ_STRIDE + rl_] = sS[7][c];
        sJr[(JP_JAB2 + 0) * SJ_STRIDE + rl_] = sS[8][c];
        sJr[(JP_JAB2 + 1) * SJ_STRIDE + rl_] = sS[9][c];
        sJr[(JP_JAB2 + 2) * SJ_STRIDE + rl_] = sS[10][c];
      }

      if (doJp && wr && valid && !isLin) {
        const float JabJIdx_00 = sS[4][c], JabJIdx_01 = sS[5][c], JabJIdx_10 = sS[6][c], JabJIdx_11 = sS[7][c];
        const float JI_r0 = sS[12][c], JI_r1 = sS[13][c];
        // JpJdF of EFResidual::takeDataF (OB/EnergyFunctionalStructs.cpp:39-44)
        const float v0 = JIdxJIdx_00 * d_d_x + JIdxJIdx_10 * d_d_y;
        const float v1 = JIdxJIdx_10 * d_d_x + JIdxJIdx_11 * d_d_y;
        float4 o0, o1;
        o0.x = dxi_x[0] * v0 + dxi_y[0] * v1;
        o0.y = dxi_x[1] * v0 + dxi_y[1] * v1;
        o0.z = dxi_x[2] * v0 + dxi_y[2] * v1;
        o0.w = dxi_x[3] * v0 + dxi_y[3] * v1;
        o1.x = dxi_x[4] * v0 + dxi_y[4] * v1;
        o1.y = dxi_x[5] * v0 + dxi_y[5] * v1;
        o1.z = JabJIdx_00 * d_d_x + JabJIdx_01 * d_d_y;
        o1.w = JabJIdx_10 * d_d_x + JabJIdx_11 * d_d_y;
        // per-residual terms of Hdd_acc / bd_acc / Hcd_acc (OB/AccumulatedTopHessian.cpp:124-127), zero while inactive;
        // without doApply they are provisional like JpJd: k_apply_res clears them if the residual does not end up IN
        const bool termsLive = doApply ? activeAfter : (st != SOS_RES_OOB);
        float4 p0, p1;
        p0.x = v0 * d_d_x + v1 * d_d_y;
        p0.y = JI_r0 * d_d_x + JI_r1 * d_d_y;
        p0.z = dCx0 * v0 + dCy0 * v1;
        p0.w = dCx1 * v0 + dCy1 * v1;
        p1.x = dCx2 * v0 + dCy2 * v1;
        p1.y = dCx3 * v0 + dCy3 * v1;
        p1.z = 1.f;  // counts towards ngoodres
        p1.w = 0.f;  // *_accAF sums
        if (!termsLive) o0 = o1 = p0 = p1 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 *jp = reinterpret_cast<float4 *>(d.JpJd + 8 * (size_t)s);
        jp[0] = o0;
        jp[1] = o1;
        float4 *pt = reinterpret_cast<float4 *>(d.s_pterm + 8 * (size_t)s);
        pt[0] = p0;
        pt[1] = p1;
      }

      // ---- operand rows of the block sums (AccumulatedTopHessianSSE::addPoint<0>, OB/AccumulatedTopHessian.cpp:101-127
      // with AccumulatorApprox::update / updateTopRight / updateBotRight): per residual two rows
      //   L1 = (x | 0 0 0 | use 0 0)      R1 = (a x + b y | JabJIdx_00 JabJIdx_10 JI_r0 | Jab2_00 Jab2_01 Jab_r0)
      //   L2 = (y | 0 0 0 | 0 use 0)      R2 = (b x + c y | JabJIdx_01 JabJIdx_11 JI_r1 | Jab2_11 Jab_r1 rr)
      // so that sum_r L^T R holds the 10x10 block (rows/cols 0..9), the 10x3 top-right (cols 10..12) and, in rows 13
      // and 14, the six bottom-right sums (cols 13..15).  Rows of residuals that do not contribute are zero.
      if (doL || doR) {
        float *Lb = sBig + tl2 * L2_LR_TILE, *Rb = Lb + 16 * L2_LR_PST;
        const float xs[10] = {dCx0, dCx1, dCx2, dCx3, dxi_x[0], dxi_x[1], dxi_x[2], dxi_x[3], dxi_x[4], dxi_x[5]};
        const float ys[10] = {dCy0, dCy1, dCy2, dCy3, dxi_y[0], dxi_y[1], dxi_y[2], dxi_y[3], dxi_y[4], dxi_y[5]};
        if (doL) {
#pragma unroll
          for (int j = 0; j < 10; j++) {
            Lb[j * L2_LR_PST + rl_] = use ? xs[j] : 0.f;
            Lb[j * L2_LR_PST + 32 + rl_] = use ? ys[j] : 0.f;
          }
          const float one = use ? 1.f : 0.f;
          Lb[13 * L2_LR_PST + rl_] = one;
          Lb[13 * L2_LR_PST + 32 + rl_] = 0.f;
          Lb[14 * L2_LR_PST + rl_] = 0.f;
          Lb[14 * L2_LR_PST + 32 + rl_] = one;
          const unsigned long long ub = __ballot(use);
          if (rl_ == 0) {  // residual count of the tile and the padding of the 96-float record
            float *o = fuse_top + (size_t)tile2 * SOS_TOPN;
            o[91] = (float)__popc((unsigned)(ub >> (32 * tl2)));
            o[92] = 0.f; o[93] = 0.f; o[94] = 0.f; o[95] = 0.f;
          }
        } else {
          const float wa = JIdxJIdx_00, wb = JIdxJIdx_10, wc = JIdxJIdx_11;
#pragma unroll
          for (int j = 0; j < 10; j++) {
            Rb[j * L2_LR_PST + rl_] = use ? __fmaf_rn(wa, xs[j], wb * ys[j]) : 0.f;
            Rb[j * L2_LR_PST + 32 + rl_] = use ? __fmaf_rn(wb, xs[j], wc * ys[j]) : 0.f;
          }
          const float r1t[6] = {sS[4][c], sS[6][c], sS[12][c], sS[8][c], sS[9][c], sS[14][c]};
          const float r2t[6] = {sS[5][c], sS[7][c], sS[13][c], sS[10][c], sS[15][c], sS[16][c]};
#pragma unroll
          for (int j = 0; j < 6; j++) {
            Rb[(10 + j) * L2_LR_PST + rl_] = use ? r1t[j] : 0.f;
            Rb[(10 + j) * L2_LR_PST + 32 + rl_] = use ? r2t[j] : 0.f;
          }
        }
      }
    }
  }
  // everything the host reads (tile energies, newest-frame energies) was stored by the phase-2 wave
  if (fuse_top) {
    // ---- the 13x13 block sums of each tile on the matrix cores: D (16x16) = sum over the tile's 64 rows L^T R,
    // v_mfma_f32_16x16x4_f32 x 16, one wave per tile
    __syncthreads();
    LIN_STAMP(6);
    const int tlm = role - 4, tilem = tbase + tlm;
    if (role >= 4 && role < 4 + L2_TILES && tilem < tlim) {
      const float *Lb = sBig + tlm * L2_LR_TILE, *Rb = Lb + 16 * L2_LR_PST;
      const int m = lane & 15, kq = lane >> 4;
      const bool aLive = m < 10 || m == 13 || m == 14;  // the other rows of L are zero by construction (never stored)
      f32x4 accs[4];
      float4 av[4], bv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {  // lane group kq takes rows 16 kq .. 16 kq + 15: MFMA t uses rows {t, 16+t, 32+t, 48+t}
        av[j] = *reinterpret_cast<const float4 *>(Lb + m * L2_LR_PST + 16 * kq + 4 * j);
        bv[j] = *reinterpret_cast<const float4 *>(Rb + m * L2_LR_PST + 16 * kq + 4 * j);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {  // four independent chains of four, then a pairwise sum: shorter dependency chains
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};  // on the matrix pipe and a flatter summation tree than one chain of sixteen
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(aLive ? av[j].x : 0.f, bv[j].x, a4, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(aLive ? av[j].y : 0.f, bv[j].y, a4, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(aLive ? av[j].z : 0.f, bv[j].z, a4, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(aLive ? av[j].w : 0.f, bv[j].w, a4, 0, 0, 0);
        accs[j] = a4;
      }
      const f32x4 acc = (accs[0] + accs[1]) + (accs[2] + accs[3]);
      // scatter into the packed record: 55 uniques of the 10x10 block (row-major upper triangle), 10x3 top-right,
      // the 6 bottom-right sums (layout of top_value<K> above)
      float *out = fuse_top + (size_t)tilem * SOS_TOPN;
      const int nn = m;
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int mm = kq * 4 + v;
        int idx = -1;
        if (nn < 10) {
          if (mm <= nn) idx = mm * 10 - (mm * (mm - 1)) / 2 + (nn - mm);
        } else if (nn < 13) {
          if (mm < 10) idx = 55 + 3 * mm + (nn - 10);
        } else {
          if (mm == 13) idx = 85 + (nn - 13);
          else if (mm == 14) idx = 88 + (nn - 13);
        }
        if (idx >= 0) out[idx] = acc[v];
      }
    }
    LIN_STAMP(7);
    return;  // the tiles stay on chip
  }
  __syncthreads();

  // =============================== phase 3: coalesced store of the staged tiles ===============================
  if (tile_ok) {
    float *Jt = d.J + (size_t)tile * SOS_TILE_FLOATS;
    const unsigned linmask = sLin2[tloc];
    for (int q = t256; q < SOS_JPLANES * 8; q += 256) {
      const int plane = q >> 3, chunk = q & 7;
      const float4 v = *reinterpret_cast<const float4 *>(&sJ[plane * SJ_STRIDE + 4 * chunk]);
      float *dst = Jt + plane * SOS_TILE + 4 * chunk;
      const unsigned lm = (linmask >> (4 * chunk)) & 0xfu;
      if (lm == 0) {
        *reinterpret_cast<float4 *>(dst) = v;
      } else {  // keep the frozen Jacobian of linearized residuals sharing this tile (rare)
        if (!(lm & 1u)) dst[0] = v.x;
        if (!(lm & 2u)) dst[1] = v.y;
        if (!(lm & 4u)) dst[2] = v.z;
        if (!(lm & 8u)) dst[3] = v.w;
      }
    }
  }
}

// launch-shape twin of k_linearize2 that does nothing: its duration is the floor any kernel of this grid / LDS size pays
__global__ __launch_bounds__(256 * L2_TILES, 6) void k_lin_floor(float *sink) {
  __shared__ float sBig[32 * L2_TILES * 17 * 8 + L2_NS * 32 * L2_TILES];
  if (sink) {
    sBig[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    sink[blockIdx.x] = sBig[(threadIdx.x + 1) & 511];
  }
}

// sum of the returned energies in double, fixed order: deterministic
__global__ __launch_bounds__(1024) void k_sum_ret(const float *__restrict__ ret, int n, double *out) {
  __shared__ double sm[1024];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 1024) a += (double)ret[i];
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

// ================================================================================================
// applyRes(true) / resetOOB
// ================================================================================================
__global__ void k_apply_res(BaDev d) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.ntilesA * SOS_TILE) return;
  if (d.s_point[s] < 0) return;
  unsigned f = d.s_flags[s];
  if (f & DF_LINEARIZED) return;
  if (d.s_state[s] == SOS_RES_OOB) return;  // can never go back from OOB (k_linearize left JpJd / pterm zero)
  const int ns = d.s_newstate[s];
  f = (ns == SOS_RES_IN) ? (f | DF_ACTIVE) : (f & ~DF_ACTIVE);
  d.s_flags[s] = (uint8_t)f;
  d.s_state[s] = (uint8_t)ns;
  d.s_energy[s] = d.s_newenergy[s];
  if (ns != SOS_RES_IN) {
    float4 *jp = reinterpret_cast<float4 *>(d.JpJd + 8 * (size_t)s);
    jp[0] = jp[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 *pt = reinterpret_cast<float4 *>(d.s_pterm + 8 * (size_t)s);
    pt[0] = pt[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// PointFrameResidual::applyRes(true), Jacobian half (FS/Residuals.cpp:311-319: std::swap(J, efResidual->J) + takeDataF for
// the residuals whose new state is IN): one block per tile copies the columns of the committed residuals from the
// scratch tile (PointFrameResidual::J) into the applied tile (EFResidual::J), with their JpJdF rows and point terms.
// Runs BEFORE k_apply_res (it reads the states that kernel overwrites).
__global__ __launch_bounds__(256) void k_commit_new(BaDev d, const float *__restrict__ Jnew, const float *__restrict__ JpJdNew,
                                                     const float *__restrict__ ptermNew) {
  const int tile = blockIdx.x, r = threadIdx.x & 31;
  const int s = tile * SOS_TILE + r;
  const bool commit = d.s_point[s] >= 0 && !(d.s_flags[s] & DF_LINEARIZED) && d.s_state[s] != SOS_RES_OOB &&
                      d.s_newstate[s] == SOS_RES_IN;
  if (!commit) return;
  const size_t base = (size_t)tile * SOS_TILE_FLOATS + r;
  for (int pl = threadIdx.x >> 5; pl < SOS_JPLANES; pl += 8) d.J[base + (size_t)pl * SOS_TILE] = Jnew[base + (size_t)pl * SOS_TILE];
  if (threadIdx.x < 32) {
    const float4 *a = reinterpret_cast<const float4 *>(JpJdNew + 8 * (size_t)s), *b = reinterpret_cast<const float4 *>(ptermNew + 8 * (size_t)s);
    float4 *ja = reinterpret_cast<float4 *>(d.JpJd + 8 * (size_t)s), *pb = reinterpret_cast<float4 *>(d.s_pterm + 8 * (size_t)s);
    ja[0] = a[0]; ja[1] = a[1];
    pb[0] = b[0]; pb[1] = b[1];
  }
}

__global__ void k_reset_oob(BaDev d) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.ntilesA * SOS_TILE) return;
  if (d.s_point[s] < 0 || (d.s_flags[s] & DF_LINEARIZED) || !(d.s_flags[s] & DF_VALID)) return;  // (dead: sos_ba_kill_residuals)
  d.s_newenergy[s] = 0;
  d.s_energy[s] = 0;
  d.s_newstate[s] = SOS_RES_OUTLIER;
  d.s_state[s] = SOS_RES_IN;
}

// ================================================================================================
// k_top_accumulate: lane = residual, half-wave (32 lanes) = tile.  91 uniques of the 13x13 block are
// formed per residual in registers and reduced over the tile with a transposed butterfly (93 shuffles
// instead of 5*91); lane l of the half ends up owning 3 of the 96 (padded) sums.
// ================================================================================================
template <int HALF>
__device__ __forceinline__ void bfly_step(float *v, int m, bool hi) {
#pragma unroll
  for (int i = 0; i < HALF; i++) {
    const float a = v[i], b = v[i + HALF];
    const float send = hi ? a : b;
    const float keep = hi ? b : a;
    v[i] = keep + __shfl_xor(send, m, 64);
  }
}

// gather variant: `list` holds, per virtual tile, 32 sorted residual indices (-1 = empty)
template <bool GATHER>
__device__ __forceinline__ void top_accumulate_body(const BaDev &d, int blk, int tile0, int ntile, int mode,
                                                    const int *__restrict__ list, const int *__restrict__ list_pair,
                                                    float *__restrict__ top_part) {
  const int lane = threadIdx.x & 63;
  const int vt = (blk * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);  // virtual tile
  const int r = lane & 31;
  const bool tile_ok = vt < ntile;
  int s, pair;
  if (GATHER) {
    s = tile_ok ? list[vt * SOS_TILE + r] : -1;
    pair = tile_ok ? list_pair[vt] : 0;
  } else {
    s = tile_ok ? (tile0 + vt) * SOS_TILE + r : -1;
    pair = tile_ok ? d.t_pair[tile0 + vt] : 0;
  }
  const bool have = s >= 0 && d.s_point[s >= 0 ? s : 0] >= 0;
  const int ss = have ? s : 0;
  const unsigned f = d.s_flags[ss];
  bool use = have && (f & DF_ACTIVE);
  if (mode == 0) use = use && !(f & DF_LINEARIZED);
  if (mode == 1) use = use && (f & DF_LINEARIZED);

  const float *Jt = d.J + (size_t)(ss >> 5) * SOS_TILE_FLOATS + (ss & 31);
#define JL(pl) Jt[(pl)*SOS_TILE]
  float x[10], y[10];
#pragma unroll
  for (int i = 0; i < 4; i++) { x[i] = JL(JP_DC0 + i); y[i] = JL(JP_DC1 + i); }
#pragma unroll
  for (int i = 0; i < 6; i++) { x[4 + i] = JL(JP_DXI0 + i); y[4 + i] = JL(JP_DXI1 + i); }
  const float Jpdd0 = JL(JP_DD), Jpdd1 = JL(JP_DD + 1);
  const float a = JL(JP_JIDX2), b = JL(JP_JIDX2 + 1), c = JL(JP_JIDX2 + 2);
  const float jab00 = JL(JP_JABJIDX), jab01 = JL(JP_JABJIDX + 1), jab10 = JL(JP_JABJIDX + 2), jab11 = JL(JP_JABJIDX + 3);
  const float ab00 = JL(JP_JAB2), ab01 = JL(JP_JAB2 + 1), ab11 = JL(JP_JAB2 + 2);

  // resApprox (OB/AccumulatedTopHessian.cpp:68-98)
  float Jp_delta_x = 0, Jp_delta_y = 0, dp6 = 0, dp7 = 0;
  if (mode == 1) {
    const float *dp = d.adHTdelta + 8 * pair;
    float dx = 0, dy = 0, dcx = 0, dcy = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) { dx += x[4 + i] * dp[i]; dy += y[4 + i] * dp[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { dcx += x[i] * d.cdelta[i]; dcy += y[i] * d.cdelta[i]; }
    const float dd = d.pts[have ? d.s_point[ss] : 0].deltaF;
    Jp_delta_x = dx + dcx + Jpdd0 * dd;
    Jp_delta_y = dy + dcy + Jpdd1 * dd;
    dp6 = dp[6];
    dp7 = dp[7];
  }
  float JI_r0 = 0, JI_r1 = 0, Jab_r0 = 0, Jab_r1 = 0, rr = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const float ji0 = JL(JP_JIDX0 + i), ji1 = JL(JP_JIDX1 + i), ja0 = JL(JP_JAB0 + i), ja1 = JL(JP_JAB1 + i);
    float ra;
    if (mode == 0) ra = JL(JP_RESF + i);
    else {
      float rtz = d.s_rtz[8 * (size_t)ss + i];
      if (mode == 1) {
        rtz = rtz + ji0 * Jp_delta_x;
        rtz = rtz + ji1 * Jp_delta_y;
        rtz = rtz + ja0 * dp6;
        rtz = rtz + ja1 * dp7;
      }
      ra = rtz;
    }
    JI_r0 += ra * ji0;
    JI_r1 += ra * ji1;
    Jab_r0 += ra * ja0;
    Jab_r1 += ra * ja1;
    rr += ra * ra;
  }
#undef JL
  // per-residual terms of the point sums (OB/AccumulatedTopHessian.cpp:124-127)
  {
    const float Ji2_Jpdd0 = a * Jpdd0 + b * Jpdd1;
    const float Ji2_Jpdd1 = b * Jpdd0 + c * Jpdd1;
    if (have && mode != 0) {  // mode 0: written by k_linearize / k_apply_res
      float4 p0, p1;
      p0.x = use ? Ji2_Jpdd0 * Jpdd0 + Ji2_Jpdd1 * Jpdd1 : 0.f;
      p0.y = use ? JI_r0 * Jpdd0 + JI_r1 * Jpdd1 : 0.f;
      p0.z = use ? x[0] * Ji2_Jpdd0 + y[0] * Ji2_Jpdd1 : 0.f;
      p0.w = use ? x[1] * Ji2_Jpdd0 + y[1] * Ji2_Jpdd1 : 0.f;
      p1.x = use ? x[2] * Ji2_Jpdd0 + y[2] * Ji2_Jpdd1 : 0.f;
      p1.y = use ? x[3] * Ji2_Jpdd0 + y[3] * Ji2_Jpdd1 : 0.f;
      p1.z = use ? 1.f : 0.f;                                   // counts towards ngoodres
      p1.w = (mode != 0) ? 1.f : 0.f;                           // goes to the *_accLF sums
      float4 *pt = reinterpret_cast<float4 *>(d.s_pterm + 8 * (size_t)ss);
      pt[0] = p0;
      pt[1] = p1;
    }
  }

  float v[SOS_TOPN];
  {
    int k = 0;
#pragma unroll
    for (int rr_ = 0; rr_ < 10; rr_++)
#pragma unroll
      for (int cc = rr_; cc < 10; cc++) {  // AccumulatorApprox::update, OB/MatrixAccumulators.h:928-1055
        const float t = a * x[cc] * x[rr_] + c * y[cc] * y[rr_] + b * (x[cc] * y[rr_] + y[cc] * x[rr_]);
        v[k++] = use ? t : 0.f;
      }
#pragma unroll
    for (int i = 0; i < 10; i++) {  // updateTopRight :1057-1101
      v[k++] = use ? x[i] * jab00 + y[i] * jab01 : 0.f;
      v[k++] = use ? x[i] * jab10 + y[i] * jab11 : 0.f;
      v[k++] = use ? x[i] * JI_r0 + y[i] * JI_r1 : 0.f;
    }
    v[85] = use ? ab00 : 0.f;  // updateBotRight :1103-1112
    v[86] = use ? ab01 : 0.f;
    v[87] = use ? Jab_r0 : 0.f;
    v[88] = use ? ab11 : 0.f;
    v[89] = use ? Jab_r1 : 0.f;
    v[90] = use ? rr : 0.f;
    v[91] = use ? 1.f : 0.f;  // nres
    v[92] = v[93] = v[94] = v[95] = 0.f;
  }
  bfly_step<48>(v, 16, (r & 16) != 0);
  bfly_step<24>(v, 8, (r & 8) != 0);
  bfly_step<12>(v, 4, (r & 4) != 0);
  bfly_step<6>(v, 2, (r & 2) != 0);
  bfly_step<3>(v, 1, (r & 1) != 0);
  if (tile_ok) {
    const int base = ((r >> 4) & 1) * 48 + ((r >> 3) & 1) * 24 + ((r >> 2) & 1) * 12 + ((r >> 1) & 1) * 6 + (r & 1) * 3;
    float *o = top_part + (size_t)vt * SOS_TOPN + base;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
  }
}
template <bool GATHER>
__global__ __launch_bounds__(256) void k_top_accumulate(BaDev d, int tile0, int ntile, int mode,
                                                        const int *__restrict__ list,
                                                        const int *__restrict__ list_pair,
                                                        float *__restrict__ top_part, int *__restrict__ top_cnt) {
  (void)top_cnt;
  top_accumulate_body<GATHER>(d, blockIdx.x, tile0, ntile, mode, list, list_pair, top_part);
}

// ================================================================================================
// k_reduce_all: ONE launch that (a) sums the tile partials of every (mode, pair) in fp64 (as the
// reference sums its per-thread accumulators, OB/AccumulatedTopHessian.cpp:252-259) into the packed
// fp32 accumulator, (b) sums the Gram partials per host and scatters them into accD / accE / accEB,
// (c) Hcc / bc, (d) the residual counts.  Block roles are selected by blockIdx ranges.
// ================================================================================================
struct ReduceArgs {
  const float *top_part;
  const int *pair_tile_begin;  // [2*n*n + 1]
  const float *gram_part;
  const int *host_chunk_begin;  // [n + 1]
  int n, Dm, nchunks, nmodes;
  int b_top, b_sc, b_tail;  // number of blocks per role
  float *accTop;            // nmodes * n*n * 91
  float *accD, *accE, *accEB, *accHcc, *accbc, *nres;
};

__global__ __launch_bounds__(128) void k_reduce_all(ReduceArgs a) {
  int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int n = a.n;
  if (b < a.b_top) {  // ---- (a) one block per (mode, pair)
    if (tid >= SOS_TOPN) return;
    const int t0 = a.pair_tile_begin[b], t1 = a.pair_tile_begin[b + 1];
    double s = 0;
#pragma unroll 8
    for (int t = t0; t < t1; t++) s += (double)a.top_part[(size_t)t * SOS_TOPN + tid];
    if (tid < 91) a.accTop[(size_t)b * 91 + tid] = (float)s;
    return;
  }
  b -= a.b_top;
  if (b < a.b_sc) {  // ---- (b) Gram sub-blocks of host h
    const int cols = 8 * n + 5, per_host = 8 * n * cols;
    const int e = b * 128 + tid;
    if (e >= n * per_host) return;
    const int h = e / per_host, q = e - h * per_host;
    const int r = q / cols, c = q - r * cols;
    if ((r >> 4) > (c >> 4)) return;  // only the 16x16 tiles on and above the diagonal exist; their owners write the mirror image too
    double s = 0;
    const int kc0 = a.host_chunk_begin[h], kc1 = a.host_chunk_begin[h + 1];
    const size_t off = (size_t)r * a.Dm + c;
#pragma unroll 8
    for (int k = kc0; k < kc1; k++) s += (double)a.gram_part[(size_t)k * a.Dm * a.Dm + off];
    const int t1 = r >> 3, i = r & 7;
    if (c < 8 * n) {
      const int t2 = c >> 3, j = c & 7;
      a.accD[(size_t)(h + n * t1 + n * n * t2) * 64 + i * 8 + j] = (float)s;
      if ((r >> 4) < (c >> 4)) a.accD[(size_t)(h + n * t2 + n * n * t1) * 64 + j * 8 + i] = (float)s;  // G symmetric
    } else if (c < 8 * n + 4) {
      a.accE[(size_t)(h + n * t1) * 32 + i * 4 + (c - 8 * n)] = (float)s;
    } else {
      a.accEB[(size_t)(h + n * t1) * 8 + i] = (float)s;
    }
    return;
  }
  // ---- (c)+(d) tail blocks: one block per scalar (16 Hcc + 4 bc + nmodes counts), strided loads and a
  // fixed-shape tree in fp64 (a single thread walking 10^2..10^3 dependent-latency loads costs 100s of us)
  b -= a.b_sc;
  __shared__ double sm[128];
  double v = 0;
  if (b < 20) {
    const int r = 8 * n + (b < 16 ? (b >> 2) : (b - 16)), c = 8 * n + (b < 16 ? (b & 3) : 4);
    for (int k = tid; k < a.nchunks; k += 128) v += (double)a.gram_part[(size_t)k * a.Dm * a.Dm + (size_t)r * a.Dm + c];
  } else {
    const int m = b - 20;
    const int t0 = a.pair_tile_begin[m * n * n], t1 = a.pair_tile_begin[(m + 1) * n * n];
    for (int t = t0 + tid; t < t1; t += 128) v += (double)a.top_part[(size_t)t * SOS_TOPN + 91];
  }
  sm[tid] = v;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (tid < o) sm[tid] += sm[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    if (b < 16) a.accHcc[b] = (float)sm[0];
    else if (b < 20) a.accbc[b - 16] = (float)sm[0];
    else a.nres[b - 20] = (float)sm[0];
  }
}

// ================================================================================================
// k_point_prep: one thread per point, residuals visited in EFPoint::residualsAll order.  The loop is
// unrolled by 4 with all loads of a group issued before the (order-preserving) accumulation.
// ================================================================================================
struct PrepOut { float hdi, hcd[4], bdsum; };
__device__ __forceinline__ PrepOut point_prep_body(const BaDev &d, int shiftPriorToZero, int p) {
  float HddA = 0, bdA = 0, HcdA[4] = {0, 0, 0, 0}, HddL = 0, bdL = 0, HcdL[4] = {0, 0, 0, 0};
  float ngood = 0;
  const int q0 = d.p_begin[p], q1 = d.p_begin[p + 1];
  // residual list, point record and the per-residual terms are requested level by level for ALL residuals of the
  // point at once (groups of PU): the sums below stay in residualsAll order, only the memory latencies overlap
  constexpr int PU = 8;
  const sos_point *ptp = d.pts + p;
  const float priorF = ptp->priorF, deltaF = ptp->deltaF;
  const float stepOld = d.p_out[16 * (size_t)p + PO_STEP];
  for (int q = q0; q < q1; q += PU) {
    int sidx[PU];
    float4 a[PU], b[PU];
#pragma unroll
    for (int k = 0; k < PU; k++) sidx[k] = d.p_list2[min(q + k, q1 - 1)].x;
#pragma unroll
    for (int k = 0; k < PU; k++) {
      const float4 *pt = reinterpret_cast<const float4 *>(d.s_pterm + 8 * (size_t)sidx[k]);
      a[k] = pt[0];
      b[k] = pt[1];
    }
#pragma unroll
    for (int k = 0; k < PU; k++) {
      if (q + k >= q1) break;
      if (b[k].z == 0.f) continue;  // not active
      ngood += 1.f;
      if (b[k].w != 0.f) {
        HddL += a[k].x; bdL += a[k].y;
        HcdL[0] += a[k].z; HcdL[1] += a[k].w; HcdL[2] += b[k].x; HcdL[3] += b[k].y;
      } else {
        HddA += a[k].x; bdA += a[k].y;
        HcdA[0] += a[k].z; HcdA[1] += a[k].w; HcdA[2] += b[k].x; HcdA[3] += b[k].y;
      }
    }
  }
  float *o = d.p_out + 16 * (size_t)p;
  float4 o0, o1, o2, o3;
  o0.x = HddA; o0.y = bdA; o0.z = HcdA[0]; o0.w = HcdA[1];
  o1.x = HcdA[2]; o1.y = HcdA[3]; o1.z = HddL; o1.w = bdL;
  o2.x = HcdL[0]; o2.y = HcdL[1]; o2.z = HcdL[2]; o2.w = HcdL[3];
  o3.w = stepOld;
  if (ngood == 0.f) {  // OB/AccumulatedSCHessian.cpp:34-44
    o3.x = 0; o3.y = 0; o3.z = 0;
  } else {
    float H = HddA + HddL + priorF;
    if (H < 1e-10) H = 1e-10;
    o3.z = H;
    o3.x = (float)(1.0 / (double)H);
    float bdSum = bdA + bdL;
    if (shiftPriorToZero) bdSum += priorF * deltaF;
    o3.y = bdSum;
  }
  float4 *ov = reinterpret_cast<float4 *>(o);
  ov[0] = o0; ov[1] = o1; ov[2] = o2; ov[3] = o3;
  PrepOut r;
  r.hdi = o3.x;
  r.hcd[0] = o0.z + o2.x; r.hcd[1] = o0.w + o2.y; r.hcd[2] = o1.x + o2.z; r.hcd[3] = o1.y + o2.w;  // Hcd_accAF + Hcd_accLF
  r.bdsum = o3.y;
  return r;
}
__global__ void k_point_prep(BaDev d, int shiftPriorToZero, const int *__restrict__ plist, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  point_prep_body(d, shiftPriorToZero, plist ? plist[i] : i);
}

// ================================================================================================
// k_sc_gram: G = sum_p Hdi_p e_p e_p^T with e_p = [JpJdF(p,t=0..n-1) (8n) | Hcd (4) | bdSum | 0...]
// over a chunk of SOS_GC points of one host.  The D / E / EB / Hcc / bc accumulators of
// AccumulatedSCHessianSSE::addPoint (OB/AccumulatedSCHessian.cpp:57-78) are sub-blocks of G.
// v_mfma_f32_16x16x4_f32: K = points.  4 waves share the Dm/16 x Dm/16 output tiles.
// JpJd rows of inactive residuals are zero (see k_linearize / k_apply_res), so no flag lookups.
// ================================================================================================
#ifndef SOS_GC
#define SOS_GC 32
#endif

// PREP: -1 = the per-point sums were produced by k_point_prep (p_out); 0 / 1 = produce them here for the chunk's
// own points with shiftPriorToZero = PREP (the points of a window are partitioned over the chunks)
template <int PREP>
__device__ __forceinline__ void sc_gram_body(const BaDev &d, int blk, const int *__restrict__ chunk_pt, int Dm, int ld,
                                             float *__restrict__ gram_part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *A = smem;                   // [SOS_GC][ld]
  float *sHdi = smem + SOS_GC * ld;  // [SOS_GC]
  const int tid = threadIdx.x;
  const int n = d.n;
  for (int q = tid; q < SOS_GC * ld / 4; q += 256) reinterpret_cast<float4 *>(A)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  // stage: thread per (point, target); t == n is the point's own column block (Hcd, bdSum) and Hdi
  for (int q = tid; q < SOS_GC * (n + 1); q += 256) {
    const int t = q / SOS_GC, pl = q - t * SOS_GC;  // the 32 prep items (t == n) end up in one half-wave
    const int p = chunk_pt[blk * SOS_GC + pl];
    if (p < 0) {
      if (t == n) sHdi[pl] = 0.f;
      continue;
    }
    if (t == n) {
      float *row = A + pl * ld + 8 * n;
      if (PREP >= 0) {
        const PrepOut r = point_prep_body(d, PREP, p);
        sHdi[pl] = r.hdi;
        row[0] = r.hcd[0]; row[1] = r.hcd[1]; row[2] = r.hcd[2]; row[3] = r.hcd[3];
        row[4] = r.bdsum;
      } else {
        const float4 *po = reinterpret_cast<const float4 *>(d.p_out + 16 * (size_t)p);
        const float4 v0 = po[0], v1 = po[1], v2 = po[2], v3 = po[3];
        sHdi[pl] = v3.x;
        row[0] = v0.z + v2.x;  // Hcd_accAF + Hcd_accLF
        row[1] = v0.w + v2.y;
        row[2] = v1.x + v2.z;
        row[3] = v1.y + v2.w;
        row[4] = v3.y;         // bdSumF
      }
    } else {
      const int s = d.p_res_t[(size_t)p * n + t];
      if (s >= 0) {
        const float4 *jp = reinterpret_cast<const float4 *>(d.JpJd + 8 * (size_t)s);
        float4 *row = reinterpret_cast<float4 *>(A + pl * ld + 8 * t);
        row[0] = jp[0];
        row[1] = jp[1];
      }
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int T = Dm >> 4;
  const int kq = lane >> 4, col = lane & 15;
  float *g = gram_part + (size_t)blk * Dm * Dm;
  // G is symmetric: only the 16x16 tiles on and above the diagonal are formed and stored (T (T + 1) / 2 of T^2); the
  // consumers read the mirror image for the rest (k_reduce_all)
  for (int ut = wave; ut < T * (T + 1) / 2; ut += 4) {
    int mt = 0, rem = ut;
    while (rem >= T - mt) { rem -= T - mt; mt++; }
    const int m0 = mt << 4, n0 = (mt + rem) << 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < SOS_GC / 4; kk++) {
      const int k = kk * 4 + kq;
      const float av = sHdi[k] * A[k * ld + m0 + col];
      const float bv = A[k * ld + n0 + col];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rgi = 0; rgi < 4; rgi++) g[(size_t)(m0 + kq * 4 + rgi) * Dm + n0 + col] = acc[rgi];
  }
}
__global__ __launch_bounds__(256) void k_sc_gram(BaDev d, const int *__restrict__ chunk_pt /* nchunks*SOS_GC point ids */,
                                                 int Dm, int ld, float *__restrict__ gram_part) {
  sc_gram_body<-1>(d, blockIdx.x, chunk_pt, Dm, ld, gram_part);
}
// `flag` != nullptr: this kernel is queued right behind a producer whose results the host is polling for; a kernel
// only starts after its predecessor's end-of-kernel release, so the first thread can publish the predecessor's
// sequence number at once -- the separate k_publish launch (and its ~5 us in the chain) disappears.
__global__ __launch_bounds__(256) void k_sc_gram_prep(BaDev d, const int *__restrict__ chunk_pt, int Dm, int ld,
                                                      float *__restrict__ gram_part, int *flag, int seq) {
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  sc_gram_body<1>(d, blockIdx.x, chunk_pt, Dm, ld, gram_part);
}
// One launch for the two independent halves of the accumulation of a window without linearised residuals:
// blocks [0, nTopBlocks) = AccumulatedTopHessian over the A tiles, the rest = per-point sums + Schur Gram chunks
// (their inputs JpJd / pterm come from k_linearize, not from the top pass).
__global__ __launch_bounds__(256) void k_accumulate_fused(BaDev d, int nTopBlocks, float *__restrict__ top_part,
                                                          const int *__restrict__ chunk_pt, int Dm, int ld,
                                                          float *__restrict__ gram_part) {
  if ((int)blockIdx.x < nTopBlocks) top_accumulate_body<false>(d, blockIdx.x, 0, d.ntilesA, 0, nullptr, nullptr, top_part);
  else sc_gram_body<1>(d, blockIdx.x - nTopBlocks, chunk_pt, Dm, ld, gram_part);
}

// ================================================================================================
// fp64 stitch from the packed fp32 accumulators, two stages each (products per pair / triple in
// parallel, then fixed-order sums per output block: deterministic, no atomics)
// ================================================================================================
__device__ __forceinline__ int top_idx(int i, int j) {  // 13x13 symmetric -> index into the 91 uniques
  if (i > j) { const int t = i; i = j; j = t; }
  if (j < 10) return i * 10 - (i * (i - 1)) / 2 + (j - i);
  if (i < 10) return 55 + 3 * i + (j - 10);
  return 85 + (i == 10 ? (j - 10) : (i == 11 ? 3 + (j - 11) : 5));
}

#define SOS_TOPC 272  // doubles per pair: P1 P2 P3 (64 each), Hc1 Hc2 (32 each), b1 b2 (8 each)

// stage 1, grid (n*n, nmodes): pair (h,t) -> AH B AH^T, AT B AT^T, AH B AT^T, AH Bpc, AT Bpc, AH bp, AT bp
// (OB/AccumulatedTopHessian.cpp:261-288)
__device__ __forceinline__ void stitch_top_pairs_body(int bx, int by, int n, const float *__restrict__ acc_top, const double *__restrict__ adHost,
                                                         const double *__restrict__ adTarget, double *__restrict__ C,
                                                         double *__restrict__ Ccc) {
  __shared__ double sB[64], sAH[64], sAT[64], sT1[64], sT2[64], sBpc[32], sbp[8];
  const int tid = threadIdx.x, i = tid >> 3, j = tid & 7;
  const int pidx = bx;
  if (pidx >= n * n) {  // 20 extra blocks: H_cc (16) and b_c (4) = sums over all pairs, strided loads + fp64 tree
    const int v = pidx - n * n;
    const int r = v < 16 ? (v >> 2) : (v - 16), c = v < 16 ? (v & 3) : 12;
    const float *acc = acc_top + (size_t)by * n * n * 91;
    double sv = 0;
    for (int k = tid; k < n * n; k += 64) sv += (double)acc[(size_t)k * 91 + top_idx(r, c)];
    sT1[tid] = sv;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
      if (tid < o) sT1[tid] += sT1[tid + o];
      __syncthreads();
    }
    if (tid == 0) Ccc[(size_t)by * 20 + v] = sT1[0];
    return;
  }
  const float *blk = acc_top + ((size_t)by * n * n + pidx) * 91;
  double *out = C + ((size_t)by * n * n + pidx) * SOS_TOPC;
  sB[tid] = (double)blk[top_idx(4 + i, 4 + j)];
  sAH[tid] = adHost[(size_t)pidx * 64 + tid];
  sAT[tid] = adTarget[(size_t)pidx * 64 + tid];
  if (tid < 32) sBpc[tid] = (double)blk[top_idx(4 + (tid >> 2), tid & 3)];
  if (tid < 8) sbp[tid] = (double)blk[top_idx(4 + tid, 12)];
  __syncthreads();
  double t1 = 0, t2 = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    t1 += sAH[i * 8 + k] * sB[k * 8 + j];
    t2 += sAT[i * 8 + k] * sB[k * 8 + j];
  }
  sT1[tid] = t1;
  sT2[tid] = t2;
  __syncthreads();
  double p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    p1 += sT1[i * 8 + k] * sAH[j * 8 + k];
    p2 += sT2[i * 8 + k] * sAT[j * 8 + k];
    p3 += sT1[i * 8 + k] * sAT[j * 8 + k];
  }
  out[tid] = p1;
  out[64 + tid] = p2;
  out[128 + tid] = p3;
  if (tid < 32) {
    const int r = tid >> 2, c = tid & 3;
    double h1 = 0, h2 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      h1 += sAH[r * 8 + k] * sBpc[k * 4 + c];
      h2 += sAT[r * 8 + k] * sBpc[k * 4 + c];
    }
    out[192 + tid] = h1;
    out[224 + tid] = h2;
  }
  if (tid < 8) {
    double b1 = 0, b2 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      b1 += sAH[tid * 8 + k] * sbp[k];
      b2 += sAT[tid * 8 + k] * sbp[k];
    }
    out[256 + tid] = b1;
    out[264 + tid] = b2;
  }
}
__global__ __launch_bounds__(64) void k_stitch_top_pairs(int n, const float *__restrict__ acc_top, const double *__restrict__ adHost,
                                                         const double *__restrict__ adTarget, double *__restrict__ C,
                                                         double *__restrict__ Ccc) { stitch_top_pairs_body(blockIdx.x, blockIdx.y, n, acc_top, adHost, adTarget, C, Ccc); }


// stage 2, grid (n*(n+1)/2 + 1, nmodes): fixed-order sums per output block + symmetrisation
// (OB/AccumulatedTopHessian.h:113-126).  Hb = per mode [dim*dim H | dim b].
__device__ __forceinline__ void stitch_top_sum_body(int bx, int by, int n, const double *__restrict__ Ccc, const double *__restrict__ C,
                                                       double *__restrict__ Hb, size_t mode_stride, bool upperOnly = false) {
  const int dim = 4 + 8 * n;
  const int tid = threadIdx.x, i = tid >> 3, j = tid & 7;
  const double *Cm = C + (size_t)by * n * n * SOS_TOPC;
  double *H = Hb + (size_t)by * mode_stride;
  double *bv = H + (size_t)dim * dim;
  const int nblk = n * (n + 1) / 2;
  if (bx == nblk) {
    if (tid < 20) {
      const double sv = Ccc[(size_t)by * 20 + tid];
      if (tid < 16) H[(size_t)(tid >> 2) * dim + (tid & 3)] = sv;
      else bv[tid - 16] = sv;
    }
    return;
  }
  int a = 0, rem = bx;
  while (rem >= n - a) { rem -= n - a; a++; }
  const int bb = a + rem;
  if (a == bb) {
    double sH = 0, sHc = 0, sb = 0;
    // fixed order: pairs (a,t) t = 0..n-1 (host side), then pairs (h,a) h = 0..n-1 (target side; (a,a) is empty)
    const int hc = tid & 31, bc_ = tid & 7;
    // every load of a round of 16 terms is in flight before the first add (the adds keep their order): the sums were a chain
    // of n / 4 dependent L2 round trips
    for (int t0 = 0; t0 < n; t0 += 16) {
      double v0[16], v1[16], v2[16];
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (t0 + u < n) {
          const double *c = Cm + (size_t)(a + n * (t0 + u)) * SOS_TOPC;
          v0[u] = c[tid]; v1[u] = c[192 + hc]; v2[u] = c[256 + bc_];
        }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (t0 + u < n) { sH += v0[u]; sHc += v1[u]; sb += v2[u]; }
    }
    for (int h0 = 0; h0 < n; h0 += 16) {
      double v0[16], v1[16], v2[16];
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (h0 + u < n) {
          const double *c = Cm + (size_t)((h0 + u) + n * a) * SOS_TOPC;
          v0[u] = c[64 + tid]; v1[u] = c[224 + hc]; v2[u] = c[264 + bc_];
        }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (h0 + u < n) { sH += v0[u]; sHc += v1[u]; sb += v2[u]; }
    }
    H[(size_t)(4 + 8 * a + i) * dim + 4 + 8 * a + j] = sH;
    if (tid < 32) {
      const int r = tid >> 2, cc = tid & 3;
      if (!upperOnly) H[(size_t)(4 + 8 * a + r) * dim + cc] = sHc;
      H[(size_t)cc * dim + 4 + 8 * a + r] = sHc;
    }
    if (tid < 8) bv[4 + 8 * a + tid] = sb;
  } else {
    const double *c1 = Cm + (size_t)(a + n * bb) * SOS_TOPC, *c2 = Cm + (size_t)(bb + n * a) * SOS_TOPC;
    const double o = c1[128 + i * 8 + j] + c2[128 + j * 8 + i];
    H[(size_t)(4 + 8 * a + i) * dim + 4 + 8 * bb + j] = o;
    if (!upperOnly) H[(size_t)(4 + 8 * bb + j) * dim + 4 + 8 * a + i] = o;
  }
}
__global__ __launch_bounds__(64) void k_stitch_top_sum(int n, const double *__restrict__ Ccc, const double *__restrict__ C,
                                                       double *__restrict__ Hb, size_t mode_stride) { stitch_top_sum_body(blockIdx.x, blockIdx.y, n, Ccc, C, Hb, mode_stride); }


#define SOS_SCC 128  // doubles per (h,t1,g): C1 = AH[h,t1] M, C2 = AT[h,t1] M
#define SOS_SCE 80   // doubles per (h,t1): AH E (32), AT E (32), AH EB (8), AT EB (8)

// stage 1, grid n^3: (h, t1, g):  M = (g == h) ? sum_t2 D[h,t1,t2] AH[h,t2]^T : D[h,t1,g] AT[h,g]^T, then
// C1 = AH[h,t1] M, C2 = AT[h,t1] M (OB/AccumulatedSCHessian.cpp:117-139 regrouped); blocks with g == 0
// also produce the calib column / b products of (h,t1) (:107-115)
__device__ __forceinline__ void sc_MC_body(int bx, int by, int n, const float *__restrict__ accD, const float *__restrict__ accE,
                                              const float *__restrict__ accEB, const double *__restrict__ adHost,
                                              const double *__restrict__ adTarget, double *__restrict__ C, double *__restrict__ Ce) {
  // every operand of the block is requested before anything is waited for: the g == h blocks sum over all t2, and
  // staging one (D, A) pair per iteration made them a chain of n dependent round trips -- the critical path of the stage
  extern __shared__ __attribute__((aligned(16))) double sDA[];  // [n][64] D blocks, then [n][64] adjoint blocks
  double *sD = sDA, *sA = sDA + (size_t)n * 64;
  __shared__ double sM[64], sAH[64], sAT[64], sE[32], sEB[8];
  const int tid = threadIdx.x, i = tid >> 3, j = tid & 7;
  const int h = bx % n, t1 = (bx / n) % n, g = bx / (n * n);
  const int t2lo = (g == h) ? 0 : g, t2hi = (g == h) ? n : g + 1;
  const int pidx = h + n * t1;
  for (int t2 = t2lo; t2 < t2hi; t2++) {
    sD[(t2 - t2lo) * 64 + tid] = (double)accD[(size_t)(h + n * t1 + n * n * t2) * 64 + tid];
    sA[(t2 - t2lo) * 64 + tid] = ((g == h) ? adHost : adTarget)[(size_t)(h + n * t2) * 64 + tid];
  }
  sAH[tid] = adHost[(size_t)pidx * 64 + tid];
  sAT[tid] = adTarget[(size_t)pidx * 64 + tid];
  if (g == 0) {
    if (tid < 32) sE[tid] = (double)accE[(size_t)pidx * 32 + tid];
    if (tid < 8) sEB[tid] = (double)accEB[(size_t)pidx * 8 + tid];
  }
  __syncthreads();
  double m = 0;
  for (int q = 0; q < t2hi - t2lo; q++) {  // same order of additions as before: t2 ascending, k ascending
#pragma unroll
    for (int k = 0; k < 8; k++) m += sD[q * 64 + i * 8 + k] * sA[q * 64 + j * 8 + k];
  }
  sM[tid] = m;
  __syncthreads();
  double c1 = 0, c2 = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c1 += sAH[i * 8 + k] * sM[k * 8 + j];
    c2 += sAT[i * 8 + k] * sM[k * 8 + j];
  }
  double *out = C + ((size_t)(h * n + t1) * n + g) * SOS_SCC;
  out[tid] = c1;
  out[64 + tid] = c2;
  if (g == 0) {
    double *oe = Ce + (size_t)pidx * SOS_SCE;
    if (tid < 32) {
      const int r = tid >> 2, c = tid & 3;
      double e1 = 0, e2 = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        e1 += sAH[r * 8 + k] * sE[k * 4 + c];
        e2 += sAT[r * 8 + k] * sE[k * 4 + c];
      }
      oe[tid] = e1;
      oe[32 + tid] = e2;
    }
    if (tid < 8) {
      double b1 = 0, b2 = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        b1 += sAH[tid * 8 + k] * sEB[k];
        b2 += sAT[tid * 8 + k] * sEB[k];
      }
      oe[64 + tid] = b1;
      oe[72 + tid] = b2;
    }
  }
}
__global__ __launch_bounds__(64) void k_sc_MC(int n, const float *__restrict__ accD, const float *__restrict__ accE,
                                              const float *__restrict__ accEB, const double *__restrict__ adHost,
                                              const double *__restrict__ adTarget, double *__restrict__ C, double *__restrict__ Ce) { sc_MC_body(blockIdx.x, blockIdx.y, n, accD, accE, accEB, adHost, adTarget, C, Ce); }


// stage 2, grid n*n + 1: H_sc[g1,g2] = sum_t1 C1[g1][t1][g2] + sum_{h != g1} C2[h][g1][g2]
__device__ __forceinline__ void sc_sum_body(int bx, int by, int n, const double *__restrict__ C, const double *__restrict__ Ce,
                                               const float *__restrict__ accHcc, const float *__restrict__ accbc,
                                               double *__restrict__ H, const float *__restrict__ nres,
                                               float *__restrict__ nres_out, bool upperOnly = false) {
  const int dim = 4 + 8 * n;
  double *bv = H + (size_t)dim * dim;
  const int tid = threadIdx.x, i = tid >> 3, j = tid & 7;
  if (bx == n * n) {
    if (tid < 16) H[(size_t)(tid >> 2) * dim + (tid & 3)] = (double)accHcc[tid];
    else if (tid < 20) bv[tid - 16] = (double)accbc[tid - 16];
    else if (tid < 22 && nres_out) nres_out[tid - 20] = nres[tid - 20];  // residual counts travel with H/b
    return;
  }
  const int g1 = bx % n, g2 = bx / n;
  if (upperOnly && g1 > g2) return;  // the solve reads one triangle only
  double s = 0, sc = 0, sb = 0;
  // the (h = g1, t1 = g1) terms are exact zeros (a point has no residual to its own host), so both sums
  // can run branch-free over all n
  for (int t0 = 0; t0 < n; t0 += 16) {  // loads of a round in flight together, adds in the old order (see stitch_top_sum_body)
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (t0 + u < n) v[u] = C[((size_t)(g1 * n + t0 + u) * n + g2) * SOS_SCC + tid];
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (t0 + u < n) s += v[u];
  }
  for (int h0 = 0; h0 < n; h0 += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (h0 + u < n) v[u] = C[((size_t)((h0 + u) * n + g1) * n + g2) * SOS_SCC + 64 + tid];
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (h0 + u < n) s += v[u];
  }
  H[(size_t)(4 + 8 * g1 + i) * dim + 4 + 8 * g2 + j] = s;
  if (g1 == g2) {
    const int hc = tid & 31, bc_ = tid & 7;
    for (int t0 = 0; t0 < n; t0 += 16) {
      double v1[16], v2[16];
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (t0 + u < n) {
          const double *e = Ce + (size_t)(g1 + n * (t0 + u)) * SOS_SCE;
          v1[u] = e[hc]; v2[u] = e[64 + bc_];
        }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (t0 + u < n) { sc += v1[u]; sb += v2[u]; }
    }
    for (int h0 = 0; h0 < n; h0 += 16) {
      double v1[16], v2[16];
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (h0 + u < n) {
          const double *e = Ce + (size_t)((h0 + u) + n * g1) * SOS_SCE;
          v1[u] = e[32 + hc]; v2[u] = e[72 + bc_];
        }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (h0 + u < n) { sc += v1[u]; sb += v2[u]; }
    }
    if (tid < 32) {
      const int r = tid >> 2, c = tid & 3;
      if (!upperOnly) H[(size_t)(4 + 8 * g1 + r) * dim + c] = sc;
      H[(size_t)c * dim + 4 + 8 * g1 + r] = sc;  // transposed calib rows, OB/AccumulatedSCHessian.h:118-123
    }
    if (tid < 8) bv[4 + 8 * g1 + tid] = sb;
  }
}
__global__ __launch_bounds__(64) void k_sc_sum(int n, const double *__restrict__ C, const double *__restrict__ Ce,
                                               const float *__restrict__ accHcc, const float *__restrict__ accbc,
                                               double *__restrict__ H, const float *__restrict__ nres,
                                               float *__restrict__ nres_out) { sc_sum_body(blockIdx.x, blockIdx.y, n, C, Ce, accHcc, accbc, H, nres, nres_out); }


// The two stage-1 kernels (and the two stage-2 kernels) are independent of each other: one launch per stage.
struct StitchArgs {
  int n, nmodes;
  const float *acc_top, *accD, *accE, *accEB, *accHcc, *accbc, *nres;
  const double *adHost, *adTarget;
  double *Ctop, *Ccc, *Csc, *Ce, *H;
  float *nres_out;
  size_t mode_stride;
  int upperOnly;  // write only the block-upper triangle (+ calib rows): all the host solve reads
  DoneSignal sg;  // stage 2 -> host
};
__global__ __launch_bounds__(64) void k_stitch_stage1(StitchArgs a) {
  const int per = a.n * a.n + 20, ntop = per * a.nmodes;
  if ((int)blockIdx.x < ntop) stitch_top_pairs_body(blockIdx.x % per, blockIdx.x / per, a.n, a.acc_top, a.adHost, a.adTarget, a.Ctop, a.Ccc);
  else sc_MC_body(blockIdx.x - ntop, 0, a.n, a.accD, a.accE, a.accEB, a.adHost, a.adTarget, a.Csc, a.Ce);
}
__global__ __launch_bounds__(64) void k_stitch_stage2(StitchArgs a) {
  const int per = a.n * (a.n + 1) / 2 + 1, ntop = per * a.nmodes;
  if ((int)blockIdx.x < ntop) stitch_top_sum_body(blockIdx.x % per, blockIdx.x / per, a.n, a.Ccc, a.Ctop, a.H, a.mode_stride, a.upperOnly != 0);
  else sc_sum_body(blockIdx.x - ntop, 0, a.n, a.Csc, a.Ce, a.accHcc, a.accbc, a.H + 2 * a.mode_stride, a.nres, a.nres_out, a.upperOnly != 0);
  if (a.sg.ctr) {
    __syncthreads();  // all stores of the block issued
    if (threadIdx.x == 0) signal_block_done(a.sg);
  }
}

// ================================================================================================
// Schur complement in ABSOLUTE coordinates (the pipelined Gauss-Newton loop of a window without linearised residuals).
//
// The reference accumulates the Schur blocks per (host, target1, target2) in relative coordinates and stitches the n^3 blocks
// with the adjoints afterwards (OB/AccumulatedSCHessian.cpp:63-77, 105-139).  The stitch is linear, so it can be applied to the
// per-residual row JpJdF BEFORE the products: with  w_p = [ sum_t adHost[h,t] JpJd_t  (block of the host h) ;
// adTarget[h,t] JpJd_t (block of target t) ; Hcd ; bdSum ]  the stitched system is  H_sc|b_sc = sum_p Hdi_p w_p w_p^T  -- ONE Gram
// product of dimension 8 n + 5 per chunk of points, whose sum over the chunks IS the (4 + 8 n)-dimensional H_sc / b_sc.  The n^3
// blocks, their reduction and both stages of their stitch disappear from the iteration.  fp32 like the reference's accumulators
// (adjoints in their fp32 copies, OB/EnergyFunctional.cpp:94-98, products on the fp32 matrix cores), chunk sums in fp64.
// ================================================================================================
static inline size_t gram_abs_lds_floats(int n, int ld) { return (size_t)SOS_GC * ld + SOS_GC + (size_t)SOS_GC * n * 8 + 2 * (size_t)n * 64; }
// (the chunk `blk` of a 256-thread workgroup; every thread of the workgroup passes the same barriers)
__device__ __forceinline__ void gram_abs_body(const BaDev &d, int blk, const int *__restrict__ chunk_pt, int Dm, int ld, float *__restrict__ gram_part,
                                              const float *__restrict__ adHF, const float *__restrict__ adTF, float *smem) {
  const int n = d.n, tid = threadIdx.x;
  float *A = smem;                      // [SOS_GC][ld]: the rows w_p
  float *sHdi = A + SOS_GC * ld;        // [SOS_GC]
  float *W = sHdi + SOS_GC;             // [SOS_GC][n][8]: adHost[h,t] JpJd_t per (point, target), summed into the host block below
  float *sAH = W + SOS_GC * n * 8;      // [n][64] adjoints of the chunk's host
  float *sAT = sAH + n * 64;
  const int p0 = chunk_pt[blk * SOS_GC];  // a chunk starts with a real point; all its points share the host
  const int h = p0 >= 0 ? d.pts[p0].host : 0;
  for (int q = tid; q < (SOS_GC * ld + SOS_GC + SOS_GC * n * 8) / 4; q += 256) reinterpret_cast<float4 *>(smem)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = tid; q < n * 16; q += 256) {
    const int t = q >> 4, k4 = q & 15;
    reinterpret_cast<float4 *>(sAH)[q] = reinterpret_cast<const float4 *>(adHF + 64 * (size_t)(h + n * t))[k4];
    reinterpret_cast<float4 *>(sAT)[q] = reinterpret_cast<const float4 *>(adTF + 64 * (size_t)(h + n * t))[k4];
  }
  __syncthreads();
  for (int q = tid; q < SOS_GC * (n + 1); q += 256) {
    const int t = q / SOS_GC, pl = q - t * SOS_GC;  // the 32 prep items (t == n) end up in one half-wave
    const int p = chunk_pt[blk * SOS_GC + pl];
    if (p < 0) continue;
    if (t == n) {
      const PrepOut r = point_prep_body(d, 1, p);
      float *row = A + pl * ld + 8 * n;
      sHdi[pl] = r.hdi;
      row[0] = r.hcd[0]; row[1] = r.hcd[1]; row[2] = r.hcd[2]; row[3] = r.hcd[3];
      row[4] = r.bdsum;
    } else {
      const int s = d.p_res_t[(size_t)p * n + t];
      if (s >= 0) {
        const float4 *jp = reinterpret_cast<const float4 *>(d.JpJd + 8 * (size_t)s);
        const float4 j0 = jp[0], j1 = jp[1];
        const float J[8] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w};
        const float *ah = sAH + 64 * t, *at = sAT + 64 * t;
        float *rowT = A + pl * ld + 8 * t, *rowH = W + (pl * n + t) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          float vt = 0.f, vh = 0.f;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            vt += at[8 * i + j] * J[j];
            vh += ah[8 * i + j] * J[j];
          }
          rowT[i] = vt;
          rowH[i] = vh;
        }
      }
    }
  }
  __syncthreads();
  for (int q = tid; q < SOS_GC * 8; q += 256) {  // the host's own block: sum over the targets, t ascending
    const int pl = q >> 3, i = q & 7;
    float sv = 0.f;
    for (int t = 0; t < n; t++) sv += W[(pl * n + t) * 8 + i];
    A[pl * ld + 8 * h + i] = sv;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int T = Dm >> 4;
  const int kq = lane >> 4, col = lane & 15;
  float *g = gram_part + (size_t)blk * Dm * Dm;
  for (int ut = wave; ut < T * (T + 1) / 2; ut += 4) {  // the 16x16 tiles on and above the diagonal
    int mt = 0, rem = ut;
    while (rem >= T - mt) { rem -= T - mt; mt++; }
    const int m0 = mt << 4, n0 = (mt + rem) << 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < SOS_GC / 4; kk++) {
      const int k = kk * 4 + kq;
      const float av = sHdi[k] * A[k * ld + m0 + col];
      const float bv = A[k * ld + n0 + col];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rgi = 0; rgi < 4; rgi++) g[(size_t)(m0 + kq * 4 + rgi) * Dm + n0 + col] = acc[rgi];
  }
}
__global__ __launch_bounds__(256) void k_sc_gram_abs(BaDev d, const int *__restrict__ chunk_pt, int Dm, int ld, float *__restrict__ gram_part,
                                                     const float *__restrict__ adHF, const float *__restrict__ adTF, int *flag, int seq) {
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gram_abs_body(d, blockIdx.x, chunk_pt, Dm, ld, gram_part, adHF, adTF, smem);
}

// One launch behind the Gram kernel:
//   blocks [0, n^2)   pair (h,t): the tile sums of the pair added in fp64, then the products of stitch stage 1
//                     (OB/AccumulatedTopHessian.cpp:261-288) from them; the pair's calibration entries and residual count aside
//   the rest          H_sc | b_sc: one block per row of a 16x16 tile of the Gram matrix, eight partial sums over the chunks per
//                     element (chunks k, k + 8, ...), added in that order; written where the host (or the exchange) reads them
struct AbsStitchArgs {
  int n, Dm, nchunks;
  const float *top_part;
  const int *pair_tile_begin;
  const float *gram_part;
  const double *adHost, *adTarget;
  double *Ctop, *Pcc;  // per pair: SOS_TOPC doubles of products; 24 doubles (16 Hcc, 4 bc, count)
  double *H;           // [H_A | b_A] at 0, [H_sc | b_sc] at mode_stride, the residual count at 2 * mode_stride
  size_t mode_stride;
  DoneSignal sg;       // k_abs_stitch2 -> host (SOS_ABS_SIGNAL_IN_KERNEL=1: the last block raises the flag, no k_publish behind it)
};
// (virtual block `vb`, worked by the first 128 threads of the workgroup; every thread of the workgroup passes the same barriers)
__device__ __forceinline__ void abs_rs1_body(const AbsStitchArgs &a, int vb) {
  const int n = a.n, tid = threadIdx.x;
  __shared__ double sSum[96], sB[64], sAH[64], sAT[64], sT1[64], sT2[64], sBpc[32], sbp[8], sPart[8][16];
  if (vb < n * n) {
    const int pidx = vb;
    if (tid < 96) {
      const int t0 = a.pair_tile_begin[pidx], t1 = a.pair_tile_begin[pidx + 1];
      double sv = 0;
#pragma unroll 8
      for (int t = t0; t < t1; t++) sv += (double)a.top_part[(size_t)t * SOS_TOPN + tid];
      sSum[tid] = sv;
    }
    if (tid < 64) {
      sAH[tid] = a.adHost[(size_t)pidx * 64 + tid];
      sAT[tid] = a.adTarget[(size_t)pidx * 64 + tid];
    }
    __syncthreads();
    if (tid < 24) {
      double v = 0;
      if (tid < 16) v = sSum[top_idx(tid >> 2, tid & 3)];
      else if (tid < 20) v = sSum[top_idx(tid - 16, 12)];
      else if (tid == 20) v = sSum[91];
      a.Pcc[(size_t)pidx * 24 + tid] = v;
    }
    const int i = (tid & 63) >> 3, j = tid & 7;
    if (tid < 64) sB[tid] = sSum[top_idx(4 + i, 4 + j)];
    if (tid < 32) sBpc[tid] = sSum[top_idx(4 + (tid >> 2), tid & 3)];
    if (tid < 8) sbp[tid] = sSum[top_idx(4 + tid, 12)];
    __syncthreads();
    if (tid < 64) {
      double t1 = 0, t2 = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        t1 += sAH[i * 8 + k] * sB[k * 8 + j];
        t2 += sAT[i * 8 + k] * sB[k * 8 + j];
      }
      sT1[tid] = t1;
      sT2[tid] = t2;
    }
    __syncthreads();
    if (tid >= 64) return;  // (no barrier below this line in this branch)
    double p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      p1 += sT1[i * 8 + k] * sAH[j * 8 + k];
      p2 += sT2[i * 8 + k] * sAT[j * 8 + k];
      p3 += sT1[i * 8 + k] * sAT[j * 8 + k];
    }
    double *out = a.Ctop + (size_t)pidx * SOS_TOPC;
    out[tid] = p1;
    out[64 + tid] = p2;
    out[128 + tid] = p3;
    if (tid < 32) {
      const int r = tid >> 2, c = tid & 3;
      double h1 = 0, h2 = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        h1 += sAH[r * 8 + k] * sBpc[k * 4 + c];
        h2 += sAT[r * 8 + k] * sBpc[k * 4 + c];
      }
      out[192 + tid] = h1;
      out[224 + tid] = h2;
    }
    if (tid < 8) {
      double b1 = 0, b2 = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        b1 += sAH[tid * 8 + k] * sbp[k];
        b2 += sAT[tid * 8 + k] * sbp[k];
      }
      out[256 + tid] = b1;
      out[264 + tid] = b2;
    }
    return;
  }
  // ---- H_sc | b_sc
  const int b = vb - n * n, T = a.Dm >> 4;
  const int ut = b >> 4, rr = b & 15;
  int mt = 0, rem = ut;
  while (rem >= T - mt) { rem -= T - mt; mt++; }
  const int r = (mt << 4) + rr, c = ((mt + rem) << 4) + (tid & 15), sub = (tid >> 4) & 7;
  const int cols = 8 * n + 5;
  double sv = 0;
  if (tid < 128 && r < cols - 1 && c < cols && c >= r) {
    const float *gp = a.gram_part + (size_t)r * a.Dm + c;
    const size_t stride = (size_t)a.Dm * a.Dm;
    for (int k0 = sub; k0 < a.nchunks; k0 += 64) {  // eight loads in flight per thread
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k0 + 8 * u < a.nchunks) v[u] = gp[(size_t)(k0 + 8 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k0 + 8 * u < a.nchunks) sv += (double)v[u];
    }
  }
  if (tid < 128) sPart[sub][tid & 15] = sv;
  __syncthreads();
  if (tid < 16 && r < cols - 1 && c < cols && c >= r) {
    double tot = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) tot += sPart[u][tid];
    const int dim = 4 + 8 * n;
    double *Hs = a.H + a.mode_stride, *bs = Hs + (size_t)dim * dim;
    const int R = r < 8 * n ? 4 + r : r - 8 * n;  // Gram order [frames | calib | b]  ->  system order [calib | frames]
    if (c == cols - 1) bs[R] = tot;
    else {
      const int Cc = c < 8 * n ? 4 + c : c - 8 * n;
      if (R <= Cc) Hs[(size_t)R * dim + Cc] = tot;
      else Hs[(size_t)Cc * dim + R] = tot;
    }
  }
}
__global__ __launch_bounds__(128) void k_abs_reduce_stitch1(AbsStitchArgs a) { abs_rs1_body(a, blockIdx.x); }
// ... and the second one: the sums of stitch stage 2 for the top system (upper triangle), the calibration block / b_c / the
// residual count from the per-pair entries
// (virtual block `vb`, worked by the first 64 threads of the workgroup; every thread of the workgroup passes the same barriers)
__device__ __forceinline__ void abs_st2_body(const AbsStitchArgs &a, int vb) {
  const int n = a.n, nblk = n * (n + 1) / 2, tid = threadIdx.x;
  if (vb < nblk) {
    if (tid < 64) stitch_top_sum_body(vb, 0, n, nullptr, a.Ctop, a.H, a.mode_stride, true);
    return;
  }
  // 21 scalars, each the sum over the n^2 pairs in pair order: three threads per scalar take a third of the pairs each
  __shared__ double sp[3][21];
  const int o = tid % 21, part = tid / 21;
  if (part < 3) {
    const int nn = n * n, per = (nn + 2) / 3, k0 = part * per, k1 = min(nn, k0 + per);
    double sv = 0;
    for (int k = k0; k < k1; k += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k + u < k1) v[u] = a.Pcc[(size_t)(k + u) * 24 + o];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k + u < k1) sv += v[u];
    }
    sp[part][o] = sv;
  }
  __syncthreads();
  if (tid < 21) {
    const double tot = (sp[0][tid] + sp[1][tid]) + sp[2][tid];
    const int dim = 4 + 8 * n;
    if (tid < 16) {
      if ((tid >> 2) <= (tid & 3)) a.H[(size_t)(tid >> 2) * dim + (tid & 3)] = tot;
    } else if (tid < 20) a.H[(size_t)dim * dim + (tid - 16)] = tot;
    else a.H[2 * a.mode_stride] = tot;
  }
}
__global__ __launch_bounds__(64) void k_abs_stitch2(AbsStitchArgs a) {
  abs_st2_body(a, blockIdx.x);
  if (a.sg.ctr) {
    __syncthreads();  // all stores of the block issued
    if (threadIdx.x == 0) signal_block_done(a.sg);
  }
}

// ------------------------------------------------------------------------------------------------
__global__ void k_copy_f64(double *__restrict__ dst, const double *__restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// ================================================================================================
// resubstituteFPt (OB/EnergyFunctional.cpp:526-551): one thread per point; loads of 4 residuals are in
// flight together, the subtraction order is the reference's.  With applyStep the point's idepth is
// advanced on the device exactly as doStepFromBackup does on the host
// (FS/FullSystemOptimize.cpp:207-213: setIdepth(backup + fac*step); setIdepthZero(same)).
// ================================================================================================
__device__ __forceinline__ void resubstitute_body(const BaDev &d, int p, const float *xc, const float *xAd,
                                                  float *__restrict__ step_out, int applyStep, float stepfacD) {
  float *o = d.p_out + 16 * (size_t)p;
  const float4 *ov = reinterpret_cast<const float4 *>(o);
  const float4 v0 = ov[0], v1 = ov[1], v2 = ov[2], v3 = ov[3];
  float step = 0.f;
  if (v3.x != 0.f) {  // HdiF == 0 <=> no active residual (ngoodres == 0)
    float b = v3.y;
    float dot = 0;
    dot += xc[0] * (v0.z + v2.x);
    dot += xc[1] * (v0.w + v2.y);
    dot += xc[2] * (v1.x + v2.z);
    dot += xc[3] * (v1.y + v2.w);
    b -= dot;
    const int q0 = d.p_begin[p], q1 = d.p_begin[p + 1];
    for (int q = q0; q < q1; q += 4) {
      float4 ja[4], jb[4], xa[4], xb[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int2 e = d.p_list2[min(q + k, q1 - 1)];
        const float4 *jp = reinterpret_cast<const float4 *>(d.JpJd + 8 * (size_t)e.x);
        const float4 *xp = reinterpret_cast<const float4 *>(xAd + 8 * (size_t)e.y);
        ja[k] = jp[0]; jb[k] = jp[1];
        xa[k] = xp[0]; xb[k] = xp[1];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (q + k >= q1) break;
        float dd = 0;  // inactive residuals hold JpJd == 0: dd == 0 and b - 0 == b
        dd += xa[k].x * ja[k].x; dd += xa[k].y * ja[k].y; dd += xa[k].z * ja[k].z; dd += xa[k].w * ja[k].w;
        dd += xb[k].x * jb[k].x; dd += xb[k].y * jb[k].y; dd += xb[k].z * jb[k].z; dd += xb[k].w * jb[k].w;
        b -= dd;
      }
    }
    step = -b * v3.x;
  }
  o[PO_STEP] = step;
  if (step_out) step_out[p] = step;
  if (applyStep) {
    sos_point *pt = d.pts + p;
    const float idn = pt->idepth_scaled + stepfacD * step;
    pt->idepth_scaled = idn;
    pt->idepth_zero_scaled = idn;
    pt->deltaF = 0.f;
    const int q0 = d.p_begin[p], q1 = d.p_begin[p + 1];
    for (int q = q0; q < q1; q++) {
      float2 *g = reinterpret_cast<float2 *>(d.r_geo + d.p_list2[q].x) + 1;
      *g = make_float2(idn, idn);
    }
  }
}
__global__ void k_resubstitute(BaDev d, const float *__restrict__ xc, const float *__restrict__ xAd,
                               float *__restrict__ step_out, int applyStep, float stepfacD) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  resubstitute_body(d, p, xc, xAd, step_out, applyStep, stepfacD);
}

// Per-tile copies of the precalc records (t_pre[tile] = precalc[t_pair[tile]], 7 float4 + 1 pad): the linearisation
// reads its pair's record with the tile index alone, one dependent load level less.  Item e = (tile, float4 q).
__device__ __forceinline__ void expand_precalc_item(int e, int ntiles, const int *__restrict__ t_pair, const float4 *__restrict__ pre,
                                                    float4 *__restrict__ t_pre) {
  const int tile = e >> 3, q = e & 7;
  if (tile >= ntiles || q == 7) return;
  t_pre[e] = pre[7 * (size_t)t_pair[tile] + q];
}
__global__ void k_expand_precalc(int ntiles, const int *__restrict__ t_pair, const float4 *__restrict__ pre, float4 *__restrict__ t_pre) {
  expand_precalc_item(blockIdx.x * blockDim.x + threadIdx.x, ntiles, t_pair, pre, t_pre);
}

// Fused per-iteration variant: the solved increment x arrives as a kernel argument; every block rebuilds the
// xAd table (OB/EnergyFunctional.cpp:503-516: xAd[h,t] = x_h^T adHostF + x_t^T adTargetF, fp32, summed left to
// right) in LDS from the resident fp32 adjoints, so the back-substitution depends on no staged data.  Blocks
// beyond the point blocks copy the per-step inputs of the following linearisation (precalc, adHTdelta, cDelta,
// frame thresholds) from the device-mapped pinned block into device memory.
struct XArg { float v[SOS_CPARS + 8 * SOS_MAX_FRAMES]; };
#ifndef SOS_RSB
#define SOS_RSB 128
#endif
// one stage-in block: the first nStageBlocks copy the per-step inputs, the rest write the per-tile precalc records
// straight from the mapped block
__device__ __forceinline__ void stage_block(int sb, int tid, const BaDev &d, float4 *__restrict__ stage_dst, const float4 *__restrict__ stage_src,
                                            int n4, int nStageBlocks, const float4 *__restrict__ pre_src, float4 *__restrict__ t_pre) {
  if (sb < nStageBlocks) {
    const int i = sb * SOS_RSB + tid;
    if (i < n4) stage_dst[i] = stage_src[i];
  } else {
    expand_precalc_item((sb - nStageBlocks) * SOS_RSB + tid, d.ntiles, d.t_pair, pre_src, t_pre);
  }
}
// the stage-in alone, for a linearisation whose back-substitution was enqueued ahead (sos_ba_gn_resub)
__global__ __launch_bounds__(SOS_RSB) void k_stage_expand(BaDev d, float4 *__restrict__ stage_dst, const float4 *__restrict__ stage_src, int n4,
                                                      int nStageBlocks, const float4 *__restrict__ pre_src, float4 *__restrict__ t_pre) {
  stage_block((int)blockIdx.x, threadIdx.x, d, stage_dst, stage_src, n4, nStageBlocks, pre_src, t_pre);
}
// nid_part != nullptr (device-resident loop): every wave leaves the sum of |idepth| of its points AFTER the step in nid_part[2 block +
// wave] -- doStepFromBackup's sumNID of the NEXT iteration (FS/FullSystemOptimize.cpp:207-213: it reads idepth_backup), so that the
// single-workgroup solve kernel does not have to stride through the point records itself
__device__ __forceinline__ void resub_point_block(const BaDev &d, const XArg *x, const float *__restrict__ adHF, const float *__restrict__ adTF,
                                                  float *__restrict__ step_out, float stepfacD, const float *__restrict__ x_dev, float *sxAd,
                                                  const double *xd = nullptr, double *nid_part = nullptr) {
  const int tid = threadIdx.x;
  const int n = d.n, dim = SOS_CPARS + 8 * n;
  float *sx = sxAd + n * n * 8;
  // ---- every global load of this thread's point is issued before the table is built: the memory latencies of the
  // point record, its residual list and the JpJd rows overlap with each other's and with the table build
  const int p = blockIdx.x * SOS_RSB + tid;
  const bool live = p < d.P;
  const int pp = live ? p : 0;
  const float4 *ov = reinterpret_cast<const float4 *>(d.p_out + 16 * (size_t)pp);
  const float4 v0 = ov[0], v1 = ov[1], v2 = ov[2], v3 = ov[3];
  const int q0 = d.p_begin[pp], q1 = live ? d.p_begin[pp + 1] : q0;
  constexpr int RU = 16;  // residuals held in registers; longer lists finish in the tail loop below
  int2 e[RU];
  float4 ja[RU], jb[RU];
  for (int i = tid; i < dim; i += SOS_RSB) sx[i] = xd ? (float)xd[i] : (x_dev ? x_dev[i] : x->v[i]);  // a kernel argument, or the device-resident loop's x
  __syncthreads();
  // The table (item = (pair idx = n*h + t, half jh): 4 of the 8 outputs from 8 + 8 float4 loads of the adjoint rows)
  // depends on nothing else: its loads are issued ahead of the point's dependent chain (list entries -> JpJd rows), so the
  // two run side by side instead of one after the other.
  const int nItems = n * n * 2;
  auto table_item = [&](int q, const float4 *ah, const float4 *at) {
    const int idx = q >> 1;
    const int h = idx / n, t = idx - h * n;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float xh = sx[SOS_CPARS + 8 * h + i], xt = sx[SOS_CPARS + 8 * t + i];
      s1.x += xh * ah[i].x; s1.y += xh * ah[i].y; s1.z += xh * ah[i].z; s1.w += xh * ah[i].w;
      s2.x += xt * at[i].x; s2.y += xt * at[i].y; s2.z += xt * at[i].z; s2.w += xt * at[i].w;
    }
    float4 o;
    o.x = s1.x + s2.x; o.y = s1.y + s2.y; o.z = s1.z + s2.z; o.w = s1.w + s2.w;
    reinterpret_cast<float4 *>(sxAd)[q] = o;
  };
  auto table_load = [&](int q, float4 *ah, float4 *at) {
    const int idx = q >> 1, jh = q & 1;
    const int h = idx / n, t = idx - h * n;
    const float4 *AH = reinterpret_cast<const float4 *>(adHF + 64 * (size_t)(h + n * t)) + jh;
    const float4 *AT = reinterpret_cast<const float4 *>(adTF + 64 * (size_t)(h + n * t)) + jh;
#pragma unroll
    for (int i = 0; i < 8; i++) { ah[i] = AH[2 * i]; at[i] = AT[2 * i]; }
  };
  {
    float4 ah[8], at[8];
    const bool first = tid < nItems;
    if (first) table_load(tid, ah, at);
#pragma unroll
    for (int k = 0; k < RU; k++) e[k] = d.p_list16[16 * (size_t)pp + k];  // fixed stride: no dependent lookup in front
    if (first) table_item(tid, ah, at);
  }
#pragma unroll
  for (int k = 0; k < RU; k++) {
    if (live && e[k].x >= 0) {
      const float4 *jp = reinterpret_cast<const float4 *>(d.JpJd + 8 * (size_t)e[k].x);
      ja[k] = jp[0];
      jb[k] = jp[1];
    } else {
      ja[k] = jb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int q = tid + SOS_RSB; q < nItems; q += SOS_RSB) {
    float4 ah[8], at[8];
    table_load(q, ah, at);
    table_item(q, ah, at);
  }
  __syncthreads();
  float idnAbs = 0.f;
  if (live) {
  // ---- resubstituteFPt (OB/EnergyFunctional.cpp:526-551), subtraction order = EFPoint::residualsAll order
  float step = 0.f;
  if (v3.x != 0.f) {  // HdiF == 0 <=> no active residual (ngoodres == 0)
    float b = v3.y;
    float dot = 0;
    dot += sx[0] * (v0.z + v2.x);
    dot += sx[1] * (v0.w + v2.y);
    dot += sx[2] * (v1.x + v2.z);
    dot += sx[3] * (v1.y + v2.w);
    b -= dot;
#pragma unroll
    for (int k = 0; k < RU; k++) {
      if (e[k].x >= 0) {
        const float4 *xp = reinterpret_cast<const float4 *>(sxAd + 8 * (size_t)e[k].y);
        const float4 xa = xp[0], xb = xp[1];
        float dd = 0;  // inactive residuals hold JpJd == 0: dd == 0 and b - 0 == b
        dd += xa.x * ja[k].x; dd += xa.y * ja[k].y; dd += xa.z * ja[k].z; dd += xa.w * ja[k].w;
        dd += xb.x * jb[k].x; dd += xb.y * jb[k].y; dd += xb.z * jb[k].z; dd += xb.w * jb[k].w;
        b -= dd;
      }
    }
    for (int q = q0 + RU; q < q1; q++) {  // windows with more than RU observations of one point
      const int2 ee = d.p_list2[q];
      const float4 *jp = reinterpret_cast<const float4 *>(d.JpJd + 8 * (size_t)ee.x);
      const float4 *xp = reinterpret_cast<const float4 *>(sxAd + 8 * (size_t)ee.y);
      const float4 a0 = jp[0], a1 = jp[1], xa = xp[0], xb = xp[1];
      float dd = 0;
      dd += xa.x * a0.x; dd += xa.y * a0.y; dd += xa.z * a0.z; dd += xa.w * a0.w;
      dd += xb.x * a1.x; dd += xb.y * a1.y; dd += xb.z * a1.z; dd += xb.w * a1.w;
      b -= dd;
    }
    step = -b * v3.x;
  }
  d.p_out[16 * (size_t)p + PO_STEP] = step;
  if (step_out) step_out[p] = step;
  // point part of doStepFromBackup (FS/FullSystemOptimize.cpp:207-213) + the per-residual copies of the point
  sos_point *pt = d.pts + p;
  const float idn = pt->idepth_scaled + stepfacD * step;
  pt->idepth_scaled = idn;
  pt->idepth_zero_scaled = idn;
  pt->deltaF = 0.f;
#pragma unroll
  for (int k = 0; k < RU; k++)
    if (e[k].x >= 0) *(reinterpret_cast<float2 *>(d.r_geo + e[k].x) + 1) = make_float2(idn, idn);
  for (int q = q0 + RU; q < q1; q++) *(reinterpret_cast<float2 *>(d.r_geo + d.p_list2[q].x) + 1) = make_float2(idn, idn);
  idnAbs = fabsf(idn);
  }
  if (nid_part) {  // (block-uniform)
    double v = (double)idnAbs;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((tid & 63) == 0) nid_part[(SOS_RSB / 64) * blockIdx.x + (tid >> 6)] = v;
  }
}
// the same partial sums from the point records as they are (once, when the device-resident loop begins)
__global__ __launch_bounds__(SOS_RSB) void k_nid_partials(const sos_point *__restrict__ pts, int P, double *__restrict__ nid_part) {
  const int p = blockIdx.x * SOS_RSB + threadIdx.x;
  double v = p < P ? (double)fabsf(pts[p].idepth_scaled) : 0.0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) nid_part[(SOS_RSB / 64) * blockIdx.x + (threadIdx.x >> 6)] = v;
}
__global__ __launch_bounds__(SOS_RSB) void k_resub_fused(BaDev d, XArg x, const float *__restrict__ adHF, const float *__restrict__ adTF,
                                                     float *__restrict__ step_out, float stepfacD, int nPointBlocks,
                                                     float4 *__restrict__ stage_dst, const float4 *__restrict__ stage_src, int n4,
                                                     int nStageBlocks, const float4 *__restrict__ pre_src, float4 *__restrict__ t_pre,
                                                     const float *__restrict__ x_dev) {
  extern __shared__ __attribute__((aligned(16))) float sxAd[];  // [n*n*8] table, then [dim] copy of x
  if ((int)blockIdx.x >= nPointBlocks) {
    stage_block((int)blockIdx.x - nPointBlocks, threadIdx.x, d, stage_dst, stage_src, n4, nStageBlocks, pre_src, t_pre);
    return;
  }
  resub_point_block(d, &x, adHF, adTF, step_out, stepfacD, x_dev, sxAd);
}

// ------------------------------------------------------------------------------------------------
// Device-side step of the fused loop (sos_ba_gn_devstep_begin): the blocks behind the point blocks do what the host does between
// the solve and the next linearisation -- doStepFromBackup for the calibration and the frames (state += -x, SE3::exp,
// PRE_camToWorld), FrameFramePrecalc::set for the n^2 ordered pairs, setDeltaF's adHTdeltaF / cDeltaF -- from x alone (a kernel
// argument, in double) and the frame states that stay on the device.  Every one of these blocks computes the (tiny) records
// itself in LDS and then writes its share of the per-tile precalc copies; the first one also writes the canonical arrays and
// the new states.  Nothing waits for the host's precalc any more, and the stage-in launch disappears.
// ------------------------------------------------------------------------------------------------
struct DevStep {
  const double *evalC2W, *state_zero, *state_in, *calib_in;  // n x 12, n x 10, n x 10, 4 value | 4 value_zero
  double *state_out, *calib_out;
  const double *abexp;                                        // n
  float *stage;                                               // device staging buffer of the linearisation
  size_t st_pre, st_adh, st_cd, st_th, st_cal;
  double xd[SOS_CPARS + 8 * 17];                              // x of the solve (the window sizes this path accepts: n <= 17)
  float th[17];                                               // frameEnergyTH of the coming linearisation
  // device-resident loop (sos_gn_resident.inc): x comes from k_gn_solve through device memory, the newest frame's threshold was written
  // by the accumulate's order-statistic block, and the right-hand-side part bM + HM delta of the NEXT solve (OB/EnergyFunctional.cpp:
  // 1046-1050: bM_top = bM + HM * getStitchedDeltaF()) is formed here from the stepped states
  const double *x_dev;                                        // nullptr: xd above
  int th_dev;                                                 // 1: leave the staged thresholds alone
  const double *HM, *bM;                                      // nullptr: no bMd
  double *bMd;
  double *nid_part;                                           // nullptr, or the per-wave sums of |idepth| after the step (resub_point_block)
};
__device__ __forceinline__ void devstep_block(int sb, int nsb, const BaDev &d, const DevStep &g, const float *__restrict__ adHF, const float *__restrict__ adTF,
                                              float4 *__restrict__ t_pre, float *smemf, const double *xd) {
  const int tid = threadIdx.x, n = d.n;
  // LDS: [28 n^2 floats: the records] then doubles: c2w 12 n | w2c 12 n | state 10 n | calib 4, then 8 floats K
  float *pre = smemf;
  double *dsm = reinterpret_cast<double *>(smemf + ((28 * n * n + 3) / 4) * 4 + 4);
  double *c2w = dsm, *w2c = dsm + 12 * n, *stN = dsm + 24 * n, *cvN = dsm + 34 * n;
  float *sK = reinterpret_cast<float *>(cvN + 4);
  // requested first, used last: the FEJ parts of this thread's pair record and the pair of its expansion tile (their latency
  // runs under the frame exponentials)
  const sos_precalc *old = reinterpret_cast<const sos_precalc *>(g.stage + g.st_pre);
  float fej[13];
  if (tid < n * n) {
    const sos_precalc *o = old + tid;
#pragma unroll
    for (int i = 0; i < 9; i++) fej[i] = o->PRE_RTll_0[i];
#pragma unroll
    for (int i = 0; i < 3; i++) fej[9 + i] = o->PRE_tTll_0[i];
    fej[12] = o->PRE_b0_mode;
  }
  const int ex_e = sb * SOS_RSB + tid, ex_tile = ex_e >> 3, ex_q = ex_e & 7;
  const int ex_pair = (ex_tile < d.ntiles && ex_q != 7) ? d.t_pair[ex_tile] : -1;
  if (tid < n) {  // FrameHessian::setState, FS/HessianBlocks.h:217-230
    const int f = tid;
    double st[10], scv[6];
    for (int i = 0; i < 8; i++) st[i] = g.state_in[10 * f + i] + (-xd[SOS_CPARS + 8 * f + i]);
    st[8] = g.state_in[10 * f + 8];
    st[9] = g.state_in[10 * f + 9];
    for (int i = 0; i < 10; i++) stN[10 * f + i] = st[i];
    for (int i = 0; i < 3; i++) scv[i] = (double)SOS_SCALE_XI_TRANS * st[i];
    for (int i = 3; i < 6; i++) scv[i] = (double)SOS_SCALE_XI_ROT * st[i];
    double R[9], t[3];
    sos_dev_se3_exp(scv, R, t);
    const double *E = g.evalC2W + 12 * f;
    double *C = c2w + 12 * f, *W = w2c + 12 * f;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) C[3 * i + j] = R[3 * i] * E[j] + R[3 * i + 1] * E[3 + j] + R[3 * i + 2] * E[6 + j];
      C[9 + i] = t[i] + (R[3 * i] * E[9] + R[3 * i + 1] * E[10] + R[3 * i + 2] * E[11]);
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) W[3 * i + j] = C[3 * j + i];
    for (int i = 0; i < 3; i++) W[9 + i] = -(W[3 * i] * C[9] + W[3 * i + 1] * C[10] + W[3 * i + 2] * C[11]);
  }
  if (tid == 32) {  // CalibHessian::setValue, FS/HessianBlocks.h:476-491 (a lane behind the <= 17 frame lanes)
    double vs[4];
    for (int i = 0; i < 4; i++) cvN[i] = g.calib_in[i] + (-xd[i]);
    vs[0] = SOS_SCALE_F * cvN[0]; vs[1] = SOS_SCALE_F * cvN[1]; vs[2] = SOS_SCALE_C * cvN[2]; vs[3] = SOS_SCALE_C * cvN[3];
    for (int i = 0; i < 4; i++) sK[i] = (float)vs[i];
    sK[4] = 1.0f / sK[0]; sK[5] = 1.0f / sK[1]; sK[6] = -sK[2] / sK[0]; sK[7] = -sK[3] / sK[1];
  }
  __syncthreads();
  for (int pidx = tid; pidx < n * n; pidx += SOS_RSB) {  // FrameFramePrecalc::set, FS/HessianBlocks.cpp:431-461
    const int h = pidx % n, t = pidx / n;
    const double *Wt = w2c + 12 * t, *Ch = c2w + 12 * h;
    float Rf[9], tf[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int j = 0; j < 3; j++) Rf[3 * i + j] = (float)(Wt[3 * i] * Ch[j] + Wt[3 * i + 1] * Ch[3 + j] + Wt[3 * i + 2] * Ch[6 + j]);
      tf[i] = (float)(Wt[9 + i] + (Wt[3 * i] * Ch[9] + Wt[3 * i + 1] * Ch[10] + Wt[3 * i + 2] * Ch[11]));
    }
    const float Km[9] = {sK[0], 0, sK[2], 0, sK[1], sK[3], 0, 0, 1};
    const float Ki[9] = {sK[4], 0, sK[6], 0, sK[5], sK[7], 0, 0, 1};
    float KR[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) KR[3 * i + j] = Km[3 * i] * Rf[j] + Km[3 * i + 1] * Rf[3 + j] + Km[3 * i + 2] * Rf[6 + j];
    sos_precalc *pc = reinterpret_cast<sos_precalc *>(pre) + pidx;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) pc->PRE_KRKiTll[3 * i + j] = KR[3 * i] * Ki[j] + KR[3 * i + 1] * Ki[3 + j] + KR[3 * i + 2] * Ki[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++) pc->PRE_KtTll[i] = Km[3 * i] * tf[0] + Km[3 * i + 1] * tf[1] + Km[3 * i + 2] * tf[2];
    if (pidx == tid) {  // the FEJ parts do not change inside the loop (first round of the loop: prefetched above)
#pragma unroll
      for (int i = 0; i < 9; i++) pc->PRE_RTll_0[i] = fej[i];
#pragma unroll
      for (int i = 0; i < 3; i++) pc->PRE_tTll_0[i] = fej[9 + i];
      pc->PRE_b0_mode = fej[12];
    } else {
      const sos_precalc *o = old + pidx;
#pragma unroll
      for (int i = 0; i < 9; i++) pc->PRE_RTll_0[i] = o->PRE_RTll_0[i];
#pragma unroll
      for (int i = 0; i < 3; i++) pc->PRE_tTll_0[i] = o->PRE_tTll_0[i];
      pc->PRE_b0_mode = o->PRE_b0_mode;
    }
    pc->pad = 0.f;
    float eF = (float)g.abexp[h], eT = (float)g.abexp[t];  // AffLight::fromToVecExposure
    if (eF == 0 || eT == 0) eT = eF = 1;
    const double ha = SOS_SCALE_A * stN[10 * h + 6], hb = SOS_SCALE_B * stN[10 * h + 7];
    const double ta = SOS_SCALE_A * stN[10 * t + 6], tb = SOS_SCALE_B * stN[10 * t + 7];
    const double aa = exp(ta - ha) * eT / eF;
    pc->PRE_aff_mode[0] = (float)aa;
    pc->PRE_aff_mode[1] = (float)(tb - aa * hb);
  }
  __syncthreads();
  // this block's share of the per-tile copies (expand_precalc_item from the records in LDS)
  if (ex_pair >= 0) t_pre[ex_e] = reinterpret_cast<const float4 *>(pre)[7 * (size_t)ex_pair + ex_q];
  // ---- the canonical arrays and the new states, spread over the blocks (every block holds all of it in LDS): role 0 = states /
  // calibration / thresholds, role 1 = the precalc array, roles 2.. = 256 entries of adHTdeltaF each
  const int nRolesStep = 2 + (8 * n * n + SOS_RSB - 1) / SOS_RSB;
  const int dimx = SOS_CPARS + 8 * n;
  const int nRoles = nRolesStep + (g.HM ? (dimx + SOS_RSB / 4 - 1) / (SOS_RSB / 4) : 0);
  for (int role = sb; role < nRoles; role += nsb) {
    if (role >= nRolesStep) {  // bMd = bM + HM delta, four lanes per row (j = q, q + 4, ...)
      const int i = (role - nRolesStep) * (SOS_RSB / 4) + (tid >> 2), q = tid & 3;
      double sv = 0;
      if (i < dimx) {
        const double *hm = g.HM + (size_t)i * dimx;
        for (int j = q; j < dimx; j += 4) {
          const double dj = j < SOS_CPARS ? (double)(float)(cvN[j] - g.calib_in[4 + j])
                                          : stN[10 * ((j - SOS_CPARS) >> 3) + ((j - SOS_CPARS) & 7)] - g.state_zero[10 * ((j - SOS_CPARS) >> 3) + ((j - SOS_CPARS) & 7)];
          sv = fma(hm[j], dj, sv);
        }
      }
      sv += __shfl_xor(sv, 1, 64);
      sv += __shfl_xor(sv, 2, 64);
      if (q == 0 && i < dimx) g.bMd[i] = g.bM[i] + sv;
      continue;
    }
    if (role == 0) {
      if (tid < 4) {
        g.stage[g.st_cd + tid] = (float)(cvN[tid] - g.calib_in[4 + tid]);
        g.calib_out[tid] = cvN[tid];
        g.calib_out[4 + tid] = g.calib_in[4 + tid];
      }
      if (tid < 8) g.stage[g.st_cal + tid] = sK[tid];
      if (tid < n && !g.th_dev) g.stage[g.st_th + tid] = g.th[tid];
      for (int q = tid; q < 10 * n; q += SOS_RSB) g.state_out[q] = stN[q];
    } else if (role == 1) {
      float4 *cpre = reinterpret_cast<float4 *>(g.stage + g.st_pre);
      for (int q = tid; q < 7 * n * n; q += SOS_RSB) cpre[q] = reinterpret_cast<const float4 *>(pre)[q];
    } else {  // setDeltaF, OB/EnergyFunctional.cpp:163-181 (fp32, i ascending)
      const int e = (role - 2) * SOS_RSB + tid;
      if (e < 8 * n * n) {
        const int pidx = e >> 3, j = e & 7;
        const int h = pidx % n, t = pidx / n;
        const float *AH = adHF + 64 * (size_t)pidx, *AT = adTF + 64 * (size_t)pidx;
        float s1 = 0, s2 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float dh = (float)(stN[10 * h + i] - g.state_zero[10 * h + i]), dt = (float)(stN[10 * t + i] - g.state_zero[10 * t + i]);
          s1 += dh * AH[8 * i + j];
          s2 += dt * AT[8 * i + j];
        }
        g.stage[g.st_adh + e] = s1 + s2;
      }
    }
  }
}
__global__ __launch_bounds__(SOS_RSB) void k_resub_devstep(BaDev d, const float *__restrict__ adHF, const float *__restrict__ adTF,
                                                       float *__restrict__ step_out, float stepfacD, int nPointBlocks, DevStep g,
                                                       float4 *__restrict__ t_pre) {
  extern __shared__ __attribute__((aligned(16))) float sxAd[];
  if ((int)blockIdx.x >= nPointBlocks) {
    devstep_block((int)blockIdx.x - nPointBlocks, (int)gridDim.x - nPointBlocks, d, g, adHF, adTF, t_pre, sxAd, g.x_dev ? g.x_dev : g.xd);
    return;
  }
  resub_point_block(d, nullptr, adHF, adTF, step_out, stepfacD, nullptr, sxAd, g.x_dev ? g.x_dev : g.xd, g.nid_part);
}
__global__ void k_fix_lin(BaDev d, const int *__restrict__ slist, int count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int s = slist[k];
  const float *Jt = d.J + (size_t)(s >> 5) * SOS_TILE_FLOATS + (s & 31);
#define JL(pl) Jt[(pl)*SOS_TILE]
  const int pair = d.t_pair[s >> 5];
  const float *dp = d.adHTdelta + 8 * pair;
  const float deltaF = d.pts[d.s_point[s]].deltaF;
  float dx = 0, dy = 0, dcx = 0, dcy = 0;
  for (int i = 0; i < 6; i++) { dx += JL(JP_DXI0 + i) * dp[i]; dy += JL(JP_DXI1 + i) * dp[i]; }
  for (int i = 0; i < 4; i++) { dcx += JL(JP_DC0 + i) * d.cdelta[i]; dcy += JL(JP_DC1 + i) * d.cdelta[i]; }
  const float Jp_delta_x = dx + dcx + JL(JP_DD) * deltaF;
  const float Jp_delta_y = dy + dcy + JL(JP_DD + 1) * deltaF;
  for (int i = 0; i < 8; i++) {
    float rtz = JL(JP_RESF + i);
    rtz = rtz - JL(JP_JIDX0 + i) * Jp_delta_x;
    rtz = rtz - JL(JP_JIDX1 + i) * Jp_delta_y;
    rtz = rtz - JL(JP_JAB0 + i) * dp[6];
    rtz = rtz - JL(JP_JAB1 + i) * dp[7];
    d.s_rtz[8 * (size_t)s + i] = rtz;
  }
#undef JL
  d.s_flags[s] = (uint8_t)(d.s_flags[s] | DF_LINEARIZED);
  // a residual linearised after the pack sits in an A tile no mode-0 pass visits any more: until the snapshot is
  // re-packed only the marginalisation pass (mode 2) gives it point terms again
  float4 *pt = reinterpret_cast<float4 *>(d.s_pterm + 8 * (size_t)s);
  pt[0] = pt[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// calcLEnergyPt residual part (OB/EnergyFunctional.cpp:571-613): per residual energy, summed in double
__global__ void k_lenergy(BaDev d, double *__restrict__ out_per_res) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.Rpad) return;
  double E = 0;
  if (d.s_point[s] >= 0) {
    const unsigned f = d.s_flags[s];
    if ((f & DF_LINEARIZED) && (f & DF_ACTIVE)) {
      const float *Jt = d.J + (size_t)(s >> 5) * SOS_TILE_FLOATS + (s & 31);
#define JL(pl) Jt[(pl)*SOS_TILE]
      const int pair = d.t_pair[s >> 5];
      const float *dp = d.adHTdelta + 8 * pair;
      const float dd = d.pts[d.s_point[s]].deltaF;
      float dx = 0, dy = 0, dcx = 0, dcy = 0;
      for (int i = 0; i < 6; i++) { dx += JL(JP_DXI0 + i) * dp[i]; dy += JL(JP_DXI1 + i) * dp[i]; }
      for (int i = 0; i < 4; i++) { dcx += JL(JP_DC0 + i) * d.cdelta[i]; dcy += JL(JP_DC1 + i) * d.cdelta[i]; }
      const float Jp_delta_x = dx + dcx + JL(JP_DD) * dd;
      const float Jp_delta_y = dy + dcy + JL(JP_DD + 1) * dd;
      for (int i = 0; i < 8; i++) {
        float Jdelta = JL(JP_JIDX0 + i) * Jp_delta_x;
        Jdelta = Jdelta + JL(JP_JIDX1 + i) * Jp_delta_y;
        Jdelta = Jdelta + JL(JP_JAB0 + i) * dp[6];
        Jdelta = Jdelta + JL(JP_JAB1 + i) * dp[7];
        float r0 = d.s_rtz[8 * (size_t)s + i];
        r0 = r0 + r0;
        r0 = r0 + Jdelta;
        E += (double)(Jdelta * r0);
      }
#undef JL
    }
  }
  out_per_res[s] = E;
}
__global__ __launch_bounds__(1024) void k_sum_double(const double *__restrict__ v, int n, double *out) {
  __shared__ double sm[1024];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 1024) a += v[i];
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
}

// copy the (idepth, idepth_zero) of every point into the per-residual records after a host-side update
// per-residual copies of the point record (r_geo, r_cw) from the uploaded point table: built on the device at pack
// time instead of on the host (saves 3.3 MB of host assembly + upload per keyframe at 12 KF x 4096 points)
__global__ void k_expand_points(BaDev d) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.Rpad) return;
  const int p = d.s_point[s];
  float4 g = make_float4(0.f, 0.f, 1.f, 1.f);
  float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, w0 = c0, w1 = c0;
  if (p >= 0) {
    const sos_point *q = d.pts + p;
    g = make_float4(q->u, q->v, q->idepth_scaled, q->idepth_zero_scaled);
    c0 = make_float4(q->color[0], q->color[1], q->color[2], q->color[3]);
    c1 = make_float4(q->color[4], q->color[5], q->color[6], q->color[7]);
    w0 = make_float4(q->weights[0], q->weights[1], q->weights[2], q->weights[3]);
    w1 = make_float4(q->weights[4], q->weights[5], q->weights[6], q->weights[7]);
  }
  d.r_geo[s] = g;
  float4 *o = reinterpret_cast<float4 *>(const_cast<float *>(d.r_cw) + 16 * (size_t)s);  // (color, weight) pairs per pattern pixel
  o[0] = make_float4(c0.x, w0.x, c0.y, w0.y); o[1] = make_float4(c0.z, w0.z, c0.w, w0.w);
  o[2] = make_float4(c1.x, w1.x, c1.y, w1.y); o[3] = make_float4(c1.z, w1.z, c1.w, w1.w);
}
__global__ void k_unsort_center(BaDev d) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.Rpad) return;
  const int o = d.s_orig[s];
  if (o < 0) return;
  d.o_center[3 * o] = d.s_center[3 * s]; d.o_center[3 * o + 1] = d.s_center[3 * s + 1]; d.o_center[3 * o + 2] = d.s_center[3 * s + 2];
}
__global__ void k_refresh_geo(BaDev d) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.Rpad) return;
  const int p = d.s_point[s];
  if (p < 0) return;
  float4 g = d.r_geo[s];
  g.z = d.pts[p].idepth_scaled;
  g.w = d.pts[p].idepth_zero_scaled;
  d.r_geo[s] = g;
}

// read one Jacobian back in the reference's 74-float layout
__global__ void k_get_jac(BaDev d, int s, sos_rawjac *out) {
  if (threadIdx.x != 0) return;
  const float *Jt = d.J + (size_t)(s >> 5) * SOS_TILE_FLOATS + (s & 31);
#define JL(pl) Jt[(pl)*SOS_TILE]
  for (int i = 0; i < 8; i++) {
    out->resF[i] = JL(JP_RESF + i);
    out->JIdx[0][i] = JL(JP_JIDX0 + i);
    out->JIdx[1][i] = JL(JP_JIDX1 + i);
    out->JabF[0][i] = JL(JP_JAB0 + i);
    out->JabF[1][i] = JL(JP_JAB1 + i);
  }
  for (int i = 0; i < 6; i++) { out->Jpdxi[0][i] = JL(JP_DXI0 + i); out->Jpdxi[1][i] = JL(JP_DXI1 + i); }
  for (int i = 0; i < 4; i++) { out->Jpdc[0][i] = JL(JP_DC0 + i); out->Jpdc[1][i] = JL(JP_DC1 + i); }
  out->Jpdd[0] = JL(JP_DD); out->Jpdd[1] = JL(JP_DD + 1);
  out->JIdx2[0] = JL(JP_JIDX2); out->JIdx2[1] = out->JIdx2[2] = JL(JP_JIDX2 + 1); out->JIdx2[3] = JL(JP_JIDX2 + 2);
  for (int i = 0; i < 4; i++) out->JabJIdx[i] = JL(JP_JABJIDX + i);
  out->Jab2[0] = JL(JP_JAB2); out->Jab2[1] = out->Jab2[2] = JL(JP_JAB2 + 1); out->Jab2[3] = JL(JP_JAB2 + 2);
#undef JL
}
// scatter a 74-float Jacobian into the tile layout (frozen J of linearized residuals at set_window)
__global__ void k_put_jac(BaDev d, const int *__restrict__ slist, const sos_rawjac *__restrict__ src, int count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int s = slist[k];
  const sos_rawjac *in = src + k;
  float *Jt = d.J + (size_t)(s >> 5) * SOS_TILE_FLOATS + (s & 31);
#define JL(pl) Jt[(pl)*SOS_TILE]
  for (int i = 0; i < 8; i++) {
    JL(JP_RESF + i) = in->resF[i];
    JL(JP_JIDX0 + i) = in->JIdx[0][i];
    JL(JP_JIDX1 + i) = in->JIdx[1][i];
    JL(JP_JAB0 + i) = in->JabF[0][i];
    JL(JP_JAB1 + i) = in->JabF[1][i];
  }
  for (int i = 0; i < 6; i++) { JL(JP_DXI0 + i) = in->Jpdxi[0][i]; JL(JP_DXI1 + i) = in->Jpdxi[1][i]; }
  for (int i = 0; i < 4; i++) { JL(JP_DC0 + i) = in->Jpdc[0][i]; JL(JP_DC1 + i) = in->Jpdc[1][i]; }
  JL(JP_DD) = in->Jpdd[0]; JL(JP_DD + 1) = in->Jpdd[1];
  JL(JP_JIDX2) = in->JIdx2[0]; JL(JP_JIDX2 + 1) = in->JIdx2[1]; JL(JP_JIDX2 + 2) = in->JIdx2[3];
  for (int i = 0; i < 4; i++) JL(JP_JABJIDX + i) = in->JabJIdx[i];
  JL(JP_JAB2) = in->Jab2[0]; JL(JP_JAB2 + 1) = in->Jab2[1]; JL(JP_JAB2 + 2) = in->Jab2[3];
#undef JL
  // JpJdF of the frozen Jacobian (takeDataF was applied before the residual was linearized)
  const float v0 = in->JIdx2[0] * in->Jpdd[0] + in->JIdx2[1] * in->Jpdd[1];
  const float v1 = in->JIdx2[2] * in->Jpdd[0] + in->JIdx2[3] * in->Jpdd[1];
  float *o = d.JpJd + 8 * (size_t)s;
  for (int i = 0; i < 6; i++) o[i] = in->Jpdxi[0][i] * v0 + in->Jpdxi[1][i] * v1;
  o[6] = in->JabJIdx[0] * in->Jpdd[0] + in->JabJIdx[1] * in->Jpdd[1];
  o[7] = in->JabJIdx[2] * in->Jpdd[0] + in->JabJIdx[3] * in->Jpdd[1];
}

// ================================================================================================
// host side of the handle
// ================================================================================================
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  bool view = false;  // a sub-buffer of a slab owned elsewhere (sos_ba_set_window)
  int ensure(size_t n) {
    if (n <= cap) return SOS_OK;
    if (p && !view) hipFree(p);
    p = nullptr;
    cap = 0;
    view = false;
    size_t want = n + n / 4 + 64;
    if (hipMalloc(&p, sizeof(T) * want) != hipSuccess) return SOS_ERR_NOMEM;
    cap = want;
    return SOS_OK;
  }
  void release() {
    if (p && !view) hipFree(p);
    p = nullptr;
    cap = 0;
    view = false;
  }
};

static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct sos_ba {
  double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // SOS_TIMING=1: seconds in the phases of sos_ba_gn_step / gn_accumulate
  long tm_calls = 0;
  sos_ctx *ctx = nullptr;
  sos_params prm;
  bool have_window = false, have_state = false;
  int n = 0, P = 0, R = 0, Rpad = 0, ntiles = 0, ntilesA = 0, nchunks = 0, Dm = 0, ld = 0;
  int slot[SOS_MAX_FRAMES];
  std::vector<int> h_s_of_orig;   // original residual index -> sorted index
  std::vector<int> h_pair_tile_begin;  // [2*n*n + 1]
  std::vector<sos_point> h_pts;
  std::vector<sos_resid> h_res;
  std::vector<int> h_p_begin;
  std::vector<int> h_keys, h_kcount;  // sos_ba_set_window scratch (kept: no allocation per keyframe)
  // per-window slabs: every index table of the snapshot is built in ONE pinned block and uploaded with one copy; the
  // scratch that has to start at zero is ONE device block cleared by one kernel.  The DevBufs of those tables are views.
  char *up_host = nullptr, *up_dev = nullptr, *zr_dev = nullptr;
  size_t up_cap = 0, zr_cap = 0;
  char *fin_host = nullptr, *fin_dev = nullptr;  // sos_ba_linearize_final: packed results, device side and pinned host side
  size_t fin_cap = 0;
  int adj_n = 0;
  size_t up_pts_off = 0;    // offset of the point records in the upload slab
  char *adj_host = nullptr, *adj_dev = nullptr;  // adjoints [adHost | adTarget] fp64 + [adHostF | adTargetF] fp32: one pinned block, one copy
  size_t adj_cap = 0;
  sos_calib calib;
  // device buffers
  DevBuf<sos_point> d_pts;
  DevBuf<int> d_s_point, d_s_orig, d_t_pair, d_p_begin, d_p_list, d_p_res_t, d_pair_tile_begin, d_chunk_pt,
      d_host_chunk_begin, d_tmp_int;
  DevBuf<uint8_t> d_s_flags, d_s_state, d_s_newstate, d_o_newstate;
  DevBuf<float> d_s_energy, d_s_newenergy, d_s_newenergywo, d_s_ret, d_s_center, d_s_rtz, d_s_pterm, d_J, d_JpJd,
      d_p_out, d_o_newenergy, d_o_newenergywo, d_o_center, d_top_part, d_gram_part, d_acc, d_calib;
  DevBuf<double> d_adHost, d_adTarget, d_Hout, d_scalar, d_perres;
  DevBuf<float4> d_t_pre;
  DevBuf<const float *> d_t_img;
  DevBuf<int> d_t_ht;
  DevBuf<sos_rawjac> d_rawjac;
  DevBuf<int2> d_p_list2, d_p_list16;
  DevBuf<float4> d_r_geo;
  DevBuf<float> d_r_cw;
  DevBuf<float> d_stage;   // [precalc n*n*28 | adHTdelta n*n*8 | cdelta 4 | frameTH n(+pad) | calib 8 | xc 4 | xAd n*n*8]
  DevBuf<char> d_outpack;  // [tile_esum ntilesA doubles | newest energies | point steps]
  DevBuf<double> d_C;      // stitch stage-1 products
  DevBuf<double> d_ar64;   // sos_ba_allreduce_f64 staging
  DevBuf<float> d_xchg;    // sos_ba_time_kernel("exchange") scratch
  DevBuf<float> d_large;   // sos_ba_time_kernel("stream_large") 1 GiB yardstick buffer
  // device-resident Gauss-Newton loop (sos_gn_resident.inc)
  DevBuf<double> d_gn;       // HM | bM | prior | bMd (= bM + HM delta of the coming solve) | x (the solve's result, read by the step) | per-wave sums of |idepth|
  size_t gn_HM = 0, gn_bM = 0, gn_prior = 0, gn_bMd = 0, gn_x = 0, gn_nid = 0;
  double gn_cPrior = 0;
  double *gn_pin = nullptr, *gn_pin_dev = nullptr;  // mapped ring of GN_SLOTS result slots + flag
  size_t gn_pin_doubles = 0, gn_slot_doubles = 0;
  bool gn_active = false, gn_have_top = false;
  int gn_seq = 0;
  DevBuf<float> d_Jnew, d_JpJd_new, d_pterm_new;  // PointFrameResidual::J side of the two-step protocol (lazily allocated)
  bool pending_new = false;  // a sos_ba_linearize result waits in d_Jnew for sos_ba_apply_res
  size_t st_pre = 0, st_adh = 0, st_cd = 0, st_th = 0, st_cal = 0, st_xc = 0, st_xad = 0, st_floats = 0;
  bool resub_pending = false;  // sos_ba_gn_resub enqueued the back-substitution of the step sos_ba_gn_step is about to take
  // device-side step (sos_ba_gn_devstep_begin): frame states that stay on the device between the iterations of one optimize()
  bool devstep = false;
  double *ds_host = nullptr; // pinned source of the upload below (two halves)
  int ds_flip = 0;
  double *d_ds = nullptr;    // evalC2W 12 n | state_zero 10 n | state[0] 10 n | state[1] 10 n | calib[0] 8 | calib[1] 8 | ab_exposure n
  int ds_n = 0, ds_cur = 0;
  size_t out_esum = 0, out_newest = 0, out_step = 0, out_bytes = 0;
  int newest_begin = 0, newest_count = 0;
  char *pin = nullptr;     // pinned + device-mapped host block: [stage | outpack | Hb]; the fused per-iteration
  char *pin_dev = nullptr; // calls let kernels read / write it directly (no copy commands on the critical path)
  size_t pin_bytes = 0, pin_stage = 0, pin_out = 0, pin_hb = 0, pin_flags = 0;  // flags: 2 ints, 64 B apart
  bool prefetch = false;      // sos_ba_set_prefetch: gn_step enqueues the next gn_accumulate behind the linearisation
  bool acc_inflight = false;  // ... and this says its result is (or will be) in the mapped Hb block
  bool acc_inflight_haveL = false;
  bool acc_inflight_abs = false;  // ... in the layout of the absolute-coordinate path
  bool top_valid = false;     // d_top_part holds the tile sums of the current linearisation (the last one ran fused and nothing changed since)
  bool fuse_only = false;     // sos_ba_set_prefetch(ba, 2): the linearisation forms the tile sums, nothing is enqueued behind it
  DevBuf<int> d_sigctr;       // [0] linearize launches, [1] stitch launches (cumulative block counters)
  int sig_lin_seq = 0, sig_st_seq = 0, sig_st_blocks_total = 0, sig_abs_blocks_total = 0;
  int acc_wait_seq = 0;    // sequence number the stitch of the accumulate to be consumed NEXT publishes
  struct StepPend {        // a step whose launches are enqueued and whose results have not been collected yet
    bool active = false, prefetchEnqueued = false, havePointStep = false;
    int waitSeq = 0;
    float stepfacD = 0;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    // a pre-launched step leaves the enqueue of the next accumulate to its delivery (whether there is a next iteration is known after
    // the solve, and an accumulate nobody consumes would overwrite the per-point outputs of the last one): what that enqueue needs
    bool fuseTop = false, chainPublish = false, applyRes = false;
  } pend;
  // multi-GPU (sos_ba_set_comm): common capacity of the newest-frame energy lists, local / gathered device lists and
  // the device-mapped host copy of the gathered list
  sos_comm *comm = nullptr;
  int comm_size = 1, newest_cap = 0;
  bool anyL = false;  // some rank holds linearised residuals
  bool anyEmpty = false;  // some rank's shard has no residuals / no points
  DevBuf<float> d_newest_local, d_newest_all;
  float *pin_newest = nullptr, *pin_newest_dev = nullptr;
  size_t pin_newest_floats = 0;
  bool J_valid = true;        // false after a pipelined sos_ba_gn_step: the tiles were consumed on chip, d_J is stale
  hipEvent_t ev_step = nullptr;
  size_t hb_mode_stride = 0;  // doubles per (H | b) block in d_Hout
  std::vector<float> h_adHostF, h_adTargetF;
  DevBuf<float> d_adHostF, d_adTargetF;  // fp32 copies for the in-kernel xAd table
  size_t acc_floats = 0;
  // offsets into the packed accumulator
  size_t off_topA = 0, off_topL = 0, off_D = 0, off_E = 0, off_EB = 0, off_Hcc = 0, off_bc = 0, off_nres = 0;
  BaDev dev;
};

static inline int divup(int a, int b) { return (a + b - 1) / b; }

extern "C" int sos_ba_create(sos_ctx *ctx, const sos_params *prm, sos_ba **out) {
  if (!ctx || !prm || !out) return SOS_ERR_ARG;
  if (prm->w != ctx->w || prm->h != ctx->h) return SOS_ERR_ARG;
  sos_ba *ba = new sos_ba();
  ba->ctx = ctx;
  ba->prm = *prm;
  memset(&ba->dev, 0, sizeof(ba->dev));
  if (hipSetDevice(ctx->device) != hipSuccess || hipEventCreateWithFlags(&ba->ev_step, hipEventDisableTiming) != hipSuccess) {
    delete ba;
    return SOS_ERR_HIP;
  }
  *out = ba;
  return SOS_OK;
}

extern "C" int sos_ba_set_prefetch(sos_ba *ba, int on) {
  if (!ba) return SOS_ERR_ARG;
  ba->prefetch = on == 1;
  ba->fuse_only = on == 2;  // the tile sums of the top Hessian come out of the next linearisation, but no accumulate is enqueued behind it
  if (on != 1) ba->acc_inflight = false;
  return SOS_OK;
}

extern "C" int sos_ba_destroy(sos_ba *ba) {
  if (!ba) return SOS_OK;
  if (getenv("SOS_TIMING") && ba->tm_calls) {
    const double k = 1e6 / (double)ba->tm_calls;
    fprintf(stderr, "[sos_ba timing, us/call over %ld gn_step calls] stage-memcpy %.1f fill_x %.1f launch(step) %.1f launch(prefetch) %.1f "
                    "wait(step) %.1f unpack %.1f | gn_accumulate wait %.1f\n",
            ba->tm_calls, ba->tm[0] * k, ba->tm[1] * k, (ba->tm[2] - ba->tm[1]) * k, ba->tm[3] * k, ba->tm[4] * k, ba->tm[5] * k, ba->tm[6] * k);
  }
  hipSetDevice(ba->ctx->device);
  hipStreamSynchronize(ba->ctx->stream);
  if (ba->d_ds) hipFree(ba->d_ds);
  if (ba->ds_host) hipHostFree(ba->ds_host);
  if (ba->up_host) hipHostFree(ba->up_host);
  if (ba->up_dev) hipFree(ba->up_dev);
  if (ba->zr_dev) hipFree(ba->zr_dev);
  if (ba->fin_host) hipHostFree(ba->fin_host);
  if (ba->fin_dev) hipFree(ba->fin_dev);
  if (ba->adj_host) hipHostFree(ba->adj_host);
  if (ba->adj_dev) hipFree(ba->adj_dev);
  ba->d_pts.release();
  for (DevBuf<int> *b : {&ba->d_s_point, &ba->d_s_orig, &ba->d_t_pair, &ba->d_p_begin, &ba->d_p_list, &ba->d_p_res_t,
                         &ba->d_pair_tile_begin, &ba->d_chunk_pt, &ba->d_host_chunk_begin, &ba->d_tmp_int, &ba->d_sigctr})
    b->release();
  for (DevBuf<uint8_t> *b : {&ba->d_s_flags, &ba->d_s_state, &ba->d_s_newstate, &ba->d_o_newstate}) b->release();
  for (DevBuf<float> *b :
       {&ba->d_s_energy, &ba->d_s_newenergy, &ba->d_s_newenergywo, &ba->d_s_ret, &ba->d_s_center, &ba->d_s_rtz,
        &ba->d_s_pterm, &ba->d_J, &ba->d_JpJd, &ba->d_p_out, &ba->d_o_newenergy, &ba->d_o_newenergywo, &ba->d_o_center,
        &ba->d_top_part, &ba->d_gram_part, &ba->d_acc, &ba->d_adHostF, &ba->d_adTargetF, &ba->d_calib})
    b->release();
  for (DevBuf<double> *b : {&ba->d_adHost, &ba->d_adTarget, &ba->d_Hout, &ba->d_scalar, &ba->d_perres})
    b->release();
  ba->d_rawjac.release();
  ba->d_t_pre.release(); ba->d_t_img.release(); ba->d_t_ht.release();
  ba->d_p_list2.release(); ba->d_p_list16.release(); ba->d_r_geo.release(); ba->d_r_cw.release(); ba->d_stage.release(); ba->d_outpack.release(); ba->d_C.release(); ba->d_ar64.release(); ba->d_xchg.release(); ba->d_large.release(); ba->d_gn.release();
  if (ba->gn_pin) hipHostFree(ba->gn_pin); ba->d_Jnew.release(); ba->d_JpJd_new.release(); ba->d_pterm_new.release();
  if (ba->pin) hipHostFree(ba->pin);
  if (ba->ev_step) hipEventDestroy(ba->ev_step);
  if (ba->pin_newest) hipHostFree(ba->pin_newest);
  ba->d_newest_local.release(); ba->d_newest_all.release();
  delete ba;
  return SOS_OK;
}

#define ENSURE(buf, n)              \
  do {                              \
    int rc_ = (buf).ensure(n);      \
    if (rc_) return rc_;            \
  } while (0)

template <typename T>
static int upload(hipStream_t st, DevBuf<T> &buf, const std::vector<T> &v) {
  int rc = buf.ensure(v.size() ? v.size() : 1);
  if (rc) return rc;
  if (!v.empty()) SOS_HIP(hipMemcpyAsync(buf.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, st));
  return SOS_OK;
}

static int comm_setup_window(sos_ba *ba);

// zero fill of the per-window scratch slab (one launch instead of a dozen memset commands)
__global__ __launch_bounds__(256) void k_zero_slab(float4 *__restrict__ p, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// bump allocator over a slab: 256-byte aligned sub-buffers
struct SlabLayout {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 255) / 256 * 256;
    return o;
  }
};
template <typename T>
static inline void slab_view(DevBuf<T> &b, char *base, size_t off, size_t count) {
  b.release();
  b.p = reinterpret_cast<T *>(base + off);
  b.cap = count;
  b.view = true;
}

extern "C" int sos_ba_set_window(sos_ba *ba, int n, const int32_t *frame_slot, int P, const sos_point *pts, int R,
                                 const sos_resid *res, const float *res_toZeroF, const sos_rawjac *lin_J) {
  if (ba) ba->devstep = false;  // the device-side frame states belong to the previous snapshot
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || n < 1 || n > SOS_MAX_FRAMES || P < 0 || R < 0 || !frame_slot || (P && !pts) || (R && !res)) return SOS_ERR_ARG;
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  SOS_HIP(hipStreamSynchronize(st));  // (also: the previous snapshot's upload has left the pinned slab)
  const double tw0 = now_s();
  for (int i = 0; i < n; i++) {
    if (frame_slot[i] < 0 || frame_slot[i] >= SOS_MAX_SLOTS || !c->dI[frame_slot[i]][0]) return SOS_ERR_STATE;
    ba->slot[i] = frame_slot[i];
  }
  const int nn2 = 2 * n * n;
  // ---- pass 1: validate the graph (residuals contiguous per point, in point order; allPoints order: frames -> points),
  // key = (isLinearized, pair) of every residual and the key histogram
  std::vector<int> &p_begin = ba->h_p_begin, &keys = ba->h_keys, &kcount = ba->h_kcount;
  p_begin.resize((size_t)P + 1);
  keys.resize(R ? R : 1);
  kcount.assign((size_t)nn2 + 1, 0);
  {
    int r = 0;
    for (int p = 0; p < P; p++) {
      if (p && pts[p].host < pts[p - 1].host) return SOS_ERR_ARG;
      p_begin[p] = r;
      const int ph = pts[p].host;
      while (r < R && res[r].point == p) {
        const sos_resid &q = res[r];
        if (q.host != ph || q.host < 0 || q.host >= n || q.target < 0 || q.target >= n) return SOS_ERR_ARG;
        const int k = ((q.flags & SOS_RF_LINEARIZED) ? n * n : 0) + q.host + n * q.target;
        keys[r] = k;
        kcount[k + 1]++;
        r++;
      }
    }
    p_begin[P] = r;
    if (r != R) return SOS_ERR_ARG;
  }
  // ---- tiles: every key padded to whole tiles of SOS_TILE residuals
  std::vector<int> &pair_tile_begin = ba->h_pair_tile_begin;
  pair_tile_begin.resize((size_t)nn2 + 1);
  int ntilesA = 0, ntiles = 0;
  for (int k = 0; k < nn2; k++) {
    pair_tile_begin[k] = ntiles;
    ntiles += (kcount[k + 1] + SOS_TILE - 1) / SOS_TILE;
    if (k == n * n - 1) ntilesA = ntiles;
  }
  pair_tile_begin[nn2] = ntiles;
  const int Rpad = ntiles * SOS_TILE;
  // chunks of SOS_GC points per host for the Gram kernel
  int nchunks = 0;
  std::vector<int> host_chunk_begin(n + 1, 0), host_pt_begin(n + 1, 0);
  {
    int p = 0;
    for (int h = 0; h < n; h++) {
      host_chunk_begin[h] = nchunks;
      host_pt_begin[h] = p;
      const int start = p;
      while (p < P && pts[p].host == h) p++;
      nchunks += (p - start + SOS_GC - 1) / SOS_GC;
    }
    host_chunk_begin[n] = nchunks;
    host_pt_begin[n] = P;
  }
  ba->n = n; ba->P = P; ba->R = R; ba->Rpad = Rpad; ba->ntiles = ntiles; ba->ntilesA = ntilesA;
  ba->nchunks = nchunks;
  ba->Dm = ((8 * n + 5) + 15) / 16 * 16;
  ba->ld = (ba->Dm % 32 == 16) ? ba->Dm : ba->Dm + 16;
  const size_t Rp = Rpad ? Rpad : 1, Pp = P ? P : 1, Rr = R ? R : 1, Tp = ntiles ? ntiles : 1;

  // ---- layout of the upload slab (built in pinned host memory, one H2D) and of the zeroed scratch slab
  SlabLayout up, zr;
  const size_t u_pts = up.take(sizeof(sos_point) * Pp), u_s_point = up.take(sizeof(int) * Rp), u_s_orig = up.take(sizeof(int) * Rp),
               u_t_pair = up.take(sizeof(int) * Tp), u_t_img = up.take(sizeof(const float *) * Tp), u_t_ht = up.take(sizeof(int) * Tp),
               u_p_begin = up.take(sizeof(int) * ((size_t)P + 1)), u_p_list = up.take(sizeof(int) * Rr),
               u_p_list2 = up.take(sizeof(int2) * Rr), u_p_list16 = up.take(sizeof(int2) * Pp * 16),
               u_p_res_t = up.take(sizeof(int) * Pp * n), u_ptb = up.take(sizeof(int) * ((size_t)nn2 + 1)),
               u_chunk = up.take(sizeof(int) * (size_t)(nchunks ? nchunks : 1) * SOS_GC), u_hcb = up.take(sizeof(int) * ((size_t)n + 1)),
               u_flags = up.take(Rp), u_state = up.take(Rp), u_newstate = up.take(Rp), u_energy = up.take(sizeof(float) * Rp),
               u_rtz = res_toZeroF ? up.take(sizeof(float) * Rp * 8) : 0;
  const size_t z_newenergy = zr.take(sizeof(float) * Rp), z_newenergywo = zr.take(sizeof(float) * Rp), z_ret = zr.take(sizeof(float) * Rp),
               z_center = zr.take(sizeof(float) * Rp * 3), z_pterm = zr.take(sizeof(float) * Rp * 8),
               z_J = zr.take(sizeof(float) * Tp * SOS_TILE_FLOATS), z_JpJd = zr.take(sizeof(float) * Rp * 8),
               z_pout = zr.take(sizeof(float) * Pp * 16), z_ocenter = zr.take(sizeof(float) * Rr * 3), z_sig = zr.take(sizeof(int) * 32),
               z_rtz = res_toZeroF ? 0 : zr.take(sizeof(float) * Rp * 8);
  if (up.off > ba->up_cap) {
    if (ba->up_host) hipHostFree(ba->up_host);
    if (ba->up_dev) hipFree(ba->up_dev);
    ba->up_host = ba->up_dev = nullptr;
    ba->up_cap = 0;
    const size_t want = up.off + up.off / 4 + 4096;
    SOS_HIP(hipHostMalloc((void **)&ba->up_host, want, hipHostMallocDefault));
    if (hipMalloc((void **)&ba->up_dev, want) != hipSuccess) return SOS_ERR_NOMEM;
    ba->up_cap = want;
  }
  if (zr.off > ba->zr_cap) {
    if (ba->zr_dev) hipFree(ba->zr_dev);
    ba->zr_dev = nullptr;
    ba->zr_cap = 0;
    const size_t want = zr.off + zr.off / 4 + 4096;
    if (hipMalloc((void **)&ba->zr_dev, want) != hipSuccess) return SOS_ERR_NOMEM;
    ba->zr_cap = want;
  }
  char *uh = ba->up_host, *ud = ba->up_dev, *zd = ba->zr_dev;
  slab_view(ba->d_pts, ud, u_pts, Pp); slab_view(ba->d_s_point, ud, u_s_point, Rp); slab_view(ba->d_s_orig, ud, u_s_orig, Rp);
  slab_view(ba->d_t_pair, ud, u_t_pair, Tp); slab_view(ba->d_t_img, ud, u_t_img, Tp); slab_view(ba->d_t_ht, ud, u_t_ht, Tp);
  slab_view(ba->d_p_begin, ud, u_p_begin, (size_t)P + 1); slab_view(ba->d_p_list, ud, u_p_list, Rr); slab_view(ba->d_p_list2, ud, u_p_list2, Rr);
  slab_view(ba->d_p_list16, ud, u_p_list16, Pp * 16); slab_view(ba->d_p_res_t, ud, u_p_res_t, Pp * n);
  slab_view(ba->d_pair_tile_begin, ud, u_ptb, (size_t)nn2 + 1); slab_view(ba->d_chunk_pt, ud, u_chunk, (size_t)(nchunks ? nchunks : 1) * SOS_GC);
  slab_view(ba->d_host_chunk_begin, ud, u_hcb, (size_t)n + 1);
  slab_view(ba->d_s_flags, ud, u_flags, Rp); slab_view(ba->d_s_state, ud, u_state, Rp); slab_view(ba->d_s_newstate, ud, u_newstate, Rp);
  slab_view(ba->d_s_energy, ud, u_energy, Rp);
  slab_view(ba->d_s_newenergy, zd, z_newenergy, Rp); slab_view(ba->d_s_newenergywo, zd, z_newenergywo, Rp); slab_view(ba->d_s_ret, zd, z_ret, Rp);
  slab_view(ba->d_s_center, zd, z_center, Rp * 3); slab_view(ba->d_s_pterm, zd, z_pterm, Rp * 8);
  slab_view(ba->d_J, zd, z_J, Tp * SOS_TILE_FLOATS); slab_view(ba->d_JpJd, zd, z_JpJd, Rp * 8); slab_view(ba->d_p_out, zd, z_pout, Pp * 16);
  slab_view(ba->d_o_center, zd, z_ocenter, Rr * 3); slab_view(ba->d_sigctr, zd, z_sig, 32);
  if (res_toZeroF) slab_view(ba->d_s_rtz, ud, u_rtz, Rp * 8);
  else slab_view(ba->d_s_rtz, zd, z_rtz, Rp * 8);

  // ---- pass 2: the tables, written where they are uploaded from.  Residuals sorted by (isLinearized, pair), stable in the
  // original (point) order: a counting sort over the 2 n^2 keys whose slots already include the tile padding
  ba->up_pts_off = u_pts;
  sos_point *h_pts_up = reinterpret_cast<sos_point *>(uh + u_pts);
  int *s_point = reinterpret_cast<int *>(uh + u_s_point), *s_orig = reinterpret_cast<int *>(uh + u_s_orig);
  int *t_pair = reinterpret_cast<int *>(uh + u_t_pair), *t_ht = reinterpret_cast<int *>(uh + u_t_ht);
  const float **t_img = reinterpret_cast<const float **>(uh + u_t_img);
  int *p_list = reinterpret_cast<int *>(uh + u_p_list), *p_res_t = reinterpret_cast<int *>(uh + u_p_res_t);
  int2 *p_list2 = reinterpret_cast<int2 *>(uh + u_p_list2), *p_list16 = reinterpret_cast<int2 *>(uh + u_p_list16);
  uint8_t *s_flags = reinterpret_cast<uint8_t *>(uh + u_flags), *s_state = reinterpret_cast<uint8_t *>(uh + u_state);
  float *s_energy = reinterpret_cast<float *>(uh + u_energy), *s_rtz = res_toZeroF ? reinterpret_cast<float *>(uh + u_rtz) : nullptr;
  if (P) memcpy(h_pts_up, pts, sizeof(sos_point) * P);
  memcpy(uh + u_p_begin, p_begin.data(), sizeof(int) * ((size_t)P + 1));
  memcpy(uh + u_ptb, pair_tile_begin.data(), sizeof(int) * ((size_t)nn2 + 1));
  memcpy(uh + u_hcb, host_chunk_begin.data(), sizeof(int) * ((size_t)n + 1));
  memset(uh + u_newstate, SOS_RES_OOB, Rp);
  for (int k = 0; k < nn2; k++) {  // per key: its tiles, and the padding slots behind its residuals
    const int t0 = pair_tile_begin[k], t1 = pair_tile_begin[k + 1];
    const int hI = (k % (n * n)) % n, tI = (k % (n * n)) / n;
    for (int t = t0; t < t1; t++) {
      t_pair[t] = k % (n * n);
      t_img[t] = c->dIt[frame_slot[tI]];
      t_ht[t] = hI | (tI << 16);
    }
    for (int sI = t0 * SOS_TILE + kcount[k + 1]; sI < t1 * SOS_TILE; sI++) {
      s_point[sI] = -1; s_orig[sI] = -1; s_flags[sI] = 0; s_state[sI] = SOS_RES_OOB; s_energy[sI] = 0.f;
      if (s_rtz) memset(s_rtz + (size_t)sI * 8, 0, 8 * sizeof(float));
    }
    kcount[k + 1] = t0 * SOS_TILE;  // from here on: next free slot of key k
  }
  if (Rpad == 0) { s_point[0] = s_orig[0] = -1; s_flags[0] = 0; s_state[0] = SOS_RES_OOB; s_energy[0] = 0.f; }
  std::vector<int> &s_of_orig = ba->h_s_of_orig;
  s_of_orig.resize(R);
  if (P) memset(p_res_t, 0xff, sizeof(int) * (size_t)P * n);
  for (int pt = 0; pt < P; pt++) {
    int2 *l16 = p_list16 + (size_t)pt * 16;
    int k16 = 0;
    for (int o = p_begin[pt]; o < p_begin[pt + 1]; o++) {
      const sos_resid &q = res[o];
      const int sI = kcount[keys[o] + 1]++;
      s_of_orig[o] = sI;
      s_point[sI] = pt;
      s_orig[sI] = o;
      unsigned f = DF_VALID;
      if (q.flags & SOS_RF_ACTIVE) f |= DF_ACTIVE;
      if (q.flags & SOS_RF_LINEARIZED) f |= DF_LINEARIZED;
      if (q.flags & SOS_RF_ISNEW) f |= DF_ISNEW;
      s_flags[sI] = (uint8_t)f;
      s_state[sI] = (uint8_t)q.state_state;
      s_energy[sI] = q.state_energy;
      if (s_rtz) memcpy(s_rtz + (size_t)sI * 8, res_toZeroF + (size_t)o * 8, 8 * sizeof(float));
      p_list[o] = sI;
      p_list2[o] = make_int2(sI, n * q.host + q.target);
      p_res_t[(size_t)pt * n + q.target] = sI;
      if (k16 < 16) l16[k16++] = p_list2[o];
    }
    for (; k16 < 16; k16++) l16[k16] = make_int2(-1, 0);
  }
  {
    int *chunk_pt = reinterpret_cast<int *>(uh + u_chunk);
    size_t w = 0;
    for (int h = 0; h < n; h++)
      for (int q = host_pt_begin[h]; q < host_pt_begin[h + 1]; q += SOS_GC)
        for (int k = 0; k < SOS_GC; k++) chunk_pt[w++] = q + k < host_pt_begin[h + 1] ? q + k : -1;
  }
  ba->h_pts.assign(pts, pts + P);
  ba->h_res.assign(res, res + R);
  const double tw1 = now_s();
  // ---- one upload, one zero fill
  SOS_HIP(hipMemcpyAsync(ud, uh, up.off, hipMemcpyHostToDevice, st));
  k_zero_slab<<<1024, 256, 0, st>>>(reinterpret_cast<float4 *>(zd), zr.off / 16);
  const double tw2 = now_s();
  ENSURE(ba->d_r_geo, Rp); ENSURE(ba->d_r_cw, Rp * 16);
  ENSURE(ba->d_t_pre, 8 * Tp);
  ENSURE(ba->d_o_newstate, Rr); ENSURE(ba->d_o_newenergy, Rr); ENSURE(ba->d_o_newenergywo, Rr);
  ENSURE(ba->d_top_part, Tp * SOS_TOPN + 64 * SOS_TOPN);
  ENSURE(ba->d_gram_part, (size_t)(ba->nchunks ? ba->nchunks : 1) * ba->Dm * ba->Dm);
  const size_t nn = (size_t)n * n;
  ba->off_topA = 0;
  ba->off_topL = ba->off_topA + nn * 91;
  ba->off_D = ba->off_topL + nn * 91;
  ba->off_E = ba->off_D + nn * n * 64;
  ba->off_EB = ba->off_E + nn * 32;
  ba->off_Hcc = ba->off_EB + nn * 8;
  ba->off_bc = ba->off_Hcc + 16;
  ba->off_nres = ba->off_bc + 4;
  ba->acc_floats = ba->off_nres + 2;
  ENSURE(ba->d_acc, ba->acc_floats + 2);  // (+ 2: the sharded resident loop's [sum |idepth|, P] rides behind the blocks)
  // device staging of the per-step inputs (one H2D per step)
  ba->st_pre = 0;
  ba->st_adh = ba->st_pre + nn * 28;
  ba->st_cd = ba->st_adh + nn * 8;
  ba->st_th = ba->st_cd + 4;
  ba->st_cal = ba->st_th + ((size_t)n + 3) / 4 * 4;  // sos_calib (8 floats): the kernels read it from here, not from a kernel argument
  ba->st_xc = ba->st_cal + 8;
  ba->st_xad = ba->st_xc + 4;
  ba->st_floats = ba->st_xad + nn * 8;
  ENSURE(ba->d_stage, ba->st_floats);
  {  // adjoints: [adHost | adTarget] fp64 + [adHostF | adTargetF] fp32 in one pinned / one device block (sos_ba_set_state)
    const size_t o_t = sizeof(double) * 64 * nn, o_hf = 2 * o_t, o_tf = o_hf + sizeof(float) * 64 * nn, tot = o_tf + sizeof(float) * 64 * nn;
    if (tot > ba->adj_cap) {
      if (ba->adj_host) hipHostFree(ba->adj_host);
      if (ba->adj_dev) hipFree(ba->adj_dev);
      ba->adj_host = ba->adj_dev = nullptr;
      ba->adj_cap = 0;
      SOS_HIP(hipHostMalloc((void **)&ba->adj_host, tot + tot / 2, hipHostMallocDefault));
      if (hipMalloc((void **)&ba->adj_dev, tot + tot / 2) != hipSuccess) return SOS_ERR_NOMEM;
      ba->adj_cap = tot + tot / 2;
    }
    slab_view(ba->d_adHost, ba->adj_dev, 0, nn * 64); slab_view(ba->d_adTarget, ba->adj_dev, o_t, nn * 64);
    // the fp32 copies exist once sos_ba_set_state delivered adjoints for this n (their DevBufs are null until then)
    const bool keepF = ba->adj_n == n && ba->d_adHostF.p && ba->d_adTargetF.p;
    ba->adj_n = n;
    ba->d_adHostF.release(); ba->d_adTargetF.release();
    if (keepF) { slab_view(ba->d_adHostF, ba->adj_dev, o_hf, nn * 64); slab_view(ba->d_adTargetF, ba->adj_dev, o_tf, nn * 64); }
  }
  const size_t dim = 4 + 8 * (size_t)n;
  ba->hb_mode_stride = dim * dim + dim;
  ENSURE(ba->d_Hout, 3 * ba->hb_mode_stride + 8); ENSURE(ba->d_scalar, 8); ENSURE(ba->d_perres, Rp);
  ENSURE(ba->d_C, 2 * nn * SOS_TOPC + nn * n * SOS_SCC + nn * SOS_SCE + 64);
  // packed per-step outputs (one D2H per step)
  ba->newest_begin = pair_tile_begin[n * (n - 1)] * SOS_TILE;
  ba->newest_count = (ntilesA - pair_tile_begin[n * (n - 1)]) * SOS_TILE;
  ba->out_esum = 0;
  ba->out_newest = sizeof(double) * (size_t)(ntilesA ? ntilesA : 1);
  ba->out_step = ba->out_newest + sizeof(float) * (size_t)ba->newest_count;
  ba->out_bytes = ba->out_step + sizeof(float) * Pp;
  ENSURE(ba->d_outpack, ba->out_bytes);
  {
    const size_t need_stage = sizeof(float) * ba->st_floats, need_out = ba->out_bytes,
                 need_hb = sizeof(double) * 3 * ba->hb_mode_stride + 16 + 128;  // + completion flags
    const size_t tot = need_stage + need_out + need_hb + 256 + 64;
    if (tot > ba->pin_bytes) {
      if (ba->pin) hipHostFree(ba->pin);
      ba->pin = nullptr;
      SOS_HIP(hipHostMalloc((void **)&ba->pin, tot + tot / 4, hipHostMallocMapped));
      SOS_HIP(hipHostGetDevicePointer((void **)&ba->pin_dev, ba->pin, 0));
      ba->pin_bytes = tot + tot / 4;
      memset(ba->pin, 0, ba->pin_bytes);
    }
    ba->pin_stage = 0;
    ba->pin_out = (need_stage + 63) / 64 * 64;
    ba->pin_hb = ba->pin_out + (need_out + 63) / 64 * 64;
    ba->pin_flags = ba->pin_hb + (sizeof(double) * 3 * ba->hb_mode_stride + 16 + 63) / 64 * 64;
    memset(ba->pin + ba->pin_flags, 0, 128);
  }
  ba->sig_lin_seq = ba->sig_st_seq = 0;
  ba->sig_st_blocks_total = ba->sig_abs_blocks_total = 0;
  ba->acc_wait_seq = 0;
  ba->pend = sos_ba::StepPend();

  BaDev &d = ba->dev;
  memset(&d, 0, sizeof(d));
  d.n = n; d.P = P; d.R = R; d.Rpad = Rpad; d.ntiles = ntiles; d.ntilesA = ntilesA;
  d.w = ba->prm.w; d.h = ba->prm.h;
  d.wM3G = (float)(ba->prm.w - 3); d.hM3G = (float)(ba->prm.h - 3);
  d.huberTH = ba->prm.huberTH; d.outlierTH = ba->prm.outlierTHSumComponent;
  d.modeA = ba->prm.affineOptModeA; d.modeB = ba->prm.affineOptModeB;
  for (int i = 0; i < n; i++) d.img[i] = c->dI[ba->slot[i]][0];
  for (int i = 0; i < n; i++) d.imgT[i] = c->dIt[ba->slot[i]];
  d.tpr = sos_tiles_per_row(c->w);
  d.precalc = reinterpret_cast<const sos_precalc *>(ba->d_stage.p + ba->st_pre);
  d.adHTdelta = ba->d_stage.p + ba->st_adh;
  d.cdelta = ba->d_stage.p + ba->st_cd;
  d.calibp = ba->d_stage.p + ba->st_cal;
  d.tile_esum = reinterpret_cast<double *>(ba->d_outpack.p + ba->out_esum);
  d.o_newest = reinterpret_cast<float *>(ba->d_outpack.p + ba->out_newest);
  d.newest_begin = ba->newest_begin;
  d.newest_count = ba->newest_count;
  d.p_list2 = ba->d_p_list2.p;
  d.p_list16 = ba->d_p_list16.p;
  d.r_geo = ba->d_r_geo.p;
  d.r_cw = ba->d_r_cw.p;
  d.pts = ba->d_pts.p;
  d.s_point = ba->d_s_point.p; d.s_orig = ba->d_s_orig.p;
  d.s_flags = ba->d_s_flags.p; d.s_state = ba->d_s_state.p; d.s_newstate = ba->d_s_newstate.p;
  d.s_energy = ba->d_s_energy.p; d.s_newenergy = ba->d_s_newenergy.p; d.s_newenergywo = ba->d_s_newenergywo.p;
  d.s_ret = ba->d_s_ret.p; d.s_center = ba->d_s_center.p; d.s_rtz = ba->d_s_rtz.p; d.s_pterm = ba->d_s_pterm.p;
  d.t_pair = ba->d_t_pair.p; d.J = ba->d_J.p; d.JpJd = ba->d_JpJd.p;
  d.t_pre = ba->d_t_pre.p; d.t_img = ba->d_t_img.p; d.t_ht = ba->d_t_ht.p;
  d.p_begin = ba->d_p_begin.p; d.p_list = ba->d_p_list.p; d.p_res_t = ba->d_p_res_t.p; d.p_out = ba->d_p_out.p;
  d.o_newstate = ba->d_o_newstate.p; d.o_newenergy = ba->d_o_newenergy.p; d.o_newenergywo = ba->d_o_newenergywo.p;
  d.o_center = ba->d_o_center.p;

  if (Rpad > 0) k_expand_points<<<divup(Rpad, 256), 256, 0, st>>>(d);
  // frozen Jacobians of linearized residuals
  if (lin_J) {
    std::vector<int> sl;
    std::vector<sos_rawjac> jl;
    for (int o = 0; o < R; o++)
      if (res[o].flags & SOS_RF_LINEARIZED) { sl.push_back(s_of_orig[o]); jl.push_back(lin_J[o]); }
    if (!sl.empty()) {
      int rc;
      if ((rc = upload(st, ba->d_tmp_int, sl))) return rc;
      if ((rc = upload(st, ba->d_rawjac, jl))) return rc;
      k_put_jac<<<divup((int)sl.size(), 64), 64, 0, st>>>(d, ba->d_tmp_int.p, ba->d_rawjac.p, (int)sl.size());
      SOS_HIP(hipStreamSynchronize(st));  // sl / jl are pageable locals
    }
  }
  SOS_HIP(hipGetLastError());
  ba->have_window = true;
  ba->have_state = false;
  if (getenv("SOS_TIMING")) fprintf(stderr, "[set_window] host tables %.0f us, upload + zero enqueue %.0f us, ensure / layout %.0f us (slab %.2f MB, zeroed %.2f MB)\n", (tw1 - tw0) * 1e6, (tw2 - tw1) * 1e6, (now_s() - tw2) * 1e6, up.off / 1e6, zr.off / 1e6);
  return comm_setup_window(ba);
}

// stage pointers
static void launch_expand_precalc(sos_ba *ba);
static inline float *stg(sos_ba *ba, size_t off) { return ba->d_stage.p + off; }
static inline float *pstg(sos_ba *ba, size_t off) { return reinterpret_cast<float *>(ba->pin + ba->pin_stage) + off; }

extern "C" int sos_ba_set_state(sos_ba *ba, const sos_calib *calib, const sos_precalc *precalc, const float *adHTdeltaF,
                                const float *cDeltaF, const double *adHost, const double *adTarget,
                                const float *point_idepth_scaled, const float *point_idepth_zero_scaled,
                                const float *point_deltaF) {
  if (ba) ba->devstep = false;  // the host is the source of the states again
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window) return SOS_ERR_STATE;
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  SOS_HIP(hipStreamSynchronize(st));  // the pinned staging areas may still be in flight
  // every source below is copied into pinned memory first: the caller's (pageable) buffers are free when this returns and
  // nothing has to be waited for
  const size_t nn = (size_t)ba->n * ba->n;
  if (calib) {
    ba->calib = *calib; ba->dev.calib = *calib;
    memcpy(pstg(ba, ba->st_cal), calib, sizeof(sos_calib));
  }
  if (precalc) memcpy(pstg(ba, ba->st_pre), precalc, sizeof(sos_precalc) * nn);
  if (adHTdeltaF) memcpy(pstg(ba, ba->st_adh), adHTdeltaF, sizeof(float) * 8 * nn);
  if (cDeltaF) memcpy(pstg(ba, ba->st_cd), cDeltaF, sizeof(float) * 4);
  if (calib && precalc && adHTdeltaF && cDeltaF) {  // the usual case: the stage block [precalc | adHTdelta | cdelta | th | calib] in one copy
    SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_pre), pstg(ba, ba->st_pre), sizeof(float) * ba->st_xc, hipMemcpyHostToDevice, st));
  } else {
    if (calib) SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_cal), pstg(ba, ba->st_cal), sizeof(sos_calib), hipMemcpyHostToDevice, st));
    if (precalc) SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_pre), pstg(ba, ba->st_pre), sizeof(sos_precalc) * nn, hipMemcpyHostToDevice, st));
    if (adHTdeltaF) SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_adh), pstg(ba, ba->st_adh), sizeof(float) * 8 * nn, hipMemcpyHostToDevice, st));
    if (cDeltaF) SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_cd), pstg(ba, ba->st_cd), sizeof(float) * 4, hipMemcpyHostToDevice, st));
  }
  if (precalc) launch_expand_precalc(ba);
  if (adHost || adTarget) {  // fp64 adjoints of the stitch + their fp32 copies (OB/EnergyFunctional.cpp:94-98) for the in-kernel xAd table
    const size_t o_t = sizeof(double) * 64 * nn, o_hf = 2 * o_t, o_tf = o_hf + sizeof(float) * 64 * nn, tot = o_tf + sizeof(float) * 64 * nn;
    double *hH = reinterpret_cast<double *>(ba->adj_host), *hT = reinterpret_cast<double *>(ba->adj_host + o_t);
    float *hHF = reinterpret_cast<float *>(ba->adj_host + o_hf), *hTF = reinterpret_cast<float *>(ba->adj_host + o_tf);
    ba->h_adHostF.resize(64 * nn);
    ba->h_adTargetF.resize(64 * nn);
    if (adHost) {
      memcpy(hH, adHost, o_t);
      for (size_t i = 0; i < 64 * nn; i++) hHF[i] = ba->h_adHostF[i] = (float)adHost[i];
    }
    if (adTarget) {
      memcpy(hT, adTarget, o_t);
      for (size_t i = 0; i < 64 * nn; i++) hTF[i] = ba->h_adTargetF[i] = (float)adTarget[i];
    }
    if (adHost && adTarget) {
      SOS_HIP(hipMemcpyAsync(ba->adj_dev, ba->adj_host, tot, hipMemcpyHostToDevice, st));
    } else if (adHost) {
      SOS_HIP(hipMemcpyAsync(ba->adj_dev, ba->adj_host, o_t, hipMemcpyHostToDevice, st));
      SOS_HIP(hipMemcpyAsync(ba->adj_dev + o_hf, ba->adj_host + o_hf, sizeof(float) * 64 * nn, hipMemcpyHostToDevice, st));
    } else {
      SOS_HIP(hipMemcpyAsync(ba->adj_dev + o_t, ba->adj_host + o_t, o_t, hipMemcpyHostToDevice, st));
      SOS_HIP(hipMemcpyAsync(ba->adj_dev + o_tf, ba->adj_host + o_tf, sizeof(float) * 64 * nn, hipMemcpyHostToDevice, st));
    }
    if (adHost) slab_view(ba->d_adHostF, ba->adj_dev, o_hf, nn * 64);
    if (adTarget) slab_view(ba->d_adTargetF, ba->adj_dev, o_tf, nn * 64);
  }
  if ((point_idepth_scaled || point_idepth_zero_scaled || point_deltaF) && ba->P) {
    sos_point *pp = reinterpret_cast<sos_point *>(ba->up_host + ba->up_pts_off);  // pinned region the snapshot's points were uploaded from
    for (int p = 0; p < ba->P; p++) {
      if (point_idepth_scaled) ba->h_pts[p].idepth_scaled = point_idepth_scaled[p];
      if (point_idepth_zero_scaled) ba->h_pts[p].idepth_zero_scaled = point_idepth_zero_scaled[p];
      if (point_deltaF) ba->h_pts[p].deltaF = point_deltaF[p];
    }
    memcpy(pp, ba->h_pts.data(), sizeof(sos_point) * ba->P);  // the mirror also carries the steps / priors applied since the pack
    SOS_HIP(hipMemcpyAsync(ba->d_pts.p, pp, sizeof(sos_point) * ba->P, hipMemcpyHostToDevice, st));
    if (ba->Rpad > 0) k_refresh_geo<<<divup(ba->Rpad, 256), 256, 0, st>>>(ba->dev);
  }
  SOS_HIP(hipGetLastError());
  ba->have_state = true;
  return SOS_OK;
}

// calibration kernels for the memory-traffic counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are only
// calibrated for wide streaming access): a pure 16 B/lane streaming read and a pure streaming write of known size
__global__ void k_calib_read(const float4 *__restrict__ src, size_t n4, float *__restrict__ sink) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (a.x + a.y + a.z + a.w == 1234.5678f) sink[0] = a.x;  // keeps the loads alive, never true in practice
}
// large-buffer streaming yardstick (bench.py roofline.large_buffer_stream_gbs): 1 GiB read with four independent 16-byte
// loads in flight per lane and iteration -- far beyond the 256 MB of MALL, so this is HBM bandwidth without launch effects
__global__ __launch_bounds__(256) void k_stream_large(const float4 *__restrict__ src, size_t n4, float *__restrict__ sink) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    a.x += v0.x + v1.x + v2.x + v3.x; a.y += v0.y + v1.y + v2.y + v3.y;
    a.z += v0.z + v1.z + v2.z + v3.z; a.w += v0.w + v1.w + v2.w + v3.w;
  }
  for (; i < n4; i += stride) {
    const float4 v = src[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (a.x + a.y + a.z + a.w == 1234.5678f) sink[0] = a.x;
}
__global__ void k_calib_write(float4 *__restrict__ dst, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// per-step inputs: device-mapped pinned host block -> device staging, by a kernel instead of a copy command
__global__ void k_stage_in(float4 *__restrict__ dst, const float4 *__restrict__ src, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) dst[i] = src[i];
}
static int stage_in(sos_ba *ba, size_t nfloats) {
  const int n4 = (int)((nfloats + 3) / 4);
  k_stage_in<<<divup(n4, 256), 256, 0, ba->ctx->stream>>>(reinterpret_cast<float4 *>(ba->d_stage.p),
                                                        reinterpret_cast<const float4 *>(ba->pin_dev + ba->pin_stage), n4);
  if (nfloats > ba->st_pre) launch_expand_precalc(ba);
  return SOS_OK;
}
static void launch_expand_precalc(sos_ba *ba) {  // from the device copy of precalc
  if (ba->ntiles > 0)
    k_expand_precalc<<<divup(8 * ba->ntiles, 256), 256, 0, ba->ctx->stream>>>(ba->ntiles, ba->d_t_pair.p, reinterpret_cast<const float4 *>(stg(ba, ba->st_pre)),
                                                                             ba->d_t_pre.p);
}

// host side of DoneSignal: poll the mapped flag; after 50 ms fall back to a stream synchronisation (a failed launch
// must not hang the caller)
static int wait_flag(sos_ba *ba, size_t flag_off, int seq) {
  int *flag = reinterpret_cast<int *>(ba->pin + flag_off);
  const double t0 = now_s();
  unsigned spins = 0;
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < seq) {
    __builtin_ia32_pause();
    if ((++spins & 4095u) == 0) {
      const double dt = now_s() - t0;
      if (dt > 0.05) {
        SOS_HIP(hipStreamSynchronize(ba->ctx->stream));
        return __atomic_load_n(flag, __ATOMIC_ACQUIRE) >= seq ? SOS_OK : SOS_ERR_HIP;
      }
    }
  }
  return SOS_OK;
}
// Grid of k_linearize2: *nd two-tile blocks followed by one-tile blocks.  A window whose tiles come to an odd number q
// per CU (W12: 1272 tiles on 256 CUs, q = 5) would leave CUs with three two-tile blocks next to CUs with two (6 against
// 4 tiles; the kernel ends with the slowest CU); (q - 1) / 2 two-tile blocks + 1 one-tile block per CU give every CU q.
static int lin_ncu(sos_ba *ba) {
  static int ncu = 0;
  if (!ncu) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ba->ctx->device) != hipSuccess || ncu <= 0) ncu = 256;
  }
  return ncu;
}
static int lin_grid(sos_ba *ba, int *nd) {
  const int ncu = lin_ncu(ba);
  const int T = ba->ntilesA, q = divup(T, ncu);
  int n2 = divup(T, L2_TILES);
  if ((q & 1) && q >= 3) n2 = std::min(T / 2, ncu * (q - 1) / 2);
  *nd = n2;
  return n2 + std::max(0, T - L2_TILES * n2);
}
// returns the sequence number to wait for (0 = this launch does not signal)
static int launch_lin_kernel(sos_ba *ba, const BaDev &dv, int mode, float *fuse_top, bool signal = false, bool deferPublish = false) {
  if (ba->ntilesA <= 0) return 0;
  int nd;
  const int nb = lin_grid(ba, &nd);
  BaDev dv2 = dv;
  dv2.lin_nd = nd;
  // PENDING_FIRST_GPU_RUN: opt-in until the PMC passes have judged it (FETCH_SIZE x 2 <= 28 MB at W12 and no slower, else it leaves)
  static const bool linXcd = getenv("SOS_LIN_XCD") != nullptr && atoi(getenv("SOS_LIN_XCD")) != 0;
  const int ndx = nd | (linXcd ? 0x40000000 : 0);
  const int seq = signal ? ++ba->sig_lin_seq : 0;
  // one round of resident blocks (3 per CU)?  then the scalar-cache form of the first loads
  const bool scalar1 = nb <= 3 * lin_ncu(ba);
#define SOS_LAUNCH_LIN2(F, S1, FT) \
  k_linearize2<F, S1><<<nb, 256 * L2_TILES, 0, ba->ctx->stream>>>(dv2.r_geo, dv2.r_cw, dv2.t_pre, dv2.t_img, dv2.ntilesA, ndx, dv2, stg(ba, ba->st_th), mode, FT)
  if (fuse_top && mode == 1) {
    if (scalar1) SOS_LAUNCH_LIN2(true, true, fuse_top);
    else SOS_LAUNCH_LIN2(true, false, fuse_top);
  } else {
    if (scalar1) SOS_LAUNCH_LIN2(false, true, nullptr);
    else SOS_LAUNCH_LIN2(false, false, nullptr);
  }
#undef SOS_LAUNCH_LIN2
  if (signal && !deferPublish) k_publish<<<1, 1, 0, ba->ctx->stream>>>(reinterpret_cast<int *>(ba->pin_dev + ba->pin_flags), seq);
  return seq;
}
static int launch_linearize(sos_ba *ba, int doApply) {
  launch_lin_kernel(ba, ba->dev, doApply, nullptr);
  return SOS_OK;
}

__global__ void k_fill_f32(float *__restrict__ p, float v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_copy_f32(float *__restrict__ dst, const float *__restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
// per-window setup of the exchange: the ranks agree on the capacity of the energy lists (max over ranks) and on
// whether any of them holds linearised residuals (decides the number of stitch modes)
static int comm_setup_window(sos_ba *ba) {
  if (!ba->comm || !ba->have_window) return SOS_OK;
  hipStream_t st = ba->ctx->stream;
  ba->comm_size = sos_comm_size(ba->comm);
  // [2]: a rank with an empty shard cannot take the absolute-coordinate Schur path; the ranks must pick the same exchange
  int h2[3] = {ba->newest_count, ba->ntiles > ba->ntilesA ? 1 : 0, (ba->ntilesA <= 0 || ba->nchunks <= 0) ? 1 : 0};
  if (ba->d_tmp_int.ensure(3)) return SOS_ERR_NOMEM;
  SOS_HIP(hipMemcpyAsync(ba->d_tmp_int.p, h2, sizeof(h2), hipMemcpyHostToDevice, st));
  int rc = sos_comm_allreduce_max_i32(ba->comm, ba->d_tmp_int.p, 3, st);
  if (rc) return rc;
  SOS_HIP(hipMemcpyAsync(h2, ba->d_tmp_int.p, sizeof(h2), hipMemcpyDeviceToHost, st));
  SOS_HIP(hipStreamSynchronize(st));
  ba->newest_cap = h2[0] > 0 ? h2[0] : 1;
  ba->anyL = h2[1] != 0;
  ba->anyEmpty = h2[2] != 0;
  const size_t tot = (size_t)ba->newest_cap * ba->comm_size;
  if (ba->d_newest_local.ensure(ba->newest_cap) || ba->d_newest_all.ensure(tot)) return SOS_ERR_NOMEM;
  if (tot > ba->pin_newest_floats) {
    if (ba->pin_newest) hipHostFree(ba->pin_newest);
    ba->pin_newest = nullptr;
    SOS_HIP(hipHostMalloc((void **)&ba->pin_newest, sizeof(float) * (tot + tot / 4 + 64), hipHostMallocMapped));
    SOS_HIP(hipHostGetDevicePointer((void **)&ba->pin_newest_dev, ba->pin_newest, 0));
    ba->pin_newest_floats = tot + tot / 4 + 64;
  }
  k_fill_f32<<<divup(ba->newest_cap, 256), 256, 0, st>>>(ba->d_newest_local.p, -1.f, ba->newest_cap);  // padding = "no energy"
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(st));
  return SOS_OK;
}

extern "C" int sos_ba_set_comm(sos_ba *ba, sos_comm *comm) {
  if (!ba) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  SOS_HIP(hipStreamSynchronize(ba->ctx->stream));
  ba->acc_inflight = false, ba->top_valid = false;
  ba->comm = comm;
  ba->comm_size = comm ? sos_comm_size(comm) : 1;
  if (!comm) { ba->anyL = false, ba->anyEmpty = false; return SOS_OK; }
  return comm_setup_window(ba);
}

// the same all-gather for callers outside the fused calls (the final linearizeAll(true) of optimize()): local
// energies (>= 0, at most the window's newest-frame residual count) in, the energies of all ranks out
extern "C" int sos_ba_gather_energies(sos_ba *ba, const float *local, int count, float *all, int *total) {
  if (!ba || !ba->have_window || count < 0 || (count && !local) || !all || !total) return SOS_ERR_ARG;
  if (!ba->comm) {
    if (count) memcpy(all, local, sizeof(float) * count);
    *total = count;
    return SOS_OK;
  }
  if (count > ba->newest_cap) return SOS_ERR_ARG;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  SOS_HIP(hipStreamSynchronize(st));
  const int tot = ba->newest_cap * ba->comm_size;
  for (int i = 0; i < ba->newest_cap; i++) ba->pin_newest[i] = i < count ? local[i] : -1.f;
  k_copy_f32<<<divup(ba->newest_cap, 256), 256, 0, st>>>(ba->d_newest_local.p, ba->pin_newest_dev, ba->newest_cap);
  const int rcc = sos_comm_allgather_f32(ba->comm, ba->d_newest_local.p, ba->d_newest_all.p, ba->newest_cap, st);
  if (rcc) return rcc;
  k_copy_f32<<<divup(tot, 256), 256, 0, st>>>(ba->pin_newest_dev, ba->d_newest_all.p, tot);
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(st));
  int k = 0;
  for (int i = 0; i < tot; i++)
    if (ba->pin_newest[i] >= 0) all[k++] = ba->pin_newest[i];
  *total = k;
  // leave the local list padded for the fused path (it only rewrites its first newest_count entries)
  k_fill_f32<<<divup(ba->newest_cap, 256), 256, 0, st>>>(ba->d_newest_local.p, -1.f, ba->newest_cap);
  SOS_HIP(hipStreamSynchronize(st));
  return SOS_OK;
}

// sum of a host fp64 buffer over all ranks (keyframe-rate exchanges such as the marginalisation prior update):
// staged through the stitch output buffer; a no-op without a communicator
extern "C" int sos_ba_allreduce_f64(sos_ba *ba, double *buf, size_t count) {
  if (!ba || !buf) return SOS_ERR_ARG;
  if (!ba->comm || count == 0) return SOS_OK;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  if (ba->d_ar64.ensure(count)) return SOS_ERR_NOMEM;  // kept on the handle: no allocation to leak on the error exits
  SOS_HIP(hipMemcpyAsync(ba->d_ar64.p, buf, sizeof(double) * count, hipMemcpyHostToDevice, st));
  const int rc = sos_comm_allreduce_sum_f64(ba->comm, ba->d_ar64.p, count, st);
  if (rc) return rc;
  SOS_HIP(hipMemcpyAsync(buf, ba->d_ar64.p, sizeof(double) * count, hipMemcpyDeviceToHost, st));
  SOS_HIP(hipStreamSynchronize(st));
  return SOS_OK;
}

extern "C" int sos_ba_newest_capacity(sos_ba *ba, int *count) {
  if (!ba || !count || !ba->have_window) return SOS_ERR_STATE;
  *count = ba->comm ? ba->newest_cap * ba->comm_size : ba->newest_count;
  return SOS_OK;
}

// d_J is not written by the pipelined iterations (the tiles are reduced on chip): any consumer of the stored
// Jacobians first recomputes them at the unchanged linearisation state (identical values, nothing else is written)
static int ensure_J(sos_ba *ba) {
  if (ba->J_valid) return SOS_OK;
  launch_lin_kernel(ba, ba->dev, 2, nullptr);
  ba->J_valid = true;
  return SOS_OK;
}

extern "C" int sos_ba_linearize(sos_ba *ba, const float *frameEnergyTH, double *energySum, uint8_t *newState,
                                float *newEnergy, float *newEnergyWithOutlier, float *centerProjectedTo) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || !ba->have_state || !frameEnergyTH) return SOS_ERR_STATE;
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_th), frameEnergyTH, sizeof(float) * ba->n, hipMemcpyHostToDevice, st));
  {  // the new Jacobians go to the PointFrameResidual::J side; EFResidual::J changes at sos_ba_apply_res only
    const size_t Rp = (size_t)ba->Rpad;
    if (ba->d_Jnew.ensure((size_t)(ba->ntiles ? ba->ntiles : 1) * SOS_TILE_FLOATS) || ba->d_JpJd_new.ensure(Rp * 8 + 8) ||
        ba->d_pterm_new.ensure(Rp * 8 + 8))
      return SOS_ERR_NOMEM;
    BaDev dv = ba->dev;
    dv.J = ba->d_Jnew.p;
    dv.JpJd = ba->d_JpJd_new.p;
    dv.s_pterm = ba->d_pterm_new.p;
    launch_lin_kernel(ba, dv, 0, nullptr);
    ba->pending_new = true;
  }
  if (energySum) {
    k_sum_ret<<<1, 1024, 0, st>>>(ba->d_s_ret.p, ba->ntilesA * SOS_TILE, ba->d_scalar.p);
    SOS_HIP(hipMemcpyAsync(energySum, ba->d_scalar.p, sizeof(double), hipMemcpyDeviceToHost, st));
  }
  const size_t R = ba->R;
  if (newState && R) SOS_HIP(hipMemcpyAsync(newState, ba->d_o_newstate.p, R, hipMemcpyDeviceToHost, st));
  if (newEnergy && R) SOS_HIP(hipMemcpyAsync(newEnergy, ba->d_o_newenergy.p, sizeof(float) * R, hipMemcpyDeviceToHost, st));
  if (newEnergyWithOutlier && R)
    SOS_HIP(hipMemcpyAsync(newEnergyWithOutlier, ba->d_o_newenergywo.p, sizeof(float) * R, hipMemcpyDeviceToHost, st));
  if (centerProjectedTo && R) {  // the fused iterations keep the centres in tile order only
    k_unsort_center<<<divup(ba->Rpad, 256), 256, 0, st>>>(ba->dev);
    SOS_HIP(hipMemcpyAsync(centerProjectedTo, ba->d_o_center.p, sizeof(float) * 3 * R, hipMemcpyDeviceToHost, st));
  }
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(st));
  // linearized residuals are not part of activeResiduals: report their stored state
  if (newState || newEnergyWithOutlier)
    for (size_t o = 0; o < R; o++)
      if (ba->h_res[o].flags & SOS_RF_LINEARIZED) {
        if (newState) newState[o] = (uint8_t)ba->h_res[o].state_state;
        if (newEnergyWithOutlier) newEnergyWithOutlier[o] = -1.f;
      }
  return SOS_OK;
}

extern "C" int sos_ba_apply_res(sos_ba *ba) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  const int nthr = ba->ntilesA * SOS_TILE;
  if (nthr > 0 && ba->pending_new) {
    k_commit_new<<<ba->ntilesA, 256, 0, ba->ctx->stream>>>(ba->dev, ba->d_Jnew.p, ba->d_JpJd_new.p, ba->d_pterm_new.p);
    ba->J_valid = true;  // every residual that is active after this apply has just received its Jacobian
  }
  ba->pending_new = false;
  if (nthr > 0) k_apply_res<<<divup(nthr, 256), 256, 0, ba->ctx->stream>>>(ba->dev);
  SOS_HIP(hipGetLastError());
  return SOS_OK;
}

extern "C" int sos_ba_reset_oob(sos_ba *ba) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  const int nthr = ba->ntilesA * SOS_TILE;
  if (nthr > 0) k_reset_oob<<<divup(nthr, 256), 256, 0, ba->ctx->stream>>>(ba->dev);
  SOS_HIP(hipGetLastError());
  return SOS_OK;
}

// residuals the host dropped from its graph since the snapshot was made: dead on the device from here on (no flags, zero
// JpJdF / point terms, state OOB), so the snapshot can serve the rest of the keyframe without a second pack
__global__ void k_kill_residuals(BaDev d, const int *__restrict__ slist, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int s = slist[i];
  d.s_flags[s] = 0;
  d.s_state[s] = SOS_RES_OOB;
  d.s_newstate[s] = SOS_RES_OOB;
  float4 *jp = reinterpret_cast<float4 *>(d.JpJd + 8 * (size_t)s), *pt = reinterpret_cast<float4 *>(d.s_pterm + 8 * (size_t)s);
  jp[0] = jp[1] = pt[0] = pt[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}
extern "C" int sos_ba_kill_residuals(sos_ba *ba, const int32_t *residIdx, int count) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;
  if (!ba || !ba->have_window || (count && !residIdx)) return SOS_ERR_STATE;
  if (count <= 0) return SOS_OK;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  std::vector<int> sl(count);
  for (int k = 0; k < count; k++) {
    if (residIdx[k] < 0 || residIdx[k] >= ba->R) return SOS_ERR_ARG;
    sl[k] = ba->h_s_of_orig[residIdx[k]];
    ba->h_res[residIdx[k]].flags = 0;
    ba->h_res[residIdx[k]].state_state = SOS_RES_OOB;
  }
  int rc = upload(st, ba->d_tmp_int, sl);
  if (rc) return rc;
  k_kill_residuals<<<divup(count, 64), 64, 0, st>>>(ba->dev, ba->d_tmp_int.p, count);
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(st));  // sl is pageable
  return SOS_OK;
}

extern "C" int sos_ba_fix_linearization(sos_ba *ba, const int32_t *residIdx, int count) {
  if (ba && ba->have_window && ba->have_state) ensure_J(ba);
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || !ba->have_state || (count && !residIdx)) return SOS_ERR_STATE;
  if (count <= 0) return SOS_OK;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  std::vector<int> sl(count);
  for (int k = 0; k < count; k++) {
    if (residIdx[k] < 0 || residIdx[k] >= ba->R) return SOS_ERR_ARG;
    sl[k] = ba->h_s_of_orig[residIdx[k]];
    ba->h_res[residIdx[k]].flags |= SOS_RF_LINEARIZED;
  }
  int rc = upload(st, ba->d_tmp_int, sl);
  if (rc) return rc;
  k_fix_lin<<<divup(count, 64), 64, 0, st>>>(ba->dev, ba->d_tmp_int.p, count);
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipStreamSynchronize(st));
  return SOS_OK;
}

// ---- accumulation pipeline ---------------------------------------------------------------------
static int launch_top(sos_ba *ba) {
  hipStream_t st = ba->ctx->stream;
  ensure_J(ba);
  const int nA = ba->ntilesA, nL = ba->ntiles - ba->ntilesA;
  if (nA > 0)
    k_top_accumulate<false><<<divup(nA, 8), 256, 0, st>>>(ba->dev, 0, nA, 0, nullptr, nullptr, ba->d_top_part.p, nullptr);
  if (nL > 0)
    k_top_accumulate<false><<<divup(nL, 8), 256, 0, st>>>(ba->dev, nA, nL, 1, nullptr, nullptr,
                                                        ba->d_top_part.p + (size_t)nA * SOS_TOPN, nullptr);
  return SOS_OK;
}
static size_t gram_lds(const sos_ba *ba) { return sizeof(float) * ((size_t)SOS_GC * ba->ld + SOS_GC); }
static int launch_sc(sos_ba *ba, int shiftPriorToZero) {
  hipStream_t st = ba->ctx->stream;
  if (ba->P > 0) k_point_prep<<<divup(ba->P, 64), 64, 0, st>>>(ba->dev, shiftPriorToZero, nullptr, ba->P);
  if (ba->nchunks > 0) k_sc_gram<<<ba->nchunks, 256, gram_lds(ba), st>>>(ba->dev, ba->d_chunk_pt.p, ba->Dm, ba->ld, ba->d_gram_part.p);
  return SOS_OK;
}
static void fill_reduce(const sos_ba *ba, ReduceArgs &a, const float *top_part, const int *pair_tile_begin, const float *gram_part,
                        const int *host_chunk_begin, int nchunks, int nmodes, float *acc) {
  const int n = ba->n;
  a.top_part = top_part; a.pair_tile_begin = pair_tile_begin; a.gram_part = gram_part; a.host_chunk_begin = host_chunk_begin;
  a.n = n; a.Dm = ba->Dm; a.nchunks = nchunks; a.nmodes = nmodes;
  a.b_top = nmodes * n * n;
  a.b_sc = divup(n * 8 * n * (8 * n + 5), 128);
  a.b_tail = 20 + nmodes;
  a.accTop = acc + ba->off_topA;
  a.accD = acc + ba->off_D; a.accE = acc + ba->off_E; a.accEB = acc + ba->off_EB;
  a.accHcc = acc + ba->off_Hcc; a.accbc = acc + ba->off_bc; a.nres = acc + ba->off_nres;
}
static int launch_reduce(sos_ba *ba) {
  ReduceArgs a;
  fill_reduce(ba, a, ba->d_top_part.p, ba->d_pair_tile_begin.p, ba->d_gram_part.p, ba->d_host_chunk_begin.p, ba->nchunks, 2,
              ba->d_acc.p);
  k_reduce_all<<<a.b_top + a.b_sc + a.b_tail, 128, 0, ba->ctx->stream>>>(a);
  return SOS_OK;
}
// stitch kernels: d_Hout = [H_A | b_A | H_L | b_L | H_sc | b_sc]
static int launch_stitch(sos_ba *ba, const float *acc, int nmodes, double *Hout = nullptr, bool toDevice = false) {
  hipStream_t st = ba->ctx->stream;
  const int n = ba->n;
  const size_t nn = (size_t)n * n;
  double *Ctop = ba->d_C.p, *Csc = Ctop + 2 * nn * SOS_TOPC, *Ce = Csc + nn * n * SOS_SCC, *Ccc = Ce + nn * SOS_SCE;
  double *H = Hout ? Hout : ba->d_Hout.p;
  StitchArgs a;
  a.n = n; a.nmodes = nmodes;
  a.acc_top = acc + ba->off_topA; a.accD = acc + ba->off_D; a.accE = acc + ba->off_E; a.accEB = acc + ba->off_EB;
  a.accHcc = acc + ba->off_Hcc; a.accbc = acc + ba->off_bc; a.nres = acc + ba->off_nres;
  a.adHost = ba->d_adHost.p; a.adTarget = ba->d_adTarget.p;
  a.Ctop = Ctop; a.Ccc = Ccc; a.Csc = Csc; a.Ce = Ce; a.H = H;
  a.nres_out = Hout ? reinterpret_cast<float *>(Hout + 3 * ba->hb_mode_stride) : nullptr;
  a.mode_stride = ba->hb_mode_stride;
  a.upperOnly = Hout ? 1 : 0;
  a.sg = {nullptr, nullptr, 0, 0};
  const int nb2 = (n * (n + 1) / 2 + 1) * nmodes + n * n + 1;
  // (SOS_STITCH_SIGNAL_IN_KERNEL: only this chain's last kernel signals by itself -- ~160 blocks with one system fence each against the
  // 4 us one-thread k_publish behind it; the linearisation keeps its chained publish, whose in-kernel form was measured slower)
  static const bool inKernel = getenv("SOS_STITCH_SIGNAL_IN_KERNEL") != nullptr;
  if (toDevice) {  // the device-resident loop: upper triangles + counts into d_Hout, nobody polls
    a.H = ba->d_Hout.p;
    a.nres_out = reinterpret_cast<float *>(ba->d_Hout.p + 3 * ba->hb_mode_stride);
    a.upperOnly = 1;
    k_stitch_stage1<<<(n * n + 20) * nmodes + n * n * n, 64, sizeof(double) * 128 * (size_t)n, st>>>(a);
    k_stitch_stage2<<<nb2, 64, 0, st>>>(a);
    return SOS_OK;
  }
  if (Hout) {  // the fused path: the host polls for the end of stage 2
    ++ba->sig_st_seq;
    if (inKernel) {
      ba->sig_st_blocks_total += nb2;
      a.sg.ctr = ba->d_sigctr.p + 16;
      a.sg.flag = reinterpret_cast<int *>(ba->pin_dev + ba->pin_flags + 64);
      a.sg.target = ba->sig_st_blocks_total;
      a.sg.seq = ba->sig_st_seq;
    }
  }
  k_stitch_stage1<<<(n * n + 20) * nmodes + n * n * n, 64, sizeof(double) * 128 * (size_t)n, st>>>(a);
  k_stitch_stage2<<<nb2, 64, 0, st>>>(a);
  if (Hout && !inKernel) k_publish<<<1, 1, 0, st>>>(reinterpret_cast<int *>(ba->pin_dev + ba->pin_flags + 64), ba->sig_st_seq);
  return SOS_OK;
}

extern "C" int sos_ba_accumulate_local(sos_ba *ba) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || !ba->have_state) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  launch_top(ba);
  launch_sc(ba, 1);
  launch_reduce(ba);
  SOS_HIP(hipGetLastError());
  return SOS_OK;
}

extern "C" int sos_ba_acc_buffer(sos_ba *ba, float **dev_ptr, size_t *nfloats) {
  if (!ba || !ba->have_window) return SOS_ERR_STATE;
  if (dev_ptr) *dev_ptr = ba->d_acc.p;
  if (nfloats) *nfloats = ba->acc_floats;
  return SOS_OK;
}

// D2H of the stitched system through the pinned buffer (one copy + the two counts)
static int fetch_hb(sos_ba *ba, const float *acc, double *H_A, double *b_A, double *H_L, double *b_L, double *H_sc, double *b_sc,
                    int *resInA, int *resInL, bool haveL) {
  hipStream_t st = ba->ctx->stream;
  const size_t dim = 4 + 8 * (size_t)ba->n, ms = ba->hb_mode_stride;
  double *ph = reinterpret_cast<double *>(ba->pin + ba->pin_hb);
  float *pn = reinterpret_cast<float *>(ph + 3 * ms);
  SOS_HIP(hipMemcpyAsync(ph, ba->d_Hout.p, sizeof(double) * 3 * ms, hipMemcpyDeviceToHost, st));
  SOS_HIP(hipMemcpyAsync(pn, acc + ba->off_nres, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
  SOS_HIP(hipStreamSynchronize(st));
  if (H_A) memcpy(H_A, ph, sizeof(double) * dim * dim);
  if (b_A) memcpy(b_A, ph + dim * dim, sizeof(double) * dim);
  if (haveL) {
    if (H_L) memcpy(H_L, ph + ms, sizeof(double) * dim * dim);
    if (b_L) memcpy(b_L, ph + ms + dim * dim, sizeof(double) * dim);
  } else {
    if (H_L) memset(H_L, 0, sizeof(double) * dim * dim);
    if (b_L) memset(b_L, 0, sizeof(double) * dim);
  }
  if (H_sc) memcpy(H_sc, ph + 2 * ms, sizeof(double) * dim * dim);
  if (b_sc) memcpy(b_sc, ph + 2 * ms + dim * dim, sizeof(double) * dim);
  if (resInA) *resInA = (int)pn[0];
  if (resInL) *resInL = haveL ? (int)pn[1] : 0;
  return SOS_OK;
}

extern "C" int sos_ba_stitch(sos_ba *ba, double *H_A, double *b_A, double *H_L, double *b_L, double *H_sc, double *b_sc,
                             int *resInA, int *resInL) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || !ba->have_state) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  launch_stitch(ba, ba->d_acc.p, 2);
  SOS_HIP(hipGetLastError());
  return fetch_hb(ba, ba->d_acc.p, H_A, b_A, H_L, b_L, H_sc, b_sc, resInA, resInL, true);
}

extern "C" int sos_ba_accumulate(sos_ba *ba, double *H_A, double *b_A, double *H_L, double *b_L, double *H_sc,
                                 double *b_sc, int *resInA, int *resInL) {
  int rc = sos_ba_accumulate_local(ba);
  if (rc) return rc;
  return sos_ba_stitch(ba, H_A, b_A, H_L, b_L, H_sc, b_sc, resInA, resInL);
}

// accumulate + stitch of the whole window with the stage-2 stitch kernels writing H/b (and the residual counts)
// straight into the device-mapped pinned block: no copy command between the last kernel and the host
// the absolute-coordinate Schur path (k_sc_gram_abs): windows without linearised residuals on any rank
static bool abs_path_ok(const sos_ba *ba) {
  // PENDING_FIRST_GPU_RUN: opt-in (SOS_ABS_SC=1) until the GPU suite has run on it (executed under tests/emu in round 4: at par with the reference's fp32
  // arithmetic from T6 up, 3.2x further from the fp64 step at T4)
  static const bool off = getenv("SOS_ABS_SC") == nullptr;
  return !off && ba->ntiles == ba->ntilesA && ba->ntilesA > 0 && ba->nchunks > 0 && !(ba->comm && (ba->anyL || ba->anyEmpty)) && ba->d_adHostF.p &&
         ba->d_adTargetF.p;
}
static int enqueue_gn_accumulate_abs(sos_ba *ba, bool topDone, int *pubFlag, int pubSeq) {
  hipStream_t st = ba->ctx->stream;
  const int n = ba->n;
  const size_t nn = (size_t)n * n, ms = ba->hb_mode_stride;
  const size_t lds = sizeof(float) * gram_abs_lds_floats(n, ba->ld);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_sc_gram_abs), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_done = true;
  }
  if (!topDone) {  // the tile sums from the stored Jacobians first
    if (pubFlag) k_publish<<<1, 1, 0, st>>>(pubFlag, pubSeq);
    ensure_J(ba);
    k_top_accumulate<false><<<divup(ba->ntilesA, 8), 256, 0, st>>>(ba->dev, 0, ba->ntilesA, 0, nullptr, nullptr, ba->d_top_part.p, nullptr);
    pubFlag = nullptr;
  }
  k_sc_gram_abs<<<ba->nchunks, 256, lds, st>>>(ba->dev, ba->d_chunk_pt.p, ba->Dm, ba->ld, ba->d_gram_part.p, ba->d_adHostF.p, ba->d_adTargetF.p,
                                               pubFlag, pubSeq);
  AbsStitchArgs a;
  a.n = n; a.Dm = ba->Dm; a.nchunks = ba->nchunks;
  a.top_part = ba->d_top_part.p; a.pair_tile_begin = ba->d_pair_tile_begin.p; a.gram_part = ba->d_gram_part.p;
  a.adHost = ba->d_adHost.p; a.adTarget = ba->d_adTarget.p;
  a.Ctop = ba->d_C.p;
  a.Pcc = ba->d_C.p + 2 * nn * SOS_TOPC;  // (the n^3 blocks of the relative-coordinate stitch are not used on this path)
  double *pinH = reinterpret_cast<double *>(ba->pin_dev + ba->pin_hb);
  a.H = ba->comm ? ba->d_Hout.p : pinH;
  a.mode_stride = ms;
  const int T = ba->Dm >> 4;
  k_abs_reduce_stitch1<<<(int)nn + T * (T + 1) / 2 * 16, 128, 0, st>>>(a);
  // completion: a k_publish behind the last kernel, or (SOS_STITCH_SIGNAL_IN_KERNEL, A/B knob) the last block of k_abs_stitch2 itself --
  // 80 blocks, one system-scope fence each; H_sc was written by the previous launch and is visible at its end
  static const bool inKernel = getenv("SOS_STITCH_SIGNAL_IN_KERNEL") != nullptr;
  const int nb2 = n * (n + 1) / 2 + 1;
  a.sg = {nullptr, nullptr, 0, 0};
  ++ba->sig_st_seq;
  if (inKernel && !ba->comm) {
    ba->sig_abs_blocks_total += nb2;
    a.sg.ctr = ba->d_sigctr.p + 24;
    a.sg.flag = reinterpret_cast<int *>(ba->pin_dev + ba->pin_flags + 64);
    a.sg.target = ba->sig_abs_blocks_total;
    a.sg.seq = ba->sig_st_seq;
  }
  k_abs_stitch2<<<nb2, 64, 0, st>>>(a);
  if (ba->comm) {  // THE exchange step of the path, on the stitched fp64 system (the stitch is linear): [H_A b_A | H_sc b_sc | count]
    const int rcc = sos_comm_allreduce_sum_f64(ba->comm, ba->d_Hout.p, 2 * ms + 1, st);
    if (rcc) return rcc;
    k_copy_f64<<<divup((int)(2 * ms + 1), 256), 256, 0, st>>>(pinH, ba->d_Hout.p, 2 * ms + 1);
  }
  if (!a.sg.ctr) k_publish<<<1, 1, 0, st>>>(reinterpret_cast<int *>(ba->pin_dev + ba->pin_flags + 64), ba->sig_st_seq);
  ba->acc_inflight_haveL = false;
  ba->acc_inflight_abs = true;
  ba->acc_wait_seq = ba->sig_st_seq;
  return SOS_OK;
}
static int enqueue_gn_accumulate(sos_ba *ba, bool topDone = false, int *pubFlag = nullptr, int pubSeq = 0) {
  if (abs_path_ok(ba)) return enqueue_gn_accumulate_abs(ba, topDone, pubFlag, pubSeq);
  ba->acc_inflight_abs = false;
  const bool haveL = ba->ntiles > ba->ntilesA || (ba->comm && ba->anyL);
  if (topDone && ba->ntiles == ba->ntilesA) {  // the tile sums came out of the linearisation itself: only the Schur half is left
    if (ba->nchunks > 0)
      k_sc_gram_prep<<<ba->nchunks, 256, gram_lds(ba), ba->ctx->stream>>>(ba->dev, ba->d_chunk_pt.p, ba->Dm, ba->ld, ba->d_gram_part.p,
                                                                             pubFlag, pubSeq);
    else if (pubFlag) k_publish<<<1, 1, 0, ba->ctx->stream>>>(pubFlag, pubSeq);
  } else if (ba->ntiles == ba->ntilesA && ba->ntilesA > 0 && ba->nchunks > 0) {  // top and Schur halves are independent: one launch
    if (pubFlag) k_publish<<<1, 1, 0, ba->ctx->stream>>>(pubFlag, pubSeq);
    ensure_J(ba);
    const int nTop = divup(ba->ntilesA, 8);
    k_accumulate_fused<<<nTop + ba->nchunks, 256, gram_lds(ba), ba->ctx->stream>>>(ba->dev, nTop, ba->d_top_part.p, ba->d_chunk_pt.p,
                                                                                ba->Dm, ba->ld, ba->d_gram_part.p);
  } else {  // linearised residuals: their point terms come from the mode-1 top pass
    if (pubFlag) k_publish<<<1, 1, 0, ba->ctx->stream>>>(pubFlag, pubSeq);
    launch_top(ba);
    launch_sc(ba, 1);
  }
  launch_reduce(ba);
  if (ba->comm) {  // THE exchange step of the path: packed fp32 blocks summed over all ranks, on this stream
    const int rcc = sos_comm_allreduce_sum_f32(ba->comm, ba->d_acc.p, ba->acc_floats, ba->ctx->stream);
    if (rcc) return rcc;
  }
  launch_stitch(ba, ba->d_acc.p, haveL ? 2 : 1, reinterpret_cast<double *>(ba->pin_dev + ba->pin_hb));
  ba->acc_inflight_haveL = haveL;
  ba->acc_wait_seq = ba->sig_st_seq;
  return SOS_OK;
}

// The enqueue half of sos_ba_gn_accumulate on its own: nothing is waited for.  A caller with host work that does not depend on H / b
// (the IMU factors of the visual-inertial solve) calls this first, so that the work overlaps the accumulation also when no
// sos_ba_gn_step prefetched it.  PENDING_FIRST_GPU_RUN (round 3, written without GPU access; executed under tests/emu in round 4; used with SOS_IMU_OVERLAP=1).
extern "C" int sos_ba_gn_accumulate_begin(sos_ba *ba) {
  if (!ba || !ba->have_window || !ba->have_state) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  if (!ba->acc_inflight) {
    enqueue_gn_accumulate(ba, ba->top_valid);
    SOS_HIP(hipGetLastError());
    ba->acc_inflight = true;
  }
  return SOS_OK;
}

// Fused per-iteration call #1: accumulate + stitch, H_top = H_A + H_L, b_top = b_A + b_L (no priors).
extern "C" int sos_ba_gn_accumulate(sos_ba *ba, double *H_top, double *b_top, double *H_sc, double *b_sc, int *resInA,
                                    int *resInL) {
  if (!ba || !ba->have_window || !ba->have_state || !H_top || !b_top || !H_sc || !b_sc) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  if (!ba->acc_inflight) {  // otherwise the previous sos_ba_gn_step already enqueued it (sos_ba_set_prefetch)
    enqueue_gn_accumulate(ba, ba->top_valid);
    SOS_HIP(hipGetLastError());
  }
  ba->acc_inflight = false, ba->top_valid = false;
  const bool haveL = ba->acc_inflight_haveL;
  const double ta = now_s();
  {
    const int rcw = wait_flag(ba, ba->pin_flags + 64, ba->acc_wait_seq);
    if (rcw) return rcw;
  }
  ba->tm[6] += now_s() - ta;
  const size_t dim = 4 + 8 * (size_t)ba->n, ms = ba->hb_mode_stride;
  const double *ph = reinterpret_cast<const double *>(ba->pin + ba->pin_hb);
  const float *pn = reinterpret_cast<const float *>(ph + 3 * ms);
  if (ba->acc_inflight_abs) {  // [H_A b_A | H_sc b_sc | count] from the absolute-coordinate path
    for (size_t i = 0; i < dim; i++) {
      const size_t o = i * dim + i, len = dim - i;
      memcpy(H_top + o, ph + o, sizeof(double) * len);
      memcpy(H_sc + o, ph + ms + o, sizeof(double) * len);
    }
    memcpy(b_top, ph + dim * dim, sizeof(double) * dim);
    memcpy(b_sc, ph + ms + dim * dim, sizeof(double) * dim);
    if (resInA) *resInA = (int)ph[2 * ms];
    if (resInL) *resInL = 0;
    return SOS_OK;
  }
  // only the upper triangle (col >= row) is produced and copied: it is all the solve reads
  for (size_t i = 0; i < dim; i++) {
    const size_t o = i * dim + i, len = dim - i;
    if (haveL) {
      for (size_t j = 0; j < len; j++) H_top[o + j] = ph[ms + o + j] + ph[o + j];  // HL_top + HA_top
    } else {
      memcpy(H_top + o, ph + o, sizeof(double) * len);
    }
    memcpy(H_sc + o, ph + 2 * ms + o, sizeof(double) * len);
  }
  if (haveL) {
    for (size_t i = 0; i < dim; i++) b_top[i] = ph[ms + dim * dim + i] + ph[dim * dim + i];
  } else {
    memcpy(b_top, ph + dim * dim, sizeof(double) * dim);
  }
  memcpy(b_sc, ph + 2 * ms + dim * dim, sizeof(double) * dim);
  if (resInA) *resInA = (int)pn[0];
  if (resInL) *resInL = haveL ? (int)pn[1] : 0;
  return SOS_OK;
}

extern "C" int sos_ba_get_point_hessian(sos_ba *ba, float *idepth_hessian, float *HdiF, float *bdSumF) {
  if (!ba || !ba->have_window) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  std::vector<float> po((size_t)ba->P * 16);
  if (ba->P) SOS_HIP(hipMemcpyAsync(po.data(), ba->d_p_out.p, sizeof(float) * po.size(), hipMemcpyDeviceToHost, ba->ctx->stream));
  SOS_HIP(hipStreamSynchronize(ba->ctx->stream));
  for (int p = 0; p < ba->P; p++) {
    if (idepth_hessian) idepth_hessian[p] = po[(size_t)p * 16 + PO_IDH];
    if (HdiF) HdiF[p] = po[(size_t)p * 16 + PO_HDI];
    if (bdSumF) bdSumF[p] = po[(size_t)p * 16 + PO_BDSUM];
  }
  return SOS_OK;
}

// xc / xAd (OB/EnergyFunctional.cpp:499-516) into the pinned stage, fp32, summed left to right
static void fill_x(sos_ba *ba, const double *x) {
  const int n = ba->n, dim = 4 + 8 * n;
  float xF[4 + 8 * SOS_MAX_FRAMES];
  for (int i = 0; i < dim; i++) xF[i] = (float)x[i];
  float *xc = pstg(ba, ba->st_xc), *xAd = pstg(ba, ba->st_xad);
  for (int i = 0; i < 4; i++) xc[i] = xF[i];
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) {
      const float *AH = &ba->h_adHostF[64 * (size_t)(h + n * t)], *AT = &ba->h_adTargetF[64 * (size_t)(h + n * t)];
      float *o = xAd + 8 * (size_t)(n * h + t);
      for (int j = 0; j < 8; j++) {
        float s1 = 0, s2 = 0;
        for (int i = 0; i < 8; i++) {
          s1 += xF[4 + 8 * h + i] * AH[8 * i + j];
          s2 += xF[4 + 8 * t + i] * AT[8 * i + j];
        }
        o[j] = s1 + s2;
      }
    }
}

extern "C" int sos_ba_resubstitute(sos_ba *ba, const double *x, float *pointStep) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || !ba->have_state || !x) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  const size_t nn = (size_t)ba->n * ba->n;
  SOS_HIP(hipStreamSynchronize(st));
  fill_x(ba, x);
  SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_xc), pstg(ba, ba->st_xc), sizeof(float) * (4 + 8 * nn), hipMemcpyHostToDevice, st));
  float *dstep = reinterpret_cast<float *>(ba->d_outpack.p + ba->out_step);
  if (ba->P > 0) k_resubstitute<<<divup(ba->P, 64), 64, 0, st>>>(ba->dev, stg(ba, ba->st_xc), stg(ba, ba->st_xad), dstep, 0, 1.f);
  SOS_HIP(hipGetLastError());
  if (pointStep && ba->P) SOS_HIP(hipMemcpyAsync(pointStep, dstep, sizeof(float) * ba->P, hipMemcpyDeviceToHost, st));
  SOS_HIP(hipStreamSynchronize(st));
  return SOS_OK;
}

// Fused per-iteration call #2: back-substitution with x, point step applied on the device
// (doStepFromBackup), new per-step state, linearizeAll(false) (+ applyRes when applyRes != 0).
// One H2D of the packed inputs, one D2H of the packed outputs, one synchronisation.
// The back-substitution of a step depends on x alone: enqueued as soon as the solve is done, it runs while the host
// still derives the new poses and the n^2 precalc records; the following sos_ba_gn_step(x = NULL) then only stages
// those in (one launch) ahead of the linearisation.
extern "C" int sos_ba_gn_resub(sos_ba *ba, const double *x, float stepfacD) {
  if (!ba || !x || !ba->have_window || !ba->have_state) return SOS_ERR_STATE;
  if (!(ba->P > 0 && ba->d_adHostF.p && ba->d_adTargetF.p)) return SOS_ERR_STATE;  // caller passes x to sos_ba_gn_step instead
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  ba->acc_inflight = false, ba->top_valid = false;
  const size_t nn = (size_t)ba->n * ba->n;
  XArg xa;
  const int dim = 4 + 8 * ba->n;
  for (int i = 0; i < dim; i++) xa.v[i] = (float)x[i];
  float *dstep = reinterpret_cast<float *>(ba->pin_dev + ba->pin_out + ba->out_step);
  const int nPB = divup(ba->P, SOS_RSB);
  k_resub_fused<<<nPB, SOS_RSB, sizeof(float) * (8 * nn + 4 + 8 * (size_t)ba->n), c->stream>>>(
      ba->dev, xa, ba->d_adHostF.p, ba->d_adTargetF.p, dstep, stepfacD, nPB, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr);
  SOS_HIP(hipGetLastError());
  ba->resub_pending = true;
  return SOS_OK;
}

// Back half of a fused iteration: linearizeAll(false) + applyRes on the state the stream has reached, the next
// iteration's accumulate chain enqueued behind it (sos_ba_set_prefetch), results through the mapped block.
static int lin_apply_enqueue(sos_ba *ba, BaDev &dv, int applyRes, bool havePointStep, float stepfacD, double t0, double t1) {
  sos_ctx *c = ba->ctx;
  hipStream_t st = c->stream;
  // nobody reads the original-order copies of the per-residual results in a fused iteration (they are scattered 1..12 byte
  // stores, one partial line each): only the two-step sos_ba_linearize delivers them
  dv.o_newstate = nullptr; dv.o_newenergy = nullptr; dv.o_newenergywo = nullptr; dv.o_center = nullptr;
  // pipelined iterations of a window without linearised residuals reduce the tiles on chip (no J traffic at all)
  const bool fuseTop = (ba->prefetch || ba->fuse_only) && applyRes && ba->ntiles == ba->ntilesA;
  // when the next accumulate is enqueued right behind, its first kernel publishes this step's completion
  // (with a communicator the all-gathered energies are copied out behind the linearisation; the chained publish then follows them)
  const bool chainPublish = ba->prefetch && applyRes;
  int waitSeq = launch_lin_kernel(ba, dv, applyRes ? 1 : 0, fuseTop ? ba->d_top_part.p : nullptr, ba->comm == nullptr, chainPublish);
  if (ba->comm) {  // energies of the newest frame of ALL ranks (same frameEnergyTH everywhere), then tell the host
    const int tot = ba->newest_cap * ba->comm_size;
    const int rcc = sos_comm_allgather_f32(ba->comm, ba->d_newest_local.p, ba->d_newest_all.p, ba->newest_cap, st);
    if (rcc) return rcc;
    k_copy_f32<<<divup(tot, 256), 256, 0, st>>>(ba->pin_newest_dev, ba->d_newest_all.p, tot);
    waitSeq = ++ba->sig_lin_seq;
    if (!chainPublish) k_publish<<<1, 1, 0, st>>>(reinterpret_cast<int *>(ba->pin_dev + ba->pin_flags), waitSeq);
  }
  ba->J_valid = !fuseTop;
  ba->top_valid = fuseTop;
  SOS_HIP(hipGetLastError());
  sos_ba::StepPend &pd = ba->pend;
  pd.active = true;
  pd.waitSeq = waitSeq; pd.havePointStep = havePointStep; pd.stepfacD = stepfacD;
  pd.t0 = t0; pd.t1 = t1; pd.t2 = now_s(); pd.t3 = pd.t2;
  pd.fuseTop = fuseTop; pd.chainPublish = chainPublish; pd.applyRes = applyRes != 0;
  pd.prefetchEnqueued = ba->prefetch && applyRes;
  if (pd.prefetchEnqueued) {  // the next iteration's accumulate + stitch runs while the host digests this step
    if (!waitSeq) SOS_HIP(hipEventRecord(ba->ev_step, st));
    enqueue_gn_accumulate(ba, fuseTop, chainPublish ? reinterpret_cast<int *>(ba->pin_dev + ba->pin_flags) : nullptr, waitSeq);
    SOS_HIP(hipGetLastError());
    ba->acc_inflight = true;
    pd.t3 = now_s();
  }
  return SOS_OK;
}
// ... and the collecting half: wait for the step (never for the accumulate behind it), unpack the mapped results
static int lin_apply_finish(sos_ba *ba, double *energySum, float *newestEnergies, int *newestCount, float *pointStep) {
  hipStream_t st = ba->ctx->stream;
  char *po = ba->pin + ba->pin_out;
  sos_ba::StepPend &pd = ba->pend;
  const int waitSeq = pd.waitSeq;
  pd.active = false;
  if (waitSeq) {
    const int rcw = wait_flag(ba, ba->pin_flags, waitSeq);
    if (rcw) return rcw;
  } else if (pd.prefetchEnqueued) {
    SOS_HIP(hipEventSynchronize(ba->ev_step));
  } else {
    SOS_HIP(hipStreamSynchronize(st));
  }
  const double t4 = now_s();
  ba->tm[0] += pd.t1 - pd.t0; ba->tm[2] += pd.t2 - pd.t1; ba->tm[3] += pd.t3 - pd.t2; ba->tm[4] += t4 - pd.t3; ba->tm_calls++;
  if (energySum) {
    const double *es = reinterpret_cast<const double *>(po + ba->out_esum);
    double e = 0;
    for (int t = 0; t < ba->ntilesA; t++) e += es[t];
    *energySum = e;
  }
  if (newestEnergies) {
    const float *ne = ba->comm ? ba->pin_newest : reinterpret_cast<const float *>(po + ba->out_newest);
    const int cnt = ba->comm ? ba->newest_cap * ba->comm_size : ba->newest_count;
    int k = 0;
    for (int i = 0; i < cnt; i++)
      if (ne[i] >= 0) newestEnergies[k++] = ne[i];
    if (newestCount) *newestCount = k;
  }
  const float *hs = reinterpret_cast<const float *>(po + ba->out_step);
  if (pd.havePointStep) {
    const float stepfacD = pd.stepfacD;
    for (int p = 0; p < ba->P; p++) {  // keep the host mirror of the snapshot in step with the device
      const float idn = ba->h_pts[p].idepth_scaled + stepfacD * hs[p];
      ba->h_pts[p].idepth_scaled = idn;
      ba->h_pts[p].idepth_zero_scaled = idn;
      ba->h_pts[p].deltaF = 0.f;
    }
    if (pointStep) memcpy(pointStep, hs, sizeof(float) * ba->P);
  }
  ba->tm[5] += now_s() - t4;
  return SOS_OK;
}
static int lin_apply_tail(sos_ba *ba, BaDev &dv, int applyRes, bool havePointStep, float stepfacD, double *energySum,
                          float *newestEnergies, int *newestCount, float *pointStep, double t0, double t1) {
  const int rc = lin_apply_enqueue(ba, dv, applyRes, havePointStep, stepfacD, t0, t1);
  if (rc) { ba->pend.active = false; return rc; }
  return lin_apply_finish(ba, energySum, newestEnergies, newestCount, pointStep);
}



extern "C" int sos_ba_gn_step(sos_ba *ba, const double *x, float stepfacD, const sos_calib *calib, const sos_precalc *precalc,
                              const float *adHTdeltaF, const float *cDeltaF, const float *frameEnergyTH, int applyRes,
                              double *energySum, float *newestEnergies, int *newestCount, float *pointStep) {
  const bool devStep = ba && ba->devstep && x && !precalc;  // the device derives poses / precalc / deltas from x itself
  if (!ba || !ba->have_window || !ba->have_state || !frameEnergyTH || (!devStep && (!calib || !precalc || !adHTdeltaF || !cDeltaF)))
    return SOS_ERR_STATE;
  if (devStep && !(ba->P > 0 && ba->d_adHostF.p && ba->d_adTargetF.p && ba->ds_n == ba->n)) return SOS_ERR_STATE;
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const size_t nn = (size_t)ba->n * ba->n;
  if (calib) {
    ba->calib = *calib;
    ba->dev.calib = *calib;
  }
  ba->acc_inflight = false, ba->top_valid = false;
  const double t0 = now_s();
  if (!devStep) {
    memcpy(pstg(ba, ba->st_pre), precalc, sizeof(sos_precalc) * nn);
    memcpy(pstg(ba, ba->st_adh), adHTdeltaF, sizeof(float) * 8 * nn);
    memcpy(pstg(ba, ba->st_cd), cDeltaF, sizeof(float) * 4);
    memcpy(pstg(ba, ba->st_th), frameEnergyTH, sizeof(float) * ba->n);
    memcpy(pstg(ba, ba->st_cal), calib, sizeof(sos_calib));
  }
  // outputs go straight to the device-mapped pinned block: per-tile energy sums, newest-frame energies, point steps
  char *po = ba->pin + ba->pin_out, *po_dev = ba->pin_dev + ba->pin_out;
  float *dstep = reinterpret_cast<float *>(po_dev + ba->out_step);
  BaDev dv = ba->dev;
  dv.tile_esum = reinterpret_cast<double *>(po_dev + ba->out_esum);
  dv.o_newest = ba->comm ? ba->d_newest_local.p : reinterpret_cast<float *>(po_dev + ba->out_newest);
  const double t1 = now_s();
  const bool resubAhead = !x && ba->resub_pending;
  ba->resub_pending = false;
  if (devStep) {  // back-substitution + the host's step / precalc work, one launch, nothing staged from the host
    static_assert(sizeof(BaDev) + sizeof(DevStep) + 64 < 4096, "kernel arguments of k_resub_devstep");
    const int n = ba->n, dim = 4 + 8 * n;
    DevStep g;
    for (int i = 0; i < dim; i++) g.xd[i] = x[i];
    for (int i = 0; i < n; i++) g.th[i] = frameEnergyTH[i];
    g.x_dev = nullptr; g.th_dev = 0; g.HM = g.bM = nullptr; g.bMd = nullptr; g.nid_part = nullptr;
    double *b = ba->d_ds;
    g.evalC2W = b; g.state_zero = b + 12 * n;
    g.state_in = b + 22 * n + 10 * n * ba->ds_cur; g.state_out = b + 22 * n + 10 * n * (ba->ds_cur ^ 1);
    g.calib_in = b + 42 * n + 8 * ba->ds_cur; g.calib_out = b + 42 * n + 8 * (ba->ds_cur ^ 1);
    g.abexp = b + 42 * n + 16;
    ba->ds_cur ^= 1;
    g.stage = ba->d_stage.p;
    g.st_pre = ba->st_pre; g.st_adh = ba->st_adh; g.st_cd = ba->st_cd; g.st_th = ba->st_th; g.st_cal = ba->st_cal;
    const int nPB = divup(ba->P, SOS_RSB), nEB = std::max(1, divup(8 * ba->ntiles, SOS_RSB));  /* at least one step block: a window without residual tiles still steps its frames */
    const size_t lds = std::max(sizeof(float) * (8 * nn + 4 + 8 * (size_t)n), sizeof(float) * (28 * nn + 16) + sizeof(double) * (34 * (size_t)n + 8));
    ba->tm[1] += now_s() - t1;
    k_resub_devstep<<<nPB + nEB, SOS_RSB, lds, st>>>(dv, ba->d_adHostF.p, ba->d_adTargetF.p, dstep, stepfacD, nPB, g, ba->d_t_pre.p);
  } else if (resubAhead) {  // the back-substitution is already running: only the stage-in is left, one launch
    const int n4 = (int)((ba->st_xc + 3) / 4), nSB = divup(n4, SOS_RSB), nEB = std::max(1, divup(8 * ba->ntiles, SOS_RSB));  /* at least one step block: a window without residual tiles still steps its frames */
    k_stage_expand<<<nSB + nEB, SOS_RSB, 0, st>>>(dv, reinterpret_cast<float4 *>(ba->d_stage.p), reinterpret_cast<const float4 *>(ba->pin_dev + ba->pin_stage),
                                                 n4, nSB, reinterpret_cast<const float4 *>(ba->pin_dev + ba->pin_stage + sizeof(float) * ba->st_pre),
                                                 ba->d_t_pre.p);
  } else if (x && ba->P > 0 && ba->d_adHostF.p && ba->d_adTargetF.p) {
    // back-substitution from x alone + stage-in of the linearisation inputs: one launch
    XArg xa;
    const int dim = 4 + 8 * ba->n;
    for (int i = 0; i < dim; i++) xa.v[i] = (float)x[i];
    const int n4 = (int)((ba->st_xc + 3) / 4), nPB = divup(ba->P, SOS_RSB);
    ba->tm[1] += now_s() - t1;
    const int nSB = divup(n4, SOS_RSB), nEB = std::max(1, divup(8 * ba->ntiles, SOS_RSB));  /* at least one step block: a window without residual tiles still steps its frames */
    k_resub_fused<<<nPB + nSB + nEB, SOS_RSB, sizeof(float) * (8 * nn + 4 + 8 * (size_t)ba->n), st>>>(
        dv, xa, ba->d_adHostF.p, ba->d_adTargetF.p, dstep, stepfacD, nPB, reinterpret_cast<float4 *>(ba->d_stage.p),
        reinterpret_cast<const float4 *>(ba->pin_dev + ba->pin_stage), n4, nSB,
        reinterpret_cast<const float4 *>(ba->pin_dev + ba->pin_stage + sizeof(float) * ba->st_pre), ba->d_t_pre.p, nullptr);
  } else if (x) {
    fill_x(ba, x);
    stage_in(ba, ba->st_floats);
    if (ba->P > 0)
      k_resubstitute<<<divup(ba->P, 64), 64, 0, st>>>(dv, stg(ba, ba->st_xc), stg(ba, ba->st_xad), dstep, 1, stepfacD);
  } else {
    stage_in(ba, ba->st_xc);
  }
  return lin_apply_tail(ba, dv, applyRes, x != nullptr || resubAhead, stepfacD, energySum, newestEnergies, newestCount, pointStep, t0, t1);
}

// ------------------------------------------------------------------------------------------------
// keyframe-rate fused calls (the front and the back of FullSystem::optimize around the Gauss-Newton loop)
// ------------------------------------------------------------------------------------------------
// FS/FullSystemOptimize.cpp:316-344 with setting_forceAceptStep: resetOOB of the active residuals, linearizeAll(false),
// applyRes -- one launch chain, the first iteration's accumulate enqueued behind it when prefetch is on.
extern "C" int sos_ba_linearize_apply(sos_ba *ba, const float *frameEnergyTH, int resetOOB, double *energySum, float *newestEnergies,
                                      int *newestCount) {
  if (!ba || !ba->have_window || !ba->have_state || !frameEnergyTH) return SOS_ERR_STATE;
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  ba->acc_inflight = false, ba->top_valid = false;
  const double t0 = now_s();
  SOS_HIP(hipStreamSynchronize(st));  // the mapped stage block is about to be rewritten
  memcpy(pstg(ba, ba->st_th), frameEnergyTH, sizeof(float) * ba->n);
  SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_th), pstg(ba, ba->st_th), sizeof(float) * ba->n, hipMemcpyHostToDevice, st));
  if (resetOOB && ba->ntilesA > 0) k_reset_oob<<<divup(ba->ntilesA * SOS_TILE, 256), 256, 0, st>>>(ba->dev);
  char *po_dev = ba->pin_dev + ba->pin_out;
  BaDev dv = ba->dev;
  dv.tile_esum = reinterpret_cast<double *>(po_dev + ba->out_esum);
  dv.o_newest = ba->comm ? ba->d_newest_local.p : reinterpret_cast<float *>(po_dev + ba->out_newest);
  ba->pending_new = false;
  return lin_apply_tail(ba, dv, 1, false, 0.f, energySum, newestEnergies, newestCount, nullptr, t0, now_s());
}

// per residual (thread i < ntilesA * SOS_TILE) the record of sos_ba_linearize_final in the original order; per point
// (thread i < P) the isNew statistics of FS/FullSystemOptimize.cpp:55-71 over its active residuals, in residualsAll order
__global__ __launch_bounds__(256) void k_final_pack(BaDev d, sos_resid_final *__restrict__ out, float *__restrict__ pmax, int *__restrict__ pcnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.ntilesA * SOS_TILE) {
    const int s = i, orig = d.s_orig[s];
    if (orig >= 0) {
      sos_resid_final r;
      r.state_NewEnergy = d.s_newenergy[s];
      r.state_NewEnergyWithOutlier = d.s_newenergywo[s];
      r.state_energy = d.s_energy[s];
      r.centerProjectedTo[0] = d.s_center[3 * s]; r.centerProjectedTo[1] = d.s_center[3 * s + 1]; r.centerProjectedTo[2] = d.s_center[3 * s + 2];
      r.state_NewState = d.s_newstate[s];
      r.state_state = d.s_state[s];
      r.active = (d.s_flags[s] & DF_ACTIVE) ? 1 : 0;
      r.pad = 0;
      out[orig] = r;
    }
  }
  if (i < d.P) {
    float mx = -1.f;
    int cnt = 0;
    for (int q = d.p_begin[i]; q < d.p_begin[i + 1]; q++) {
      const int2 e = d.p_list2[q];  // (s, n * host + target)
      const unsigned f = d.s_flags[e.x];
      if ((f & (DF_ACTIVE | DF_ISNEW | DF_LINEARIZED)) != (DF_ACTIVE | DF_ISNEW)) continue;
      const int h = e.y / d.n, t = e.y - h * d.n;
      const sos_precalc &pc = d.precalc[h + d.n * t];
      const float4 g = d.r_geo[e.x];
      const float *K = pc.PRE_KRKiTll, *Kt = pc.PRE_KtTll;
      const float inf0 = K[0] * g.x + K[1] * g.y + K[2], inf1 = K[3] * g.x + K[4] * g.y + K[5], inf2 = K[6] * g.x + K[7] * g.y + K[8];
      const float q0 = inf0 + Kt[0] * g.z, q1 = inf1 + Kt[1] * g.z, q2 = inf2 + Kt[2] * g.z;
      const float dx = inf0 / inf2 - q0 / q2, dy = inf1 / inf2 - q1 / q2;
      const float relBS = (float)(0.01 * sqrtf(dx * dx + dy * dy));
      if (relBS > mx) mx = relBS;
      cnt++;
    }
    pmax[i] = mx;
    pcnt[i] = cnt;
  }
}

// FullSystem::linearizeAll(true) (FS/FullSystemOptimize.cpp:125-182): linearize + applyRes(true) of every active residual,
// then what the host bookkeeping reads, in ONE device-to-host copy into a pinned block owned by the handle
extern "C" int sos_ba_linearize_final(sos_ba *ba, const float *frameEnergyTH, int resetOOB, double *energySum, const sos_resid_final **records,
                                      const float **pointMaxRelBaseline, const int32_t **pointNewGood, float *newestEnergies,
                                      int *newestCount) {
  if (!ba || !ba->have_window || !ba->have_state || !frameEnergyTH || !records || !pointMaxRelBaseline || !pointNewGood) return SOS_ERR_STATE;
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  ba->acc_inflight = false, ba->top_valid = false;
  const size_t Rr = ba->R ? ba->R : 1, Pp = ba->P ? ba->P : 1;
  const size_t o_pmax = (sizeof(sos_resid_final) * Rr + 255) / 256 * 256, o_pcnt = o_pmax + (sizeof(float) * Pp + 255) / 256 * 256,
               bytes = o_pcnt + sizeof(int) * Pp;
  if (bytes > ba->fin_cap) {
    SOS_HIP(hipStreamSynchronize(st));
    if (ba->fin_host) hipHostFree(ba->fin_host);
    if (ba->fin_dev) hipFree(ba->fin_dev);
    ba->fin_host = ba->fin_dev = nullptr;
    ba->fin_cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    SOS_HIP(hipHostMalloc((void **)&ba->fin_host, want, hipHostMallocDefault));
    if (hipMalloc((void **)&ba->fin_dev, want) != hipSuccess) return SOS_ERR_NOMEM;
    ba->fin_cap = want;
  }
  const bool prefetch = ba->prefetch, fuseOnly = ba->fuse_only;
  ba->prefetch = ba->fuse_only = false;  // nothing follows this linearisation, and its Jacobians are stored (marginalisation reads them)
  int rc = sos_ba_linearize_apply(ba, frameEnergyTH, resetOOB, energySum, newestEnergies, newestCount);
  ba->prefetch = prefetch;
  ba->fuse_only = fuseOnly;
  if (rc) return rc;
  const int nthr = std::max(ba->ntilesA * SOS_TILE, ba->P);
  if (nthr > 0)
    k_final_pack<<<divup(nthr, 256), 256, 0, st>>>(ba->dev, reinterpret_cast<sos_resid_final *>(ba->fin_dev),
                                                  reinterpret_cast<float *>(ba->fin_dev + o_pmax), reinterpret_cast<int *>(ba->fin_dev + o_pcnt));
  SOS_HIP(hipGetLastError());
  SOS_HIP(hipMemcpyAsync(ba->fin_host, ba->fin_dev, bytes, hipMemcpyDeviceToHost, st));
  SOS_HIP(hipStreamSynchronize(st));
  *records = reinterpret_cast<const sos_resid_final *>(ba->fin_host);
  *pointMaxRelBaseline = reinterpret_cast<const float *>(ba->fin_host + o_pmax);
  *pointNewGood = reinterpret_cast<const int32_t *>(ba->fin_host + o_pcnt);
  return SOS_OK;
}

extern "C" int sos_ba_gn_devstep_begin(sos_ba *ba, const sos_gn_frame *frames, const double *calib_value4, const double *calib_value_zero4) {
  if (!ba || !frames || !calib_value4 || !calib_value_zero4) return SOS_ERR_ARG;
  if (!ba->have_window || !ba->have_state || ba->n < 1 || ba->n > 17 || ba->P <= 0 || !ba->d_adHostF.p || !ba->d_adTargetF.p)
    return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  const int n = ba->n;
  const size_t cnt = (size_t)43 * n + 16;
  if (ba->ds_n != n) {
    SOS_HIP(hipStreamSynchronize(ba->ctx->stream));
    if (ba->d_ds) hipFree(ba->d_ds);
    if (ba->ds_host) hipHostFree(ba->ds_host);
    ba->d_ds = nullptr;
    ba->ds_host = nullptr;
    ba->ds_n = 0;
    SOS_HIP(hipMalloc(&ba->d_ds, sizeof(double) * cnt));
    SOS_HIP(hipHostMalloc((void **)&ba->ds_host, sizeof(double) * cnt * 2, hipHostMallocDefault));
    ba->ds_n = n;
    ba->ds_flip = 0;
  }
  // pinned source, two halves used alternately: the copy is asynchronous and a second begin may follow before it has run
  ba->ds_flip ^= 1;
  double *h = ba->ds_host + (size_t)ba->ds_flip * cnt;
  memset(h, 0, sizeof(double) * cnt);
  for (int f = 0; f < n; f++) {
    memcpy(&h[12 * f], frames[f].camToWorld_evalPT, sizeof(double) * 12);
    memcpy(&h[12 * n + 10 * f], frames[f].state_zero, sizeof(double) * 10);
    memcpy(&h[22 * n + 10 * f], frames[f].state, sizeof(double) * 10);
    h[42 * n + 16 + f] = (double)frames[f].ab_exposure;
  }
  for (int i = 0; i < 4; i++) { h[42 * n + i] = calib_value4[i]; h[42 * n + 4 + i] = calib_value_zero4[i]; }
  ba->ds_cur = 0;
  SOS_HIP(hipMemcpyAsync(ba->d_ds, h, sizeof(double) * cnt, hipMemcpyHostToDevice, ba->ctx->stream));
  ba->devstep = true;
  return SOS_OK;
}
extern "C" int sos_ba_gn_devstep_end(sos_ba *ba) {
  if (!ba) return SOS_ERR_ARG;
  ba->devstep = false;
  return SOS_OK;
}

extern "C" int sos_ba_calc_lenergy(sos_ba *ba, double *E) {
  if (!ba || !ba->have_window || !ba->have_state || !E) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  double e = 0;
  if (ba->Rpad > 0) {
    k_lenergy<<<divup(ba->Rpad, 256), 256, 0, st>>>(ba->dev, ba->d_perres.p);
    k_sum_double<<<1, 1024, 0, st>>>(ba->d_perres.p, ba->Rpad, ba->d_scalar.p);
    SOS_HIP(hipMemcpyAsync(&e, ba->d_scalar.p, sizeof(double), hipMemcpyDeviceToHost, st));
    SOS_HIP(hipStreamSynchronize(st));
  }
  // E.updateSingle(deltaF^2 * priorF) per point (OB/EnergyFunctional.cpp:620)
  for (int p = 0; p < ba->P; p++) e += (double)(ba->h_pts[p].deltaF * ba->h_pts[p].deltaF * ba->h_pts[p].priorF);
  *E = e;
  return SOS_OK;
}

extern "C" int sos_ba_accumulate_marg(sos_ba *ba, const int32_t *pointIdx, int count, double *M, double *Mb, double *Msc,
                                      double *Mbsc, int *resInM) {
  if (ba && ba->have_window && ba->have_state) ensure_J(ba);
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || !ba->have_state || (count && !pointIdx)) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  const int n = ba->n;
  const size_t nn = (size_t)n * n;
  // virtual tiles: residuals of the listed points grouped by pair (addPoint<2> visits every active one)
  std::vector<std::vector<int>> bypair(nn);
  std::vector<int> plist(pointIdx, pointIdx + count);
  for (int k = 0; k < count; k++) {
    const int p = pointIdx[k];
    if (p < 0 || p >= ba->P) return SOS_ERR_ARG;
    for (int o = ba->h_p_begin[p]; o < ba->h_p_begin[p + 1]; o++)
      bypair[ba->h_res[o].host + n * ba->h_res[o].target].push_back(ba->h_s_of_orig[o]);
  }
  std::vector<int> list, list_pair, pair_tile_begin(nn + 1, 0);
  for (size_t k = 0; k < nn; k++) {
    pair_tile_begin[k] = (int)list_pair.size();
    for (size_t q = 0; q < bypair[k].size(); q += SOS_TILE) {
      list_pair.push_back((int)k);
      for (int j = 0; j < SOS_TILE; j++) list.push_back(q + j < bypair[k].size() ? bypair[k][q + j] : -1);
    }
  }
  pair_tile_begin[nn] = (int)list_pair.size();
  const int nvt = (int)list_pair.size();
  // chunks of the listed points per host
  std::vector<int> chunk_pt, host_chunk_begin(n + 1, 0);
  {
    std::vector<std::vector<int>> byhost(n);
    for (int p : plist) byhost[ba->h_pts[p].host].push_back(p);
    for (int h = 0; h < n; h++) {
      host_chunk_begin[h] = (int)(chunk_pt.size() / SOS_GC);
      for (size_t q = 0; q < byhost[h].size(); q += SOS_GC)
        for (int k = 0; k < SOS_GC; k++) chunk_pt.push_back(q + k < byhost[h].size() ? byhost[h][q + k] : -1);
    }
    host_chunk_begin[n] = (int)(chunk_pt.size() / SOS_GC);
  }
  const int nch = host_chunk_begin[n];
  // pack [list | list_pair | pair_tile_begin | plist | chunk_pt | host_chunk_begin] into one int upload
  std::vector<int> blob;
  auto put = [&](const std::vector<int> &v) { size_t o = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); blob.resize((blob.size() + 3) / 4 * 4); return o; };
  const size_t o_list = put(list), o_lp = put(list_pair), o_ptb = put(pair_tile_begin), o_pl = put(plist),
               o_cp = put(chunk_pt), o_hcb = put(host_chunk_begin);
  if (blob.empty()) blob.push_back(0);
  int rc = upload(st, ba->d_tmp_int, blob);
  if (rc) return rc;
  const int *B = ba->d_tmp_int.p;
  DevBuf<float> tp, gp, acc;
  if ((rc = tp.ensure((size_t)(nvt ? nvt : 1) * SOS_TOPN + 64 * SOS_TOPN))) return rc;
  if ((rc = gp.ensure((size_t)(nch ? nch : 1) * ba->Dm * ba->Dm))) return rc;
  if ((rc = acc.ensure(ba->acc_floats))) return rc;
  SOS_HIP(hipMemsetAsync(acc.p, 0, sizeof(float) * ba->acc_floats, st));
  if (nvt > 0)
    k_top_accumulate<true><<<divup(nvt, 8), 256, 0, st>>>(ba->dev, 0, nvt, 2, B + o_list, B + o_lp, tp.p, nullptr);
  if (count > 0) k_point_prep<<<divup(count, 64), 64, 0, st>>>(ba->dev, 0, B + o_pl, count);
  if (nch > 0) k_sc_gram<<<nch, 256, gram_lds(ba), st>>>(ba->dev, B + o_cp, ba->Dm, ba->ld, gp.p);
  ReduceArgs a;
  fill_reduce(ba, a, tp.p, B + o_ptb, gp.p, B + o_hcb, nch, 1, acc.p);
  k_reduce_all<<<a.b_top + a.b_sc + a.b_tail, 128, 0, st>>>(a);
  launch_stitch(ba, acc.p, 1);
  SOS_HIP(hipGetLastError());
  rc = fetch_hb(ba, acc.p, M, Mb, nullptr, nullptr, Msc, Mbsc, resInM, nullptr, false);
  tp.release(); gp.release(); acc.release();
  return rc;
}

extern "C" int sos_ba_update_point_priors(sos_ba *ba, const int32_t *pointIdx, const float *priorF, int count) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || (count && (!pointIdx || !priorF))) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  for (int k = 0; k < count; k++) {
    if (pointIdx[k] < 0 || pointIdx[k] >= ba->P) return SOS_ERR_ARG;
    ba->h_pts[pointIdx[k]].priorF = priorF[k];
  }
  if (ba->P) SOS_HIP(hipMemcpyAsync(ba->d_pts.p, ba->h_pts.data(), sizeof(sos_point) * ba->P, hipMemcpyHostToDevice, ba->ctx->stream));
  SOS_HIP(hipStreamSynchronize(ba->ctx->stream));
  return SOS_OK;
}

// ---- inspection ---------------------------------------------------------------------------------
extern "C" int sos_ba_get_jacobian(sos_ba *ba, int residIdx, int which, sos_rawjac *out) {
  if (ba && ba->have_window && ba->have_state && which == 0) ensure_J(ba);
  if (!ba || !ba->have_window || !out || residIdx < 0 || residIdx >= ba->R) return SOS_ERR_ARG;
  if (which != 0 && !ba->d_Jnew.p) return SOS_ERR_STATE;  // no sos_ba_linearize has filled PointFrameResidual::J yet
  SOS_HIP(hipSetDevice(ba->ctx->device));
  int rc = ba->d_rawjac.ensure(1);
  if (rc) return rc;
  BaDev dv = ba->dev;
  if (which != 0) dv.J = ba->d_Jnew.p;
  k_get_jac<<<1, 64, 0, ba->ctx->stream>>>(dv, ba->h_s_of_orig[residIdx], ba->d_rawjac.p);
  SOS_HIP(hipMemcpyAsync(out, ba->d_rawjac.p, sizeof(sos_rawjac), hipMemcpyDeviceToHost, ba->ctx->stream));
  SOS_HIP(hipStreamSynchronize(ba->ctx->stream));
  return SOS_OK;
}

extern "C" int sos_ba_get_residual_flags(sos_ba *ba, uint32_t *flags, int32_t *state_state, float *state_energy) {
  if (!ba || !ba->have_window) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  hipStream_t st = ba->ctx->stream;
  const size_t Rp = ba->Rpad;
  std::vector<uint8_t> f(Rp), s(Rp);
  std::vector<float> e(Rp);
  if (Rp) {
    SOS_HIP(hipMemcpyAsync(f.data(), ba->d_s_flags.p, Rp, hipMemcpyDeviceToHost, st));
    SOS_HIP(hipMemcpyAsync(s.data(), ba->d_s_state.p, Rp, hipMemcpyDeviceToHost, st));
    SOS_HIP(hipMemcpyAsync(e.data(), ba->d_s_energy.p, sizeof(float) * Rp, hipMemcpyDeviceToHost, st));
  }
  SOS_HIP(hipStreamSynchronize(st));
  for (int o = 0; o < ba->R; o++) {
    const int si = ba->h_s_of_orig[o];
    if (flags) {
      uint32_t v = 0;
      if (f[si] & DF_ACTIVE) v |= SOS_RF_ACTIVE;
      if (f[si] & DF_LINEARIZED) v |= SOS_RF_LINEARIZED;
      if (f[si] & DF_ISNEW) v |= SOS_RF_ISNEW;
      flags[o] = v;
    }
    if (state_state) state_state[o] = s[si];
    if (state_energy) state_energy[o] = e[si];
  }
  return SOS_OK;
}

static int gather_sorted8(sos_ba *ba, const float *dsrc, float *out) {
  const size_t Rp = ba->Rpad;
  std::vector<float> v(Rp * 8);
  if (Rp) SOS_HIP(hipMemcpyAsync(v.data(), dsrc, sizeof(float) * 8 * Rp, hipMemcpyDeviceToHost, ba->ctx->stream));
  SOS_HIP(hipStreamSynchronize(ba->ctx->stream));
  for (int o = 0; o < ba->R; o++) memcpy(out + 8 * (size_t)o, &v[8 * (size_t)ba->h_s_of_orig[o]], 8 * sizeof(float));
  return SOS_OK;
}
extern "C" int sos_ba_get_JpJdF(sos_ba *ba, float *JpJdF) {
  if (!ba || !ba->have_window || !JpJdF) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  return gather_sorted8(ba, ba->d_JpJd.p, JpJdF);
}
extern "C" int sos_ba_get_res_toZeroF(sos_ba *ba, float *rtz) {
  if (!ba || !ba->have_window || !rtz) return SOS_ERR_STATE;
  SOS_HIP(hipSetDevice(ba->ctx->device));
  return gather_sorted8(ba, ba->d_s_rtz.p, rtz);
}

// ---- kernel timing with HIP events on the context's stream ----------------------------------------
extern "C" int sos_ba_time_kernel(sos_ba *ba, const char *kernel, const float *frameEnergyTH, int iters, float *avg_ms) {
  if (ba) ba->acc_inflight = false, ba->top_valid = false;  // any state change invalidates a prefetched accumulate
  if (!ba || !ba->have_window || !ba->have_state || !kernel || iters <= 0 || !avg_ms) return SOS_ERR_STATE;
  sos_ctx *c = ba->ctx;
  SOS_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  if (frameEnergyTH) SOS_HIP(hipMemcpyAsync(stg(ba, ba->st_th), frameEnergyTH, sizeof(float) * ba->n, hipMemcpyHostToDevice, st));
  const std::string k(kernel);
  auto once = [&]() -> int {
    if (k == "linearize") return launch_linearize(ba, 0);
    if (k == "linearize_apply") return launch_linearize(ba, 1);
    if (k == "linearize_fused") {  // what the pipelined iterations run: linearize + applyRes + tile block sums, no J store
      BaDev dv = ba->dev;  // exactly the loop's launch: no original-order result copies (lin_apply_tail)
      dv.o_newstate = nullptr; dv.o_newenergy = nullptr; dv.o_newenergywo = nullptr; dv.o_center = nullptr;
      dv.tile_esum = reinterpret_cast<double *>(ba->pin_dev + ba->pin_out + ba->out_esum);
      dv.o_newest = reinterpret_cast<float *>(ba->pin_dev + ba->pin_out + ba->out_newest);
      launch_lin_kernel(ba, dv, 1, ba->d_top_part.p);
      ba->J_valid = false;
      return SOS_OK;
    }
    if (k == "stream_equal") {  // streaming read of as many bytes as the fused linearisation's algorithmic traffic
      const size_t floats = (size_t)ba->R * 1088 / 4;
      if (ba->d_calib.cap < floats) {
        if (ba->d_calib.ensure(floats)) return SOS_ERR_NOMEM;
        SOS_HIP(hipMemsetAsync(ba->d_calib.p, 0, sizeof(float) * floats, st));
      }
      k_calib_read<<<2048, 256, 0, st>>>(reinterpret_cast<const float4 *>(ba->d_calib.p), floats / 4, ba->d_top_part.p);
      return SOS_OK;
    }
    if (k == "stream_large") {  // 1 GiB coalesced read: the streaming bandwidth of the chip, measured in the same run
      const size_t floats = (size_t)1 << 28;
      if (ba->d_large.cap < floats) {
        if (ba->d_large.ensure(floats)) return SOS_ERR_NOMEM;
        SOS_HIP(hipMemsetAsync(ba->d_large.p, 0, sizeof(float) * floats, st));
      }
      k_stream_large<<<256 * 16, 256, 0, st>>>(reinterpret_cast<const float4 *>(ba->d_large.p), floats / 4, ba->d_top_part.p);
      return SOS_OK;
    }
    if (k == "lin_floor") {  // empty kernel with the linearisation's grid, block and LDS size
      int ndFloor;
      k_lin_floor<<<lin_grid(ba, &ndFloor), 256 * L2_TILES, 0, st>>>(nullptr);
      return SOS_OK;
    }
    if (k == "sc_gram_prep") {
      if (ba->nchunks > 0) k_sc_gram_prep<<<ba->nchunks, 256, gram_lds(ba), st>>>(ba->dev, ba->d_chunk_pt.p, ba->Dm, ba->ld, ba->d_gram_part.p, nullptr, 0);
      return SOS_OK;
    }
    if (k == "sc_gram_abs" || k == "abs_reduce_stitch1" || k == "abs_stitch2") {  // the kernels of the absolute-coordinate Schur path, one at a time
      if (!abs_path_ok(ba)) return SOS_OK;
      const int n = ba->n;
      const size_t nn = (size_t)n * n;
      if (k == "sc_gram_abs") {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_sc_gram_abs), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        k_sc_gram_abs<<<ba->nchunks, 256, sizeof(float) * gram_abs_lds_floats(n, ba->ld), st>>>(ba->dev, ba->d_chunk_pt.p, ba->Dm, ba->ld, ba->d_gram_part.p,
                                                                                                 ba->d_adHostF.p, ba->d_adTargetF.p, nullptr, 0);
        return SOS_OK;
      }
      AbsStitchArgs a;
      a.n = n; a.Dm = ba->Dm; a.nchunks = ba->nchunks;
      a.top_part = ba->d_top_part.p; a.pair_tile_begin = ba->d_pair_tile_begin.p; a.gram_part = ba->d_gram_part.p;
      a.adHost = ba->d_adHost.p; a.adTarget = ba->d_adTarget.p;
      a.Ctop = ba->d_C.p; a.Pcc = ba->d_C.p + 2 * nn * SOS_TOPC;
      a.H = ba->d_Hout.p; a.mode_stride = ba->hb_mode_stride;
      a.sg = {nullptr, nullptr, 0, 0};
      const int T = ba->Dm >> 4;
      if (k == "abs_reduce_stitch1") k_abs_reduce_stitch1<<<(int)nn + T * (T + 1) / 2 * 16, 128, 0, st>>>(a);
      else k_abs_stitch2<<<n * (n + 1) / 2 + 1, 64, 0, st>>>(a);
      return SOS_OK;
    }
    if (k == "exchange") {  // the per-iteration collective alone: all-reduce of a buffer the size of the packed fp32 accumulator
      if (!ba->comm) return SOS_OK;  // (every rank has to make this call with the same iteration count)
      if (ba->d_xchg.cap < ba->acc_floats) {
        if (ba->d_xchg.ensure(ba->acc_floats)) return SOS_ERR_NOMEM;
        SOS_HIP(hipMemsetAsync(ba->d_xchg.p, 0, sizeof(float) * ba->acc_floats, st));
      }
      return sos_comm_allreduce_sum_f32(ba->comm, ba->d_xchg.p, ba->acc_floats, st);
    }
    if (k == "apply_res") return sos_ba_apply_res(ba);
    if (k == "top_accumulate") return launch_top(ba);
    if (k == "sc_accumulate") return launch_sc(ba, 1);
    if (k == "reduce") return launch_reduce(ba);
    if (k == "stitch") return launch_stitch(ba, ba->d_acc.p, 2);
    if (k == "accumulate_local") return sos_ba_accumulate_local(ba);
    if (k == "calib_read") {  // streams the whole Jacobian buffer: ntiles * 9216 B
      k_calib_read<<<2048, 256, 0, st>>>(reinterpret_cast<const float4 *>(ba->d_J.p), (size_t)ba->ntiles * SOS_TILE_FLOATS / 4, ba->d_top_part.p);
      return SOS_OK;
    }
    if (k == "calib_write") {  // overwrites the Gram partials (scratch, regenerated by every accumulate): nchunks * Dm^2 * 4 B
      k_calib_write<<<2048, 256, 0, st>>>(reinterpret_cast<float4 *>(ba->d_gram_part.p), (size_t)ba->nchunks * ba->Dm * ba->Dm / 4);
      return SOS_OK;
    }
    if (k == "resub_fused") {  // the back-substitution of a pipelined step (x = 0: the state is left where it is)
      if (ba->P > 0 && ba->d_adHostF.p && ba->d_adTargetF.p) {
        XArg xa;
        memset(&xa, 0, sizeof(xa));
        const size_t nn = (size_t)ba->n * ba->n;
        const int nPB = divup(ba->P, SOS_RSB);
        k_resub_fused<<<nPB, SOS_RSB, sizeof(float) * (8 * nn + 4 + 8 * (size_t)ba->n), st>>>(
            ba->dev, xa, ba->d_adHostF.p, ba->d_adTargetF.p, reinterpret_cast<float *>(ba->d_outpack.p + ba->out_step), 0.f, nPB, nullptr, nullptr, 0, 0,
            nullptr, nullptr, nullptr);
      }
      return SOS_OK;
    }
    if (k == "resubstitute") {
      if (ba->P > 0)
        k_resubstitute<<<divup(ba->P, 64), 64, 0, st>>>(ba->dev, stg(ba, ba->st_xc), stg(ba, ba->st_xad),
                                                        reinterpret_cast<float *>(ba->d_outpack.p + ba->out_step), 0, 1.f);
      return SOS_OK;
    }
    return SOS_ERR_ARG;
  };
  int rc = once();  // warm-up, also validates the name
  if (rc) return rc;
  SOS_HIP(hipStreamSynchronize(st));
  SOS_HIP(hipEventRecord(c->ev0, st));
  for (int i = 0; i < iters; i++) once();
  SOS_HIP(hipEventRecord(c->ev1, st));
  SOS_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  SOS_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  SOS_HIP(hipGetLastError());
  *avg_ms = ms / iters;
  return SOS_OK;
}

#include "sos_gn_resident.inc"
