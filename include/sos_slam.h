/*
 * sos_slam.h -- C-ABI of the MI355X (gfx950) hot path of SOS-SLAM's sliding-window photometric
 * bundle adjustment (OptimizationBackend) and CoarseTracker direct-alignment loop.
 *
 * The reference has no FFI for this path; the seam is the set of C++ call sites listed in
 * SURVEY.md section 8(b).  Each entry point below cites the reference call it replaces
 * (paths relative to the reference root, `src/` omitted: FS = FullSystem, OB = OptimizationBackend).
 *
 * Conventions
 *   - every function returns an int status: 0 = SOS_OK, negative = error (no exceptions, no exit()).
 *   - the caller owns every host buffer; the library owns device memory behind opaque handles.
 *   - handles are not thread-safe; distinct handles are independent; all work of a context is
 *     issued on ONE HIP stream (the caller's, or one created by the context).
 *   - NaN / Inf energies are returned as data (the caller turns them into `isLost`,
 *     FS/FullSystemOptimize.cpp:427-432).
 *   - matrices are row-major unless stated; the pair index is  h + n*t  (host + nFrames*target), the
 *     index used by adHost/adTarget/adHTdeltaF/acc (OB/EnergyFunctional.cpp:82-83,169).
 *   - fp32 arithmetic is evaluated WITHOUT fused multiply-add contraction and in the operation order
 *     documented in DESIGN.md ("arithmetic convention"), identical on the device and in oracle/.
 */
#ifndef SOS_SLAM_H
#define SOS_SLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * constants of the path (util/NumType.h:36-45, util/settings.h:187-189, FS/HessianBlocks.h:53-89)
 * ---------------------------------------------------------------------------------------------- */
#define SOS_CPARS 4               /* CPARS */
#define SOS_PATTERN_NUM 8         /* patternNum, staticPattern[8] */
#define SOS_MAX_FRAMES 32         /* upper bound on window size accepted by the library */
#define SOS_MAX_SLOTS 64          /* image slots of a context */
#define SOS_PYR_LEVELS 6          /* PYR_LEVELS */

#define SOS_SCALE_IDEPTH 1.0f
#define SOS_SCALE_XI_ROT 1.0f
#define SOS_SCALE_XI_TRANS 0.5f
#define SOS_SCALE_F 50.0f
#define SOS_SCALE_C 50.0f
#define SOS_SCALE_A 10.0f
#define SOS_SCALE_B 1000.0f

/* ResState (FS/Residuals.h:43) */
#define SOS_RES_IN 0
#define SOS_RES_OOB 1
#define SOS_RES_OUTLIER 2

/* sos_resid.flags bits */
#define SOS_RF_ACTIVE 1u          /* EFResidual::isActiveAndIsGoodNEW */
#define SOS_RF_LINEARIZED 2u      /* EFResidual::isLinearized */
#define SOS_RF_ISNEW 4u           /* PointFrameResidual::isNew */

/* status codes */
#define SOS_OK 0
#define SOS_ERR_ARG (-1)
#define SOS_ERR_HIP (-2)
#define SOS_ERR_STATE (-3)
#define SOS_ERR_NOMEM (-4)
#define SOS_ERR_TIMEOUT (-5) /* a device-resident loop gave up waiting for its own workgroups (see sos_tracker_set_lm_spin_limit) */

/* ------------------------------------------------------------------------------------------------
 * plain-data records
 * ---------------------------------------------------------------------------------------------- */

/* every global the path reads, made explicit (util/settings.cpp:47-119, util/globalCalib.cpp:58-64) */
typedef struct sos_params {
  int32_t w, h;                    /* wG[0], hG[0]; wM3G = w-3, hM3G = h-3 */
  float huberTH;                   /* setting_huberTH = 9 */
  float outlierTHSumComponent;     /* setting_outlierTHSumComponent = 50*50 */
  float affineOptModeA;            /* setting_affineOptModeA = 1e12 (<0: fixed) */
  float affineOptModeB;            /* setting_affineOptModeB = 1e8 */
  float idepthFixPrior;            /* 50*50 */
  float idepthFixPriorMargFac;     /* 600*600 */
  float margWeightFac;             /* 0.5*0.5 */
  float initialCalibHessian;       /* 5e9 */
  float coarseCutoffTH;            /* 20 */
  float frameEnergyTHN;            /* 0.7 */
  float frameEnergyTHFacMedian;    /* 1.5 */
  float frameEnergyTHConstWeight;  /* 0.5 */
  float overallEnergyTHWeight;     /* 1 */
  float reserved[3];
} sos_params;

/* CalibHessian::value_scaledf / value_scaledi (FS/HessianBlocks.h:476-514) */
typedef struct sos_calib {
  float fxl, fyl, cxl, cyl;        /* value_scaledf */
  float fxli, fyli, cxli, cyli;    /* value_scaledi = 1/fx, 1/fy, -cx/fx, -cy/fy */
} sos_calib;

/* the part of FrameFramePrecalc the backend reads (FS/HessianBlocks.h:109-134; set():
 * FS/HessianBlocks.cpp:431-461).  Computed on the host in fp64 and cast. */
typedef struct sos_precalc {
  float PRE_KRKiTll[9];            /* row-major */
  float PRE_KtTll[3];
  float PRE_RTll_0[9];             /* row-major */
  float PRE_tTll_0[3];
  float PRE_aff_mode[2];
  float PRE_b0_mode;
  float pad;                       /* 28 floats */
} sos_precalc;

/* device-relevant part of PointHessian + EFPoint (FS/HessianBlocks.h:556-649,
 * OB/EnergyFunctionalStructs.h:83-114) */
typedef struct sos_point {
  float u, v;
  float idepth_scaled;
  float idepth_zero_scaled;
  float color[SOS_PATTERN_NUM];
  float weights[SOS_PATTERN_NUM];
  float priorF;                    /* EFPoint::priorF */
  float deltaF;                    /* EFPoint::deltaF = idepth - idepth_zero */
  int32_t host;                    /* frame idx of the host */
  int32_t pad;                     /* 24 x 4 B */
} sos_point;

/* PointFrameResidual + EFResidual, index-stable ids (FS/Residuals.h:49-93,
 * OB/EnergyFunctionalStructs.h:43-81).  Residuals of one point must be contiguous and in the order
 * of EFPoint::residualsAll; points are in EnergyFunctional::allPoints order (frames -> points,
 * OB/EnergyFunctional.cpp:1192-1199). */
typedef struct sos_resid {
  int32_t point;                   /* index into the point array */
  int32_t host, target;            /* hostIDX, targetIDX */
  uint32_t flags;                  /* SOS_RF_* */
  int32_t state_state;             /* SOS_RES_* */
  float state_energy;
} sos_resid;

/* 74-float RawResidualJacobian in the reference's field order (OB/RawResidualJacobian.h:29-55),
 * used to read a Jacobian back for inspection / parity checks. */
typedef struct sos_rawjac {
  float resF[8];
  float Jpdxi[2][6];
  float Jpdc[2][4];
  float Jpdd[2];
  float JIdx[2][8];
  float JabF[2][8];
  float JIdx2[4];                  /* (0,0) (0,1) (1,0) (1,1) */
  float JabJIdx[4];
  float Jab2[4];
} sos_rawjac;

typedef struct sos_ctx sos_ctx;          /* device + stream + frame (image pyramid) store */
typedef struct sos_ba sos_ba;            /* device backend of one EnergyFunctional */
typedef struct sos_tracker sos_tracker;  /* device side of one CoarseTracker / ScaleOptimizer */
typedef struct sos_comm sos_comm;        /* RCCL communicator of the multi-GPU path (one process per GPU) */

/* ------------------------------------------------------------------------------------------------
 * context and frame store
 * ---------------------------------------------------------------------------------------------- */

/* (handle lifetime: no reference counterpart)  Create a context on HIP device `device`.  `hip_stream` is a hipStream_t supplied by the caller or
 * NULL (the context then creates and owns one).  Must fail (SOS_ERR_HIP) when no GPU is present. */
int sos_ctx_create(int device, void *hip_stream, int w, int h, sos_ctx **out);
int sos_ctx_destroy(sos_ctx *ctx);
int sos_ctx_synchronize(sos_ctx *ctx);
/* number of pyramid levels for (w,h): util/globalCalib.cpp:39-52 */
int sos_ctx_pyr_levels(const sos_ctx *ctx);

/* Replaces FrameHessian::makeImages (FS/HessianBlocks.cpp:121-176; called at FS/FullSystem.cpp:650,
 * 1114).  `img` = w*h float irradiance on the host; `gammaBgrad` = 256-entry CalibHessian::B table
 * or NULL (setting_gammaWeightsPixelSelect != 1).  Builds all levels of dIp / absSquaredGrad in the
 * slot.  */
int sos_make_pyramid(sos_ctx *ctx, int slot, const float *img, const float *gammaB);
/* Upload a host-built level-0 dI (AoS (I,dx,dy), w*h*3 floats) into a slot (no pyramid). */
int sos_frame_upload_dI(sos_ctx *ctx, int slot, const float *dI_aos3);
/* Copy one pyramid level back: dI_out (wl*hl*3) and/or absgrad_out (wl*hl) may be NULL. */
int sos_frame_download_level(sos_ctx *ctx, int slot, int lvl, float *dI_out, float *absgrad_out);
int sos_frame_release(sos_ctx *ctx, int slot);

/* ------------------------------------------------------------------------------------------------
 * backend (EnergyFunctional device side)
 * ---------------------------------------------------------------------------------------------- */

/* new/delete EnergyFunctional: FS/FullSystem.cpp:83-84, 98-110 */
int sos_ba_create(sos_ctx *ctx, const sos_params *params, sos_ba **out);
int sos_ba_destroy(sos_ba *ba);

/* Packed snapshot of the window graph.  Replaces the state that insertFrame / insertPoint /
 * insertResidual / dropResidual / removePoint / marginalizeFrame / makeIDX leave behind
 * (OB/EnergyFunctional.cpp:644-728, 1186-1202); the graph mutation itself stays host-side.
 * frame_slot[i] = image slot of frame idx i.  res_toZeroF: R*8 floats (EFResidual::res_toZeroF) or
 * NULL when no residual is linearized.  lin_J: sos_rawjac per residual with SOS_RF_LINEARIZED (frozen
 * Jacobian of linearized residuals), indexed like res; may be NULL when none is linearized. */
int sos_ba_set_window(sos_ba *ba, int n, const int32_t *frame_slot, int P, const sos_point *pts,
                      int R, const sos_resid *res, const float *res_toZeroF,
                      const sos_rawjac *lin_J);

/* Per-step state: replaces FullSystem::setPrecalcValues (FS/FullSystem.cpp:1099-1107) ->
 * FrameFramePrecalc::set x n^2, EnergyFunctional::setDeltaF (OB/EnergyFunctional.cpp:163-194) and
 * setAdjointsF (:42-103).  precalc / adHTdeltaF / adHost / adTarget are indexed [h + n*t];
 * adHost/adTarget are 8x8 row-major doubles; point_* are P floats each (NULL = keep).  */
int sos_ba_set_state(sos_ba *ba, const sos_calib *calib, const sos_precalc *precalc,
                     const float *adHTdeltaF /* n*n*8 */, const float *cDeltaF /* 4 */,
                     const double *adHost /* n*n*64 */, const double *adTarget /* n*n*64 */,
                     const float *point_idepth_scaled, const float *point_idepth_zero_scaled,
                     const float *point_deltaF);

/* linearizeAll_Reductor -> PointFrameResidual::linearize over all active (non-linearized) residuals
 * (FS/FullSystemOptimize.cpp:44-77,125-143; FS/Residuals.cpp:77-271).  frameEnergyTH: n floats.
 * Outputs (each may be NULL), indexed like the `res` array of set_window:
 *   energySum        sum of the returned energies (stats[0]), accumulated in double in residual order
 *   newState         state_NewState (SOS_RES_*)
 *   newEnergy        state_NewEnergy
 *   newEnergyWithOutlier  state_NewEnergyWithOutlier (-1 when the residual went OOB)
 *   centerProjectedTo     R*3 floats
 * Linearized residuals are skipped (newState = their state_state, newEnergyWithOutlier = -1). */
int sos_ba_linearize(sos_ba *ba, const float *frameEnergyTH, double *energySum, uint8_t *newState,
                     float *newEnergy, float *newEnergyWithOutlier, float *centerProjectedTo);

/* applyRes_Reductor -> PointFrameResidual::applyRes(true) -> EFResidual::takeDataF
 * (FS/FullSystemOptimize.cpp:79-83,339-344,391-396; FS/Residuals.cpp:304-321;
 * OB/EnergyFunctionalStructs.cpp:36-45). */
int sos_ba_apply_res(sos_ba *ba);

/* PointFrameResidual::resetOOB over all non-linearized residuals (FS/Residuals.h:83-88,
 * FS/FullSystemOptimize.cpp:321-324). */
int sos_ba_reset_oob(sos_ba *ba);

/* The front of FullSystem::optimize with setting_forceAceptStep (FS/FullSystemOptimize.cpp:316-344): resetOOB of the
 * active residuals (resetOOB != 0), linearizeAll(false) and applyRes_Reductor as ONE launch chain; with sos_ba_set_prefetch
 * the first iteration's accumulate is enqueued behind it.  energySum = stats[0] of linearizeAll; newestEnergies / newestCount =
 * state_NewEnergyWithOutlier of the residuals that target the newest keyframe (what setNewFrameEnergyTH sorts, :91-99). */
int sos_ba_linearize_apply(sos_ba *ba, const float *frameEnergyTH, int resetOOB, double *energySum, float *newestEnergies,
                           int *newestCount);

/* What linearizeAll(true) leaves in a PointFrameResidual (FS/FullSystemOptimize.cpp:44-77, 148-179; FS/Residuals.cpp:304-321). */
typedef struct sos_resid_final {
  float state_NewEnergy, state_NewEnergyWithOutlier;
  float state_energy;              /* after applyRes(true) */
  float centerProjectedTo[3];      /* the last centre projection that succeeded */
  uint8_t state_NewState;          /* SOS_RES_* */
  uint8_t state_state;             /* after applyRes(true) */
  uint8_t active;                  /* efResidual->isActive() after applyRes(true): 0 = the residual goes to toRemove (:72-73) */
  uint8_t pad;                     /* 28 B */
} sos_resid_final;

/* FullSystem::linearizeAll(true) -- the final linearisation of optimize() (FS/FullSystemOptimize.cpp:425 -> 125-182):
 * linearize + applyRes(true) of every active residual, then in ONE device-to-host copy
 *   records              R records in the order of sos_ba_set_window (entries of linearized residuals are not written)
 *   pointMaxRelBaseline  per point the largest relBS over its active isNew residuals (:55-70), -1 when it has none
 *   pointNewGood         per point the number of those residuals (numGoodResiduals += ...)
 * The three arrays live in pinned memory owned by the handle and stay valid until the next call on it.
 * resetOOB != 0: PointFrameResidual::resetOOB first -- the relinearisation of flagPointsForRemoval (FS/FullSystem.cpp:575-583). */
int sos_ba_linearize_final(sos_ba *ba, const float *frameEnergyTH, int resetOOB, double *energySum, const sos_resid_final **records,
                           const float **pointMaxRelBaseline, const int32_t **pointNewGood, float *newestEnergies, int *newestCount);

/* EnergyFunctional::dropResidual (OB/EnergyFunctional.cpp:710-728) for residuals of the current snapshot: they stay in the
 * arrays but are dead from here on (no flags, state OOB, zero JpJdF / point terms), so that the snapshot serves the rest of the
 * keyframe (flagPointsForRemoval, marginalizePointsF) without a second sos_ba_set_window. */
int sos_ba_kill_residuals(sos_ba *ba, const int32_t *residIdx, int count);

/* EFResidual::fixLinearizationF for `count` residuals (FS/FullSystem.cpp:581;
 * OB/EnergyFunctionalStructs.cpp:75-103).  Uses the adHTdeltaF/cDeltaF/deltaF of the last set_state. */
int sos_ba_fix_linearization(sos_ba *ba, const int32_t *residIdx, int count);

/* accumulateAF_MT / accumulateLF_MT / accumulateSCF_MT incl. stitchDoubleMT
 * (OB/EnergyFunctional.cpp:197-254, 1040-1044; OB/AccumulatedTopHessian.cpp:35-147,231-301;
 * OB/AccumulatedSCHessian.cpp:32-158).  Each H is dense row-major (4+8n)^2 doubles, each b has 4+8n
 * doubles, caller-allocated.  H_L/b_L do NOT contain the priors (the host adds cPrior and the frame
 * priors, OB/AccumulatedTopHessian.cpp:292-300).  Any output pointer may be NULL. */
int sos_ba_accumulate(sos_ba *ba, double *H_A, double *b_A, double *H_L, double *b_L, double *H_sc,
                      double *b_sc, int *resInA, int *resInL);

/* Multi-GPU split of sos_ba_accumulate: (1) accumulate the local shard's fp32 blocks into a packed
 * device buffer, (2) the caller all-reduces that buffer (RCCL), (3) stitch from the reduced buffer.
 * sos_ba_acc_buffer returns the device pointer and float count of the packed buffer
 * [top_A n^2*91 | top_L n^2*91 | accD n^3*64 | accE n^2*32 | accEB n^2*8 | Hcc 16 | bc 4 | nresA nresL]. */
int sos_ba_accumulate_local(sos_ba *ba);
int sos_ba_acc_buffer(sos_ba *ba, float **dev_ptr, size_t *nfloats);
int sos_ba_stitch(sos_ba *ba, double *H_A, double *b_A, double *H_L, double *b_L, double *H_sc,
                  double *b_sc, int *resInA, int *resInL);

/* ---- fused per-iteration entry points (MI355X-first: two device round trips per Gauss-Newton iteration) ----
 * sos_ba_gn_accumulate = accumulateAF_MT + accumulateLF_MT + accumulateSCF_MT + stitch with
 * H_top = HL_top + HA_top, b_top = bL_top + bA_top (OB/EnergyFunctional.cpp:1040-1047, priors excluded).
 * Only the UPPER triangle (col >= row) of H_top and H_sc is written -- the matrices are symmetric and the
 * LDL^T solve reads one triangle; the stitch kernels store straight into device-mapped pinned memory. */
int sos_ba_gn_accumulate(sos_ba *ba, double *H_top, double *b_top, double *H_sc, double *b_sc, int *resInA,
                         int *resInL);
/* the enqueue half alone (no wait): host work that does not depend on H / b -- the IMU factors between accumulation and solve,
 * OB/EnergyFunctional.cpp:1053-1066 -- then overlaps the accumulation also when no sos_ba_gn_step prefetched it; the following
 * sos_ba_gn_accumulate only waits */
int sos_ba_gn_accumulate_begin(sos_ba *ba);
/* sos_ba_gn_step = resubstituteF_MT(x) (OB/EnergyFunctional.cpp:1182) + the point part of
 * doStepFromBackup on the device (idepth += stepfacD*step; idepth_zero = idepth,
 * FS/FullSystemOptimize.cpp:207-213) + setPrecalcValues/setDeltaF upload + linearizeAll(false)
 * (+ applyRes_Reductor when applyRes != 0, FS/FullSystemOptimize.cpp:371,391-396).  x == NULL skips the
 * back-substitution / point step.  Outputs (any may be NULL): energySum; newestEnergies = the
 * state_NewEnergyWithOutlier >= 0 of the residuals targeting the newest frame (capacity >= R floats),
 * what setNewFrameEnergyTH needs (FS/FullSystemOptimize.cpp:91-95); pointStep = PointHessian::step (P). */
int sos_ba_gn_step(sos_ba *ba, const double *x, float stepfacD, const sos_calib *calib, const sos_precalc *precalc,
                   const float *adHTdeltaF, const float *cDeltaF, const float *frameEnergyTH, int applyRes,
                   double *energySum, float *newestEnergies, int *newestCount, float *pointStep);
/* The back-substitution half of sos_ba_gn_step (resubstituteF_MT / resubstituteFPt, OB/EnergyFunctional.cpp:496-551), enqueued ahead: it needs x alone, so the caller issues it right after the
 * solve and computes the new poses / precalc records while it runs; the following sos_ba_gn_step must then be given
 * x = NULL.  SOS_ERR_STATE when the window cannot take this path (no points / no fp32 adjoints yet): pass x to
 * sos_ba_gn_step as before. */
int sos_ba_gn_resub(sos_ba *ba, const double *x, float stepfacD);

/* Pipelining switch of the fused calls: when on, sos_ba_gn_step(applyRes != 0) enqueues the accumulate + stitch of
 * the NEXT iteration right behind the linearisation (it depends on device state only) and returns as soon as the
 * linearisation results are on the host; the following sos_ba_gn_accumulate then only waits for it.  Any other
 * state-changing call in between discards the prefetched result.  Leave it off for the last iteration of a loop:
 * the per-point results (sos_ba_get_point_hessian) always belong to the latest accumulate that ran.
 * on = 2: nothing is enqueued, but the linearisation still reduces its Jacobian tiles to the 13x13 block sums of
 * AccumulatedTopHessianSSE::addPoint<0> on chip, and the next sos_ba_gn_accumulate (or the device-resident loop) uses them. */
int sos_ba_set_prefetch(sos_ba *ba, int on);

/* ---- device-resident Gauss-Newton loop: the loop body of FullSystem::optimize (FS/FullSystemOptimize.cpp:358-413, with
 * setting_forceAceptStep) with the HOST OUT OF THE ITERATION.  Besides the accumulate / linearisation kernels above, ONE single-workgroup
 * kernel per iteration does what solveSystemF does on the host (OB/EnergyFunctional.cpp:1046-1148: priors, HM / bM around delta, Schur
 * side, Jacobi scaling, LDL^T; IMU off), in fp64, and hands x to the step kernel of sos_ba_gn_devstep_begin through device memory
 * (resubstituteF_MT, doStepFromBackup, FrameHessian::setState, FrameFramePrecalc::set x n^2, setDeltaF); the order statistic of
 * setNewFrameEnergyTH (FS/FullSystemOptimize.cpp:84-124) is an exact selection on the device.  Frame states, calibration, FEJ poses
 * and the marginalisation prior stay on the device between begin and end; the host keeps stepping its own copies from x, as in the
 * device-side-step mode.
 *   begin    after the window has been packed, its state set (adjoints included) and linearised + applied once
 *            (FS/FullSystemOptimize.cpp:316-344): uploads what the loop needs (includes sos_ba_gn_devstep_begin)
 *   enqueue  one whole iteration as one chain of launches; returns at once with its sequence number
 *   wait     the results of iteration `seq`: they are published when its solve has run, BEFORE its back-substitution and
 *            linearisation, so the caller decides about the next iteration (canbreak) and enqueues it while this one finishes
 *   end      drains the stream, returns the point steps / inverse depths / newest-frame energies of the last iteration
 * Not available (SOS_ERR_STATE from begin; use the fused calls above): more than 17 keyframes, an empty window, a communicator. */
typedef struct sos_gn_frame {
  double camToWorld_evalPT[12];  /* FrameHessian::get_camToWorld_evalPT(): R row-major | t */
  double state[10];              /* FrameHessian::state */
  double state_zero[10];         /* FrameHessian::state_zero */
  double prior[8];               /* EFFrame::prior (FS/HessianBlocks.h:280-302) */
  float ab_exposure;
  int32_t pad;
} sos_gn_frame;
/* Device-side step of the fused loop (host solves, device does everything else): after sos_ba_gn_devstep_begin the frame states
 * (evaluation point, state, state_zero, exposure) and the calibration value stay on the device, and sos_ba_gn_step called with x,
 * frameEnergyTH and precalc == NULL (calib / adHTdeltaF / cDeltaF ignored) derives doStepFromBackup's new states, SE3::exp, the n^2
 * FrameFramePrecalc records and setDeltaF's arrays from x itself, inside the launch of the back-substitution: the host neither
 * stages them nor launches the stage-in.  The host keeps stepping its own copies of the states (same fp64 arithmetic; its SE3::exp
 * and the device's differ in the last bits, which is why the mode is explicit).  sos_ba_set_window / sos_ba_set_state end the mode.
 * SOS_ERR_STATE: more than 17 keyframes, an empty window, no fp32 adjoints. */
int sos_ba_gn_devstep_begin(sos_ba *ba, const sos_gn_frame *frames, const double *calib_value4, const double *calib_value_zero4);
int sos_ba_gn_devstep_end(sos_ba *ba);
int sos_ba_gn_resident_supported(sos_ba *ba);
int sos_ba_gn_resident_begin(sos_ba *ba, const sos_gn_frame *frames, const double *calib_value4, const double *calib_value_zero4,
                             double cPrior, const double *HM, const double *bM, const float *frameEnergyTH /* n */);
int sos_ba_gn_resident_enqueue(sos_ba *ba, int *seq_out);
/* header16: [0] seq, [1] failed (non-positive pivot: x is not usable), [6] sum |idepth_backup| and [7] the number of points
 * (doStepFromBackup's sumNID, FS/FullSystemOptimize.cpp:207-213; with a communicator attached: over the points of ALL ranks), [8] resInA, [9] resInL, [10] frameEnergyTH of the newest keyframe
 * used by this iteration's linearisation, [11..13] microseconds the solve kernel spent assembling / factorising / substituting;
 * x: 4 + 8 n (OB/EnergyFunctional.cpp:1148) */
int sos_ba_gn_resident_wait(sos_ba *ba, int seq, double *header16, double *x);
int sos_ba_gn_resident_end(sos_ba *ba, float *pointStep, float *idepth_scaled, double *energySum, float *newestEnergies,
                           int *newestCount);
/* The solve kernel of the device-resident loop on a system handed in from the host: EnergyFunctional::solveSystemF's visual part
 * (OB/EnergyFunctional.cpp:1046-1148) from its pieces -- H_top = HA_top + HL_top + priors and H_sc as UPPER triangles (dim x dim
 * row-major, dim = 4 + 8 n), b_top (priors' part included), b_sc, the marginalisation prior HM (whole) / bM, delta =
 * getStitchedDeltaF() -- with SOLVER_FIX_LAMBDA: x = S LDLT(S ((H_top + HM)(1 + lambda on the diagonal) - H_sc / (1 + lambda)) S)^-1
 * S (b_top + bM + HM delta - b_sc).  The same arithmetic as sosf_solve_system (include/sos_slam_host.h), which the parity tests hold
 * it against.  reps launches are timed: phase_us4 (may be NULL) = the kernel's own stamps (assemble, factorise, substitute) of the
 * last launch and the HIP-event wall time per launch.  SOS_ERR_STATE: a non-positive pivot (the caller solves on the host). */
int sos_gn_solve_system(sos_ctx *ctx, int n, const double *H_top, const double *b_top, const double *H_sc, const double *b_sc, const double *HM,
                        const double *bM, const double *delta, double *x, int reps, double *phase_us4);
/* Debug: the pivot-row update of the solve kernel's diagonal block (Eigen's LDLT inner loop, which OB/EnergyFunctional.cpp:1141-1148 calls,
 * as one v_fmac_f64_dpp ... row_newbcast:K per column in hand-written asm statements) against the same update through the compiler's
 * builtin, on ONE wave: lane l holds the row a_in[16 l .. 16 l + 15] and the multiplier nl[l]; for each pivot K = 0 .. 14 both forms
 * are applied to that input: out_asm / out_ref [(K * 64 + l) * 16 + j].  The two must agree bit for bit. */
int sos_dbg_gs_row_update(sos_ctx *ctx, const double *a_in, const double *nl, double *out_asm, double *out_ref);

/* ---- multi-GPU exchange; the reference has no counterpart: it is a single-process CPU backend ----
 * One process per GPU, every rank the same keyframes and its own shard of the points.  librccl is bound at run
 * time: sos_rccl_load(path) (NULL = "librccl.so" from the loader path; pass the copy the host process already uses).
 * Rank 0 obtains a 128-byte id (sos_rccl_unique_id), distributes it by any means, every rank calls sos_comm_create.
 * sos_ba_set_comm attaches the communicator to a backend: from then on the fused calls sum the packed fp32
 * accumulator blocks over all ranks (ONE all-reduce per Gauss-Newton iteration, enqueued on the library's stream
 * between the local accumulation and the stitch) and sos_ba_gn_step returns the newest-frame energies of ALL ranks
 * (one all-gather), so every rank derives the same frameEnergyTH and solves the same system.  All ranks must issue
 * the same sequence of fused calls.  comm == NULL detaches. */
int sos_rccl_load(const char *librccl_path);
int sos_rccl_unique_id(void *id128);
int sos_comm_create(const void *id128, int nranks, int rank, int device, sos_comm **out);
int sos_comm_destroy(sos_comm *comm);
int sos_comm_size(const sos_comm *comm);
int sos_comm_rank(const sos_comm *comm);
int sos_ba_set_comm(sos_ba *ba, sos_comm *comm);
/* number of floats sos_ba_gn_step may write to `newestEnergies` (all ranks' lists when a communicator is attached) */
int sos_ba_newest_capacity(sos_ba *ba, int *count);
/* all-gather of newest-frame energies for callers outside the fused calls (final linearizeAll(true)): `all` needs
 * sos_ba_newest_capacity floats; without a communicator it copies the local list */
int sos_ba_gather_energies(sos_ba *ba, const float *local, int count, float *all, int *total);
/* sum of a host fp64 buffer over all ranks, for the keyframe-rate exchanges (the shard-local M - Msc of
 * marginalizePointsF must be summed before it enters HM so that every rank keeps the same prior); no-op without a
 * communicator */
int sos_ba_allreduce_f64(sos_ba *ba, double *buf, size_t count);

/* per-point results of the accumulation: idepth_hessian (OB/AccumulatedSCHessian.cpp:50),
 * HdiF, bdSumF.  Each P floats, may be NULL. */
int sos_ba_get_point_hessian(sos_ba *ba, float *idepth_hessian, float *HdiF, float *bdSumF);

/* resubstituteF_MT / resubstituteFPt (OB/EnergyFunctional.cpp:496-551).  x has 4+8n doubles (the
 * solved increment, before sign flip); pointStep receives PointHessian::step for every point. */
int sos_ba_resubstitute(sos_ba *ba, const double *x, float *pointStep);

/* calcLEnergyF_MT without the frame / calib prior terms, which stay on the host
 * (OB/EnergyFunctional.cpp:563-642). */
int sos_ba_calc_lenergy(sos_ba *ba, double *E);

/* accumulation of marginalizePointsF (OB/EnergyFunctional.cpp:911-921): addPoint<2> + SC addPoint(p,
 * false) over `count` points, stitchDouble.  priorF of those points must already be scaled by
 * idepthFixPriorMargFac in the snapshot (OB/EnergyFunctional.cpp:901). */
int sos_ba_accumulate_marg(sos_ba *ba, const int32_t *pointIdx, int count, double *M, double *Mb,
                           double *Msc, double *Mbsc, int *resInM);

/* EFPoint::priorF edits between packs (p->priorF *= setting_idepthFixPriorMargFac,
 * OB/EnergyFunctional.cpp:901): overwrite priorF of `count` points of the snapshot. */
int sos_ba_update_point_priors(sos_ba *ba, const int32_t *pointIdx, const float *priorF, int count);

/* inspection helpers used by the parity tests (no reference counterpart: the reference reads these members directly --
 * EFResidual::J / JpJdF / res_toZeroF, OB/EnergyFunctionalStructs.h:70-73; PointFrameResidual::state_state, FS/Residuals.h:57) */
int sos_ba_get_jacobian(sos_ba *ba, int residIdx, int which /*0 = EFResidual::J, 1 = scratch*/,
                        sos_rawjac *out);
int sos_ba_get_residual_flags(sos_ba *ba, uint32_t *flags /*R*/, int32_t *state_state /*R*/,
                              float *state_energy /*R*/);
int sos_ba_get_JpJdF(sos_ba *ba, float *JpJdF /*R*8*/);
int sos_ba_get_res_toZeroF(sos_ba *ba, float *res_toZeroF /*R*8*/);

/* kernel timing on the context's stream with HIP events: runs `iters` back-to-back launches of the
 * named kernel ("linearize", "top_accumulate", "sc_accumulate", "apply_res", "resubstitute", ...) on
 * the current window state and returns the average milliseconds per launch. */
int sos_ba_time_kernel(sos_ba *ba, const char *kernel, const float *frameEnergyTH, int iters,
                       float *avg_ms);

/* ------------------------------------------------------------------------------------------------
 * coarse tracker / scale optimizer device side
 * ---------------------------------------------------------------------------------------------- */

/* new CoarseTracker(w,h,tfm_cam1_cam0,K1): FS/FullSystem.cpp:64-66; FS/ScaleOptimizer.cpp:38-87 */
int sos_tracker_create(sos_ctx *ctx, const sos_params *params, sos_tracker **out);
int sos_tracker_destroy(sos_tracker *trk);

/* makeK + setCoarseTrackingRef -> makeCoarseDepthL0 (FS/FullSystem.cpp:889-890;
 * FS/ScaleOptimizer.cpp:95-118; FS/CoarseTracker.cpp:56-242).  The npts points are those with
 * lastResiduals[0] IN: u,v = centerProjectedTo[0..1], idepth = centerProjectedTo[2],
 * hdi = EFPoint::HdiF.  refSlot = image slot of lastRef (its pyramid supplies pc_color).
 * pc_n_out: pyrLevels ints. */
int sos_tracker_set_ref(sos_tracker *trk, const sos_calib *calib, int refSlot, int npts,
                        const float *u, const float *v, const float *idepth, const float *hdi,
                        int32_t *pc_n_out);
/* Loop-closure aligner (N4): PoseEstimator::makeK + `pts = matched_frame->pts_dso`
 * (src/LoopClosure/PoseEstimator.cpp:129-148, 296-297).  The tracker object takes n 3-D points (xyz, AoS) of the
 * matched keyframe with one reference colour per pyramid level (colors[l * n + i]) as its template; afterwards
 * sos_tracker_calc_res runs PoseEstimator::calcRes (:128-286; the 3x3 argument is the plain rotation of refToNew)
 * and sos_tracker_calc_gs runs PoseEstimator::calcGSSSE (:75-126).  sos_tracker_set_ref switches back. */
int sos_tracker_set_points3d(sos_tracker *trk, const sos_calib *calib, int n, const float *xyz, const float *colors);

/* ScaleOptimizer::scaleCoarseDepthL0 (FS/CoarseTracker.cpp:244-251) */
int sos_tracker_scale_depth(sos_tracker *trk, float scale);
/* read back one level of the template point cloud (pc_u, pc_v, pc_idepth, pc_color) that makeCoarseDepthL0 builds
 * (FS/CoarseTracker.cpp:200-229, the normalisation loop); inspection helper of the parity tests */
int sos_tracker_get_pc(sos_tracker *trk, int lvl, float *pc_u, float *pc_v, float *pc_idepth,
                       float *pc_color);

/* CoarseTracker::calcResPose (FS/CoarseTracker.cpp:612-764).  RKi = R * Ki[lvl] (row-major 3x3
 * float), t float[3], affLL float[2] are computed by the host exactly as :628-634.  rs receives
 * [E, numTermsInE, flowT, 0, flowRT, satRatio]. */
int sos_tracker_calc_res(sos_tracker *trk, int lvl, int newSlot, const float *RKi, const float *t,
                         const float *affLL, float cutoffTH, double *rs /*6*/);
/* Latency hint for the LM loops: with on != 0 every sos_tracker_calc_res also runs calcGSSSEPose for the pose it
 * was given (a = affLL[0], b0 = the hint) behind itself, and every sos_tracker_calc_res_scale runs calcGSSSEScale;
 * the matching sos_tracker_calc_gs / _calc_gs_scale call is then answered without touching the device (an accepted
 * LM step costs one round trip instead of two).  Results are identical with or without the hint. */
int sos_tracker_set_gs_hint(sos_tracker *trk, int on, float b0);

/* CoarseTracker::calcGSSSEPose (FS/CoarseTracker.cpp:554-610) on the buffers of the last calc_res.
 * a = affLL[0] of the pose being linearized, b0 = lastRef_aff_g2l.b.  H 8x8 row-major, b 8. */
int sos_tracker_calc_gs(sos_tracker *trk, int lvl, float a, float b0, double *H, double *b);

/* CoarseTracker::trackNewestCoarse (FS/CoarseTracker.cpp:366-552) -- and, on a tracker in 3-D point mode
 * (sos_tracker_set_points3d), the LM loop of PoseEstimator::estimate (src/LoopClosure/PoseEstimator.cpp:288-495) -- for
 * nHyp initial poses in ONE launch: residual passes, accept / reject, the damped 8x8 solve, SE3::exp, level changes,
 * cutoff repeats and the minResForAbort test all run on the device; the host polls one flag per hypothesis.
 * Per hypothesis in: refToNew (row-major R | t) and aff_g2l; out: the same two, lastResiduals / lastInners per level,
 * lastFlowIndicators, `aborted` (the :527 test fired), the levels in the order they finished with their residual
 * (a repeated level appears twice) and the number of residual evaluations.  The caller applies the final affine sanity
 * tests of :538-551.  Ki = the caller's inverse intrinsics per level (9 floats each; identity for the loop aligner),
 * refAff = lastRef_aff_g2l (a, b).
 * The workgroups of the launch exchange their sums in flight, so all of them have to be resident; every wait is bounded (see
 * sos_tracker_set_lm_spin_limit).  SOS_ERR_TIMEOUT: a workgroup gave up, the launch drained, `hyp` holds no results -- the
 * tracker stays usable; retry, or run the loop around the primitives above (what the facade's CoarseTracker does). */
typedef struct sos_track_hyp {
  double refToNew[12];
  double aff[2];
  double lastResiduals[5];
  double flow[3];
  double visit_res[8];
  int32_t lastInners[5];
  int32_t visit_lvl[8];
  int32_t aborted, nvisits, evals;
} sos_track_hyp;
int sos_tracker_track(sos_tracker *trk, int newSlot, const float *Ki, float ref_ab_exposure, float new_ab_exposure,
                      const double *refAff, int coarsestLvl, const double *minResForAbort /*5*/, int nHyp,
                      sos_track_hyp *hyp);

/* ScaleOptimizer::optimizeScale (FS/ScaleOptimizer.cpp:120-230) as one launch of the same device loop, for nHyp initial scales
 * side by side (FullSystem::optimizeScale tries seven of them until the scale is trapped, FS/FullSystem.cpp:1135-1148): RKi =
 * rot(tfmF0ToF1) * Ki[lvl] per level (9 floats each), t = trans(tfmF0ToF1), K1 = (fx1, fy1, cx1, cy1) per level; scales[nHyp] is
 * in / out, lastResiduals (nHyp x 5, NaN for levels not visited; [5 k] is hypothesis k's return value of optimizeScale) and the
 * total number of residual evaluations are optional outputs. */
int sos_tracker_optimize_scale(sos_tracker *trk, int stereoSlot, const float *RKi, const float *t, const float *K1,
                               int coarsestLvl, int nHyp, float *scales, double *lastResiduals, int *evals);

/* Bound of every in-flight wait of the two loops above, in polling rounds (about a microsecond each; default 2^18).  A device that
 * runs other streams beside the tracker's may keep some workgroups of a loop from becoming resident for a while; the loop then
 * waits up to this bound and returns SOS_ERR_TIMEOUT instead of hanging.  0 makes every wait that is not satisfied at once give up
 * (tests use it to drive the fallback). */
int sos_tracker_set_lm_spin_limit(sos_tracker *trk, unsigned rounds);
/* Phase times of the last one-launch loop (sos_tracker_track / _optimize_scale / the loop aligner), hypothesis `hyp`, in microseconds
 * from the kernel's own 100 MHz stamps: us7 = [kernel, residual passes (calcRes + calcGSSSE products, FS/CoarseTracker.cpp:612-764,
 * 554-610), all-gather of the per-chunk sums, bookkeeping between two evaluations (:432-510) -- of which the damped 8 x 8 solve and
 * SE3::exp + the next request --, number of residual evaluations].  A profiling aid: nothing else reads it. */
int sos_tracker_lm_profile(sos_tracker *T, int hyp, double *us7);

/* ScaleOptimizer::calcResScale / calcGSSSEScale (FS/ScaleOptimizer.cpp:273-437, 232-271).
 * RKi = rot(tfmF0ToF1) * Ki[lvl], t = trans(tfmF0ToF1); K1 = (fx1,fy1,cx1,cy1) of level `lvl`. */
int sos_tracker_calc_res_scale(sos_tracker *trk, int lvl, int stereoSlot, const float *RKi,
                               const float *t, const float *K1, float scale, float cutoffTH,
                               double *rs /*6*/);
int sos_tracker_calc_gs_scale(sos_tracker *trk, int lvl, const float *t, const float *K1,
                              float scale, float *H, float *b);

/* library identification: returns "hip-gfx950" */
/* ------------------------------------------------------------------------------------------------
 * immature points (first part): construction and epipolar tracing
 * ---------------------------------------------------------------------------------------------- */

/* ImmaturePointStatus (FS/ImmaturePoint.h:40-47) */
#define SOS_IPS_GOOD 0
#define SOS_IPS_OOB 1
#define SOS_IPS_OUTLIER 2
#define SOS_IPS_SKIPPED 3
#define SOS_IPS_BADCONDITION 4
#define SOS_IPS_UNINITIALIZED 5

/* the globals traceOn / the constructor read (util/settings.cpp:82-83,118,128-143) */
typedef struct sos_trace_params {
  float maxPixSearch;           /* 0.027 */
  float stepsize;               /* setting_trace_stepsize = 1 */
  float GNThreshold;            /* 0.1 */
  float extraSlackOnTH;         /* 1.2 */
  float slackInterval;          /* 1.5 */
  float minImprovementFactor;   /* 2 */
  float huberTH;                /* 9 */
  float outlierTHSumComponent;  /* 50*50 */
  float outlierTH;              /* 12*12 */
  float overallEnergyTHWeight;  /* 1 */
  int32_t GNIterations;         /* 3 */
  int32_t minTraceTestRadius;   /* 2 */
} sos_trace_params;

/* the fields of ImmaturePoint the tracing reads and writes (FS/ImmaturePoint.h:54-100); 128 bytes */
typedef struct sos_immature {
  float u, v;
  float idepth_min, idepth_max;
  float color[SOS_PATTERN_NUM];
  float weights[SOS_PATTERN_NUM];
  float gradH[4];                /* (0,0) (0,1) (1,0) (1,1) */
  float energyTH;
  float quality;
  float lastTraceUV[2];
  float lastTracePixelInterval;
  int32_t lastTraceStatus;       /* SOS_IPS_* */
  int32_t pad[2];
} sos_immature;

/* new ImmaturePoint(u, v, host, ...) for `count` pixel positions of the frame in image slot hostSlot
 * (FS/ImmaturePoint.cpp:30-59; called from FullSystem::makeNewTraces, FS/FullSystem.cpp:1080-1097) */
int sos_immature_init(sos_ctx *ctx, const sos_trace_params *prm, int hostSlot, int count, const int32_t *u,
                      const int32_t *v, sos_immature *out);
/* ph->traceOn(fh, KRKi, Kt, aff, ...) over the `count` immature points of one host frame against the frame in
 * frameSlot: the inner loop of FullSystem::traceNewCoarse (FS/FullSystem.cpp:323-350; FS/ImmaturePoint.cpp:70-415).
 * KRKi (3x3 row-major), Kt, aff are computed by the caller exactly as FS/FullSystem.cpp:326-332.  pts is in/out. */
int sos_immature_trace(sos_ctx *ctx, const sos_trace_params *prm, int frameSlot, int count, sos_immature *pts,
                       const float *KRKi, const float *Kt, const float *aff);
/* the whole of FullSystem::traceNewCoarse (both loops of FS/FullSystem.cpp:323-350) in one launch: point i belongs to
 * keyframe hostOfPoint[i] (0..nhosts-1); KRKi / Kt / aff hold nhosts entries of 9 / 3 / 2 floats */
int sos_immature_trace_all(sos_ctx *ctx, const sos_trace_params *prm, int frameSlot, int count, sos_immature *pts,
                           const int32_t *hostOfPoint, int nhosts, const float *KRKi, const float *Kt, const float *aff);

/* Device-resident form of the same loop.  FullSystem::traceNewCoarse runs on EVERY frame, and between two keyframes nothing on the
 * host reads what it writes (idepth_min / idepth_max / quality / lastTrace*: their readers are activatePointsMT and the
 * marginalisation flags of makeKeyFrame, FS/FullSystem.cpp:375-531, 783-931) -- so the records of a keyframe's immature points can
 * stay on the device from makeNewTraces until the next keyframe decision instead of crossing the bus twice per frame.
 *   put     the immature points of host keyframe `hostKey` (records as sos_immature_init / the array calls produce them);
 *           replaces an earlier list of that key, count = 0 removes it (keyframe marginalised, points activated / deleted)
 *   trace   traceNewCoarse against the frame in frameSlot: every point of the listed keys, in place, ONE launch, nothing
 *           copied and nothing waited for; KRKi / Kt / aff hold one entry per listed key (9 / 3 / 2 floats), as for trace_all.
 *           Keys without points are skipped; SOS_ERR_ARG for a key listed twice or more than SOS_MAX_FRAMES keys
 *   get     the records of a key back to the host (count must equal the stored count); count() reports it (0 if absent)
 * Records come out bit-identical to the same sequence of sos_immature_trace_all calls. */
typedef struct sos_immset sos_immset;
int sos_immset_create(sos_ctx *ctx, sos_immset **out);
void sos_immset_destroy(sos_immset *set);
int sos_immset_put(sos_immset *set, int hostKey, int count, const sos_immature *pts);
int sos_immset_trace(sos_immset *set, const sos_trace_params *prm, int frameSlot, int nhosts, const int32_t *hostKeys,
                     const float *KRKi, const float *Kt, const float *aff);
int sos_immset_count(sos_immset *set, int hostKey, int *count);
int sos_immset_get(sos_immset *set, int hostKey, int count, sos_immature *out);

/* ---- point activation: FullSystem::optimizeImmaturePoint over the candidates chosen by activatePointsMT ------------
 * (FS/FullSystemOptPoint.cpp:47-192 with ImmaturePoint::linearizeResidual, FS/ImmaturePoint.cpp:475-545; the loop
 * FS/FullSystem.cpp:365-374,476-487).  A 1-D Levenberg-Marquardt on the inverse depth of every candidate against
 * all other keyframes of the window. */
typedef struct sos_activate_params {
  float huberTH;                /* setting_huberTH = 9 */
  float minIdepthH_act;         /* setting_minIdepthH_act = 100 */
  int32_t GNIts;                /* setting_GNItsOnPointActivation = 3 */
  int32_t minObs;               /* the minObs argument (1 from activatePointsMT_Reductor) */
} sos_activate_params;

/* what linearizeResidual reads of host->targetPrecalc[target] (FS/HessianBlocks.h:109-134): the CURRENT-state
 * PRE_RTll / PRE_tTll (not the _0 pair of sos_precalc) and PRE_aff_mode; 16 floats */
typedef struct sos_pair_tfm {
  float R[9];                   /* PRE_RTll, row-major */
  float t[3];                   /* PRE_tTll */
  float aff[2];                 /* PRE_aff_mode */
  float pad[2];
} sos_pair_tfm;

#define SOS_ACT_SKIP 0          /* returned 0: not well constrained, the point stays immature */
#define SOS_ACT_DELETE (-1)     /* returned (PointHessian*)-1: outlier / non-finite, the candidate is deleted */
#define SOS_ACT_ACTIVATED 1     /* a PointHessian is created with idepth, residuals to the frames of inMask */
typedef struct sos_activation {
  int32_t status;               /* SOS_ACT_* */
  float idepth;                 /* currentIdepth at exit (setIdepthZero / setIdepth, FS/FullSystemOptPoint.cpp:158-159) */
  uint32_t inMask;              /* bit t set: the residual towards frame idx t ended ResState::IN (:162-168) */
  float energy, Hdd, bd;        /* lastEnergy, lastHdd, lastbd at exit (inspection / parity) */
  int32_t iterations;           /* LM iterations run */
  int32_t pad;
} sos_activation;

/* FullSystem::optimizeImmaturePoint (FS/FullSystemOptPoint.cpp:47-192; called from activatePointsMT_Reductor,
 * FS/FullSystem.cpp:311-325) for `count` candidates.  nFrames keyframes, frame idx f has its images in slot
 * frameSlot[f]; pairs[host + nFrames * target]; hostOfPoint[i] = frame idx of the host of pts[i].  pts is read only
 * (u, v, color, weights, energyTH, idepth_min, idepth_max). */
int sos_immature_activate(sos_ctx *ctx, const sos_activate_params *prm, const sos_calib *calib, int nFrames,
                          const int32_t *frameSlot, const sos_pair_tfm *pairs, int count, const sos_immature *pts,
                          const int32_t *hostOfPoint, sos_activation *out);

/* ---- pixel selection in front of the immature points: PixelSelector (FS/PixelSelector2.cpp) ----------------------------
 * makeHists :69-155 (per 32x32 cell gradient histogram -> threshold, 3x3 smoothing), select :292-424 (one pixel per
 * pot / 2 pot / 4 pot cell on three pyramid levels, direction chosen from a random pattern indexed by the running
 * count of level-0 selections), makeMaps :157-290 (re-selection with another potential, random sub-sampling). */
typedef struct sos_pixsel_params {
  float minGradHistCut;             /* setting_minGradHistCut = 0.5 */
  float minGradHistAdd;             /* setting_minGradHistAdd = 7 */
  float gradDownweightPerLevel;     /* setting_gradDownweightPerLevel = 0.75 */
  int32_t selectDirectionDistribution; /* setting_selectDirectionDistribution = true */
} sos_pixsel_params;
typedef struct sos_pixsel sos_pixsel;
/* new PixelSelector(w, h): randomPattern = w*h bytes, rand() & 0xFF after srand(3141592) (:37-40), supplied by the
 * caller */
int sos_pixsel_create(sos_ctx *ctx, const sos_pixsel_params *prm, const uint8_t *randomPattern, sos_pixsel **out);
int sos_pixsel_destroy(sos_pixsel *ps);
/* PixelSelector::makeHists (FS/PixelSelector2.cpp:69-155) of the frame in `slot` (pyramid made by sos_make_pyramid); ths / thsSmoothed: (w/32)*(h/32) floats, optional */
int sos_pixsel_make_hists(sos_pixsel *ps, int slot, float *ths, float *thsSmoothed);
/* PixelSelector::select (FS/PixelSelector2.cpp:292-424): select(fh, map_out, pot, thFactor) after make_hists of the same slot; map_out: w*h floats (0, 1, 2, 4), optional;
 * n: counts of level-0 / 1 / 2 selections */
int sos_pixsel_select(sos_pixsel *ps, int slot, int pot, float thFactor, float *map_out, int32_t n[3]);
/* PixelSelector::makeMaps (FS/PixelSelector2.cpp:157-290): makeMaps(fh, map_out, density, recursionsLeft, false, thFactor): runs make_hists when the slot changed, select, the
 * re-selection recursion and the sub-sampling; *currentPotential in / out; *numSelected = the return value (numHaveSub).
 * map_out optional. */
int sos_pixsel_make_maps(sos_pixsel *ps, int slot, float density, int recursionsLeft, float thFactor, int32_t *currentPotential,
                         float *map_out, int32_t *numSelected);
/* the loop of FullSystem::makeNewTraces (FS/FullSystem.cpp:1083-1095) over the last map: selected pixels with
 * padding + 1 <= x < w - padding - 2 (same for y) in row-major order; capacity entries at most, *count = all of them */
int sos_pixsel_list(sos_pixsel *ps, int patternPadding, int capacity, int32_t *u, int32_t *v, float *type, int32_t *count);

/* ---- image front-end: Undistort / PhotometricUndistorter (U/Undistort.cpp) --------------------------------------------
 * The DSO camera file (4 lines: model + parameters, input size, "crop" | "none" | fx fy cx cy 0, output size;
 * getUndistorterForFile :240-351, readFromFile :679-890), the rectified camera matrix (makeOptimalK_crop :557-672),
 * the remap table (distortCoordinates of the five models :902-1126) and per frame the photometric correction
 * (processFrame :194-227) + bilinear remap (undistort :361-458) feeding the pyramid.  The table is built once on the
 * host (libm transcendental functions, as the reference); the per-frame work runs on the device. */
#define SOS_CAM_RADTAN 0
#define SOS_CAM_PINHOLE 1
#define SOS_CAM_EQUIDISTANT 2
#define SOS_CAM_KB 3
#define SOS_CAM_FOV 4
#define SOS_RECT_CROP (-1)
#define SOS_RECT_NONE (-3)
#define SOS_RECT_GIVEN 0
typedef struct sos_camera_model {
  int32_t model;            /* SOS_CAM_* */
  int32_t rect;             /* SOS_RECT_* */
  double pars[8];           /* parsOrg after the relative-format rescale (fx fy cx cy + distortion) */
  int32_t wOrg, hOrg, w, h; /* input and output size */
  float outCal[5];          /* line 3 when SOS_RECT_GIVEN (relative fx fy cx cy, 0) */
  int32_t pad;
} sos_camera_model;
/* Undistort::getUndistorterForFile + readFromFile (U/Undistort.cpp:240-351, 679-890) on the text of a camera file; returns SOS_ERR_ARG on the formats the reference rejects ("full" is not
 * implemented there either) */
int sos_camera_parse(const char *text, sos_camera_model *out);

typedef struct sos_undistort sos_undistort;
/* G: inverse response, GDepth >= 256 strictly increasing entries as read from pcalib.txt (normalised here as
 * :84-92), or NULL; vignette: wOrg*hOrg values of the vignette image (any scale; divided by their maximum as :113-141),
 * or NULL.  photometricMode = setting_photometricCalibration (0 none, 1 response only, 2 response + vignette).  The
 * context must have been created with the model's output size. */
int sos_undistort_create(sos_ctx *ctx, const sos_camera_model *cam, const float *G, int GDepth, const float *vignette,
                         int photometricMode, sos_undistort **out);
int sos_undistort_destroy(sos_undistort *u);
/* Undistort::getK / the remap table built by readFromFile (U/Undistort.cpp:557-672 makeOptimalK_crop, :836-890):
 * K = rectified fx fy cx cy; remapX / remapY: w*h floats, optional */
int sos_undistort_get(sos_undistort *u, float K[4], float *remapX, float *remapY, int32_t *passthrough);
/* Undistort::undistort<T>(image_raw, exposure, timestamp, factor) (U/Undistort.cpp:361-458, photometric part
 * processFrame :194-227) followed by FrameHessian::makeImages into `slot`
 * (bytesPerPixel 1 or 2); image_out (optional, w*h floats) receives the undistorted irradiance image. */
int sos_undistort_frame(sos_undistort *u, const void *raw, int bytesPerPixel, float exposure, float factor, int slot,
                        const float *gammaBgrad, float *image_out);

const char *sos_backend_name(void);

#ifdef __cplusplus
}
#endif
#endif /* SOS_SLAM_H */
