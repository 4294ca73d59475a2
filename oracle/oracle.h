/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE.  CPU restatement ("oracle") of the reference's hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (sos_slam_amd/) never links, imports or executes anything under oracle/.
 *
 * PARITY UNPINNED: the reference ships no golden vector / known-answer test for this path
 *  and cannot be compiled here (Eigen, Boost, OpenCV, ROS absent), so this
 * restatement is pinned only by (a) the Sophus property tests restated in tests/test_oracle_math.py
 * with the vectors of thirdparty/Sophus/sophus/test_se3.cpp:38-92 and (b) self-consistency
 * known-answer tests (finite-difference Jacobians, dense Schur complement, dense back-substitution).
 *
 * Arithmetic convention (shared with the HIP kernels): fp32, no FMA contraction
 * (-ffp-contract=off), sums evaluated left-to-right in the order the reference's source writes them.
 *
 * The record types (sos_params, sos_point, ...) come from include/sos_slam.h so tests can hand the
 * same numpy buffers to the oracle and to the HIP library.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/sos_slam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_window orc_window;

/* ---- backend, kernel level (one function per reference function) --------------------------------*/
orc_window *orc_window_create(const sos_params *prm, int n, int P, const sos_point *pts, int R,
                              const sos_resid *res);
void orc_window_destroy(orc_window *W);
void orc_set_image(orc_window *W, int frame, const float *dI_aos3); /* pointer is kept, not copied */
void orc_set_lin(orc_window *W, const float *res_toZeroF, const sos_rawjac *linJ);
void orc_set_state(orc_window *W, const sos_calib *calib, const sos_precalc *precalc,
                   const float *adHTdeltaF, const float *cDeltaF, const double *adHost,
                   const double *adTarget, const float *idepth_scaled,
                   const float *idepth_zero_scaled, const float *deltaF);
/* FS/FullSystemOptimize.cpp:44-77,125-143 + FS/Residuals.cpp:77-271; nthreads<=1: sequential */
double orc_linearize_all(orc_window *W, const float *frameEnergyTH, int nthreads);
void orc_apply_res(orc_window *W);                 /* FS/Residuals.cpp:304-321 */
void orc_reset_oob(orc_window *W);                 /* FS/Residuals.h:83-88 */
void orc_fix_linearization(orc_window *W, const int32_t *idx, int count);
/* OB/EnergyFunctional.cpp:197-254 + stitch; mode: 0 = reference fp32 tiered accumulators
 * (single-thread order), 1 = fp64 "truth" accumulation. nthreads>1: 6-thread-style pool (timing). */
void orc_accumulate(orc_window *W, double *H_A, double *b_A, double *H_L, double *b_L, double *H_sc,
                    double *b_sc, int *resInA, int *resInL, int fp64_truth, int nthreads);
void orc_resubstitute(orc_window *W, const double *x, float *pointStep, int nthreads);
double orc_calc_lenergy(orc_window *W);
void orc_accumulate_marg(orc_window *W, const int32_t *pointIdx, int count, double *M, double *Mb,
                         double *Msc, double *Mbsc, int *resInM);
/* raw views of internal arrays */
sos_rawjac *orc_J(orc_window *W);        /* EFResidual::J, R records */
sos_rawjac *orc_Jnew(orc_window *W);     /* PointFrameResidual::J (scratch), R records */
sos_resid *orc_res(orc_window *W);
sos_point *orc_pts(orc_window *W);
int32_t *orc_new_state(orc_window *W);
float *orc_new_energy(orc_window *W);
float *orc_new_energy_wo(orc_window *W);
float *orc_center(orc_window *W);        /* R*3 */
float *orc_JpJdF(orc_window *W);         /* R*8 */
float *orc_res_toZeroF(orc_window *W);   /* R*8 */
float *orc_point_field(orc_window *W, int which); /* 0 idepth_hessian 1 HdiF 2 bdSumF 3 Hdd_accAF
                                                     4 bd_accAF 5 Hcd_accAF(4P) 6 Hdd_accLF 7 bd_accLF
                                                     8 Hcd_accLF(4P) 9 step */

/* ---- backend, host level (GN loop: FS/FullSystemOptimize.cpp:305-489) ---------------------------*/
typedef struct orc_frame_init {
  double camToWorld[12];   /* evalPT: R row-major (9) + t (3) */
  double state[10];        /* FrameHessian::state; state_zero := state with [0..5] = 0 */
  float ab_exposure;
  int32_t frameID;
  float frameEnergyTH;
  int32_t pad;
} orc_frame_init;

void orc_host_init(orc_window *W, const orc_frame_init *frames, const double *calib_value_scaled,
                   const double *HM, const double *bM);
/* rolling windows: explicit FEJ point per frame (evalPT pose + state_zero) and calibration value / value_zero */
typedef struct orc_frame_init_ex {
  double camToWorld[12];   /* evalPT */
  double state[10];
  double state_zero[10];
  float ab_exposure;
  int32_t frameID;
  float frameEnergyTH;
  int32_t pad;
} orc_frame_init_ex;
void orc_host_init_ex(orc_window *W, const orc_frame_init_ex *frames, const double *calib_value, const double *calib_value_zero,
                      const double *HM, const double *bM);
void orc_host_get_calib_value(orc_window *W, double *value4, double *value_zero4);
void orc_host_get_evalpt(orc_window *W, int frame, double *camToWorld_evalPT12);
int32_t *orc_num_good_residuals(orc_window *W);
/* optimize() with setting_forceAceptStep selectable (FS/FullSystemOptimize.cpp:387-413: loadSateBackup on a rejected step) */
float orc_optimize_ex(orc_window *W, int mnumOptIts, int nthreads, int forceAccept, int *iterations_out, int *rejected_out);
/* switches solveSystemF of the host loop to the IMU branch (OB/EnergyFunctional.cpp:1053-1171 via orc_imu_solve): S, C, frames are
 * sosf_imu_settings / sosf_imu_calib / sosf_imu_frame[n] owned by the caller (poses refreshed before every solve, scale and
 * state_imu stepped after it with unit step factors), HM / bM the prior in the expanded dimension.  S = NULL switches back. */
void orc_host_set_imu(orc_window *W, const void *S, void *C, void *frames, const double *HM, const double *bM);
/* runs optimize(mnumOptIts); returns RMSE; fills iteration count */
float orc_optimize(orc_window *W, int mnumOptIts, int nthreads, int *iterations_out);
/* one loop body of FS/FullSystemOptimize.cpp:358-413 (used by the CPU-baseline timing) */
int orc_gn_iteration(orc_window *W, int iteration, int nthreads);
void orc_host_get_frame(orc_window *W, int frame, double *camToWorld12, double *state10,
                        double *state_zero10, float *frameEnergyTH);
void orc_host_get_calib(orc_window *W, double *value_scaled4);
/* yardstick only: accumulate H/b of the GN loop in fp64 instead of the reference's tiered fp32 */
void orc_host_set_truth_mode(orc_window *W, int on);
void orc_host_precalc(orc_window *W); /* setPrecalcValues: FS/FullSystem.cpp:1099-1107 */
const sos_precalc *orc_host_get_precalc(orc_window *W);
const float *orc_host_get_adHTdeltaF(orc_window *W);
const double *orc_host_get_adHost(orc_window *W);
const double *orc_host_get_adTarget(orc_window *W);
const double *orc_host_get_lastX(orc_window *W);
void orc_host_get_HM(orc_window *W, double *HM, double *bM);
/* EFFrame::prior / delta_prior (OB/EnergyFunctionalStructs.cpp:105-130) */
void orc_host_get_frame_prior(orc_window *W, int frame, double *prior8, double *delta_prior8);
/* keyframe-rate marginalisation: flagPointsForRemoval (explicit list) + dropPointsF + marginalizePointsF
 * (FS/FullSystem.cpp:573-596, 909-912; OB/EnergyFunctional.cpp:891-952) and the prior part of
 * marginalizeFrame (OB/EnergyFunctional.cpp:730-889, IMU off) */
void orc_host_drop_points(orc_window *W, const int32_t *pointIdx, int count);
void orc_host_marginalize_points(orc_window *W, const int32_t *pointIdx, int count, int32_t *marg_flag_out);
void orc_host_marginalize_frame_prior(orc_window *W, int frameIdx, double *HM_out /*(dim-8)^2*/, double *bM_out);

/* ---- stateless helpers ---------------------------------------------------------------------------*/
/* FrameFramePrecalc::set (FS/HessianBlocks.cpp:431-461) for one ordered pair */
void orc_precalc_pair(const double *hostEval12, const double *targetEval12, const double *hostPRE12,
                      const double *targetPRE12, const sos_calib *calib, float host_ab_exposure,
                      float target_ab_exposure, const double *host_aff_g2l /*a,b scaled*/,
                      const double *target_aff_g2l, double host_b0, sos_precalc *out,
                      float *distanceLL);
void orc_se3_exp12(const double *tangent6, double *T12);
void orc_se3_log12(const double *T12, double *tangent6);
void orc_se3_adj12(const double *T12, double *Ad36);
void orc_se3_mul12(const double *A12, const double *B12, double *C12);
void orc_se3_inv12(const double *A12, double *C12);
int orc_solve_ldlt(const double *A, const double *b, double *x, int n);

/* ---- tracker (FS/HessianBlocks.cpp:121-176, FS/CoarseTracker.cpp, FS/ScaleOptimizer.cpp) --------*/
int orc_pyr_levels(int w, int h);
/* makeImages: out_dI[lvl] (wl*hl*3) and out_abs[lvl] (wl*hl) are caller-allocated per level */
void orc_make_images(const float *img, int w, int h, const float *gammaB, int levels, float **out_dI,
                     float **out_abs);

typedef struct orc_tracker orc_tracker;
orc_tracker *orc_tracker_create(const sos_params *prm, int w, int h);
void orc_tracker_destroy(orc_tracker *T);
/* makeK + makeCoarseDepthL0; ref_dI[lvl] are the pyramid levels of lastRef */
void orc_tracker_set_ref(orc_tracker *T, const sos_calib *calib, float *const *ref_dI, int npts,
                         const float *u, const float *v, const float *idepth, const float *hdi,
                         int32_t *pc_n_out);
void orc_tracker_get_pc(orc_tracker *T, int lvl, float *pc_u, float *pc_v, float *pc_idepth,
                        float *pc_color);
void orc_tracker_scale_depth(orc_tracker *T, float scale);
void orc_tracker_calc_res(orc_tracker *T, int lvl, const float *new_dI, const float *RKi,
                          const float *t, const float *affLL, float cutoffTH, double *rs);
void orc_tracker_calc_gs(orc_tracker *T, int lvl, float a, float b0, double *H, double *b);
void orc_tracker_calc_res_scale(orc_tracker *T, int lvl, const float *stereo_dI, const float *RKi,
                                const float *t, const float *K1, float scale, float cutoffTH,
                                double *rs);
void orc_tracker_calc_gs_scale(orc_tracker *T, int lvl, const float *t, const float *K1, float scale,
                               float *H, float *b);
int orc_tracker_warp_n(orc_tracker *T);
/* yardstick only: fp64 accumulation of the calcRes / calcGSSSE sums */
void orc_tracker_set_truth_mode(orc_tracker *T, int on);
/* loop-closure aligner: PoseEstimator's template (3-D points + per-level colours); calc_res then follows
 * src/LoopClosure/PoseEstimator.cpp:128-286, and orc_tracker_track with zero reference affine parameters and no abort
 * thresholds is PoseEstimator::estimate :288-495 up to its three acceptance tests (orc_tracker_last_inners for the third) */
void orc_tracker_set_points3d(orc_tracker *T, const sos_calib *calib, int n, const float *xyz, const float *colors);
void orc_tracker_last_inners(orc_tracker *T, int *out);
/* whole LM loop, FS/CoarseTracker.cpp:366-552. lastToNew12 and aff2 are in/out. Returns 1 = ok */
int orc_tracker_track(orc_tracker *T, float *const *new_dI, float ref_ab_exposure,
                      float new_ab_exposure, const double *ref_aff_g2l, double *lastToNew12,
                      double *aff2, int coarsestLvl, const double *minResForAbort5,
                      double *lastResiduals5, double *flow3);
/* FS/ScaleOptimizer.cpp:120-230 */
float orc_tracker_optimize_scale(orc_tracker *T, float *const *stereo_dI, const double *tfmF0ToF1_12,
                                 const float *K1_level0, float *scale_inout, int coarsestLvl);

#ifdef __cplusplus
}
#endif
/* ---- immature points (orc_immature.c): ImmaturePoint constructor and traceOn ------------------------------------ */
void orc_immature_init(const sos_trace_params *prm, const float *host_dI_aos3, int w, int h, int count, const int32_t *u,
                       const int32_t *v, sos_immature *out);
void orc_immature_trace(const sos_trace_params *prm, const float *frame_dI_aos3, int w, int h, int count, sos_immature *pts,
                        const float *KRKi, const float *Kt, const float *aff);
/* FullSystem::optimizeImmaturePoint over `count` candidates (FS/FullSystemOptPoint.cpp:47-192); dI[f] = level-0 image of frame idx f */
void orc_immature_activate(const sos_activate_params *prm, const sos_calib *calib, int w, int h, int n, const float *const *dI,
                           const sos_pair_tfm *pairs, int count, const sos_immature *pts, const int32_t *hostOf,
                           sos_activation *out);
/* candidate selection of FullSystem::activatePointsMT with CoarseDistanceMap (FS/FullSystem.cpp:375-470,
 * FS/CoarseTracker.cpp:793-925); same arguments as sosf_activate_select */
float orc_next_min_act_dist(float currentMinActDist, int nPoints, float desiredPointDensity);
int orc_activate_select(int w1, int h1, int nFrames, int newest, const float *KRKi, const float *Kt, int nActive,
                        const float *act_u, const float *act_v, const float *act_idepth_scaled, const int32_t *act_host,
                        float currentMinActDist, float minTraceQuality, int nCand, const sos_immature *cand,
                        const int32_t *cand_host, const float *cand_type, const uint8_t *hostFlagged, int8_t *decision,
                        float *distFinal);

/* ---- pixel selection (orc_pixsel.c): PixelSelector::makeHists / select / makeMaps, FS/PixelSelector2.cpp ------------ */
void orc_pixsel_make_hists(const sos_pixsel_params *prm, const float *absg0, int w, int h, float *ths, float *thsSmoothed);
void orc_pixsel_select(const sos_pixsel_params *prm, const float *dI, const float *absg0, const float *absg1, const float *absg2,
                       int w, int h, const uint8_t *randomPattern, const float *thsSmoothed, int pot, float thFactor,
                       float *map_out, int32_t n_out[3]);
int orc_pixsel_make_maps(const sos_pixsel_params *prm, const float *dI, const float *absg0, const float *absg1, const float *absg2,
                         int w, int h, const uint8_t *randomPattern, const float *thsSmoothed, float density, int recursionsLeft,
                         float thFactor, int *pot, float *map_out);

/* ---- image front-end (orc_undistort.c): camera file, rectified K + remap table, photometric + geometric undistortion -- */
int orc_camera_parse(const char *text, sos_camera_model *out);
int orc_undistort_setup(const sos_camera_model *cam, double K[4], float *remapX, float *remapY, int *passthrough);
int orc_photometric_setup(float *G, int GDepth, const float *vignette, int n, int photometricMode, float *vignetteInv);
void orc_undistort_frame(const sos_camera_model *cam, const float *remapX, const float *remapY, int passthrough, const float *G,
                         int valid, const float *vignetteInv, int photometricMode, const void *raw, int bpp, float exposure,
                         float factor, float *out);

/* ---- IMU / spline factor assembly (orc_imu.c); the record types live in include/sos_slam_host.h --------------------- */
struct sosf_imu_settings; struct sosf_imu_calib; struct sosf_imu_frame;

#endif
