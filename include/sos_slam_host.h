/*
 * sos_slam_host.h -- flat C entry points of the C++ host facade (csrc/host/, libsos_host.so).
 *
 * The facade keeps the reference's FullSystem / FrameHessian / PointHessian / PointFrameResidual /
 * CalibHessian / EnergyFunctional surface in C++ (csrc/host/sos_host.hpp) and drives the HIP backend
 * through the C-ABI of sos_slam.h.  These `sosf_*` functions expose that C++ layer to C / ctypes so the
 * parity tests and bench.py can run the reference's keyframe cycle:
 *
 *   FullSystem::optimize                FS/FullSystemOptimize.cpp:305-489
 *   FullSystem::makeKeyFrame (backend part: insertFrame / insertPoint / insertResidual,
 *   flagPointsForRemoval, marginalizePointsF, marginalizeFrame)   FS/FullSystem.cpp:783-931
 *
 * Status codes are those of sos_slam.h.
 */
#ifndef SOS_SLAM_HOST_H
#define SOS_SLAM_HOST_H

#include "sos_slam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sosf_system sosf_system;

/* FrameHessian initialisation: evalPT pose (camToWorld, R row-major 9 + t 3), state (state_zero :=
 * state with [0..5] = 0, as setEvalPT_scaled / setStateZero leave it), exposure, keyframe id */
typedef struct sosf_frame_init {
  double camToWorld[12];
  double state[10];
  float ab_exposure;
  int32_t frameID;
  float frameEnergyTH;
  int32_t pad;
} sosf_frame_init;

/* new FullSystem (constructor / destructor FS/FullSystem.cpp:58-96, 98-110: EnergyFunctional, CalibHessian, empty window)
 * on HIP device `device`; fails with SOS_ERR_HIP when there is no GPU */
int sosf_create(const sos_params *params, int device, void *hip_stream, sosf_system **out);
int sosf_destroy(sosf_system *sys);
/* CalibHessian(): setValueScaled(fx,fy,cx,cy); value_zero = value (FS/HessianBlocks.h:453-475) */
int sosf_set_calib(sosf_system *sys, const double *value_scaled4);
/* new FrameHessian + makeImages (device) + ef->insertFrame (FS/FullSystem.cpp:650,814) */
int sosf_add_frame(sosf_system *sys, const sosf_frame_init *f, const float *image);
/* the same for a frame whose pyramid is already in image slot `slot` (it was tracked first: sosf_upload_image) */
int sosf_add_frame_from_slot(sosf_system *sys, const sosf_frame_init *f, int slot);

/* ---- keyframe-rate host logic of FullSystem::makeKeyFrame (FS/FullSystem.cpp:783-931), in the order it runs ----------
 * flagFramesForMarginalization (FS/FullSystemMarginalize.cpp:53-133), BEFORE the new keyframe is added: numImmature[i] =
 * immaturePoints.size() of keyframe idx i (the immature points live with the caller); flagged[i] out. */
int sosf_flag_frames_for_marginalization(sosf_system *sys, const int32_t *numImmature, uint8_t *flagged);
/* "add new residuals for old points" (FS/FullSystem.cpp:818-832): one residual towards the newest keyframe for every
 * active point of the older ones, lastResiduals shifted */
int sosf_add_new_frame_residuals(sosf_system *sys, int *count);
/* the points sos_immature_activate returned SOS_ACT_ACTIVATED for (tail of optimizeImmaturePoint,
 * FS/FullSystemOptPoint.cpp:151-185, and activatePointsMT :497-505): idepth / idepth_zero = the activation's idepth,
 * residuals towards the keyframes (idx) whose bit is set in inMask[i], lastResiduals towards the two newest */
int sosf_add_activated_points(sosf_system *sys, int count, const sos_point *pts, const uint32_t *inMask);
/* FullSystem::removeOutliers (FS/FullSystemOptimize.cpp:507-526) */
int sosf_remove_outliers(sosf_system *sys, int *dropped);
/* FullSystem::flagPointsForRemoval (FS/FullSystem.cpp:535-614: isOOB / isInlierNew decisions, re-linearisation and
 * fixLinearizationF of the points that leave as inliers) + ef->dropPointsF + ef->marginalizePointsF (:909, :912) */
int sosf_flag_points_for_removal(sosf_system *sys, int *nMarginalized, int *nDropped);
/* "Marginalize Frames" (FS/FullSystem.cpp:926-931 -> FS/FullSystemMarginalize.cpp:143-236): every flagged keyframe leaves;
 * frameIDs / camToWorld12 (cap entries) receive its id and the pose it leaves with (what the reference hands to the
 * pose graph / viewer as the keyframe's final pose) */
int sosf_marginalize_flagged_frames(sosf_system *sys, int cap, int32_t *frameIDs, double *camToWorld12, int *count);
/* identity of the window's points in allPoints order (host frameID, u, v, host idx) and per keyframe: frameID, flagged,
 * sizes of pointHessians / pointHessiansMarginalized / pointHessiansOut (any pointer may be NULL) */
int sosf_get_point_keys(sosf_system *sys, int32_t *hostFrameID, float *u, float *v, int32_t *hostIdx);
int sosf_get_frame_ids(sosf_system *sys, int32_t *frameID, uint8_t *flagged, int32_t *nPoints, int32_t *nMarg, int32_t *nOut);

/* new PointHessian + ef->insertPoint (FS/FullSystem.cpp:508-510); pts[i].host = frame idx */
int sosf_add_points(sosf_system *sys, int count, const sos_point *pts);
/* new PointFrameResidual + ef->insertResidual (FS/FullSystem.cpp:825-828); res[i].point indexes the
 * points in the order they were added (global running index) */
int sosf_add_residuals(sosf_system *sys, int count, const sos_resid *res);
/* marginalisation prior EnergyFunctional::HM / bM (OB/EnergyFunctional.h:137-138; dim x dim row-major, dim = 4 + 8 * nFrames) */
int sosf_set_prior(sosf_system *sys, const double *HM, const double *bM);
int sosf_get_prior(sosf_system *sys, double *HM, double *bM);

/* FullSystem::optimize(mnumOptIts) (FS/FullSystemOptimize.cpp:305-489): returns the RMSE through *rmse */
int sosf_optimize(sosf_system *sys, int mnumOptIts, float *rmse, int *iterations);
/* setting_forceAceptStep (util/settings.cpp:117, default on).  Off: every Gauss-Newton step is kept only when
 * E_photometric + E_L + E_M decreased (FS/FullSystemOptimize.cpp:387-413); a rejected step is undone with loadSateBackup
 * (:271-287) and the window linearised again at the old state.  The device side then runs the two-step protocol of
 * sos_ba_linearize / sos_ba_apply_res (PointFrameResidual::J and EFResidual::J as separate buffers) instead of the fused
 * calls.  sosf_get_rejected_steps: rejected steps of the last sosf_optimize. */
int sosf_set_force_accept_step(sosf_system *sys, int on);
int sosf_get_rejected_steps(sosf_system *sys, int *count);
/* bench support: pack + resetOOB + linearizeAll(false) + applyRes, then single loop bodies
 * (FS/FullSystemOptimize.cpp:358-413) */
int sosf_prepare(sosf_system *sys);
int sosf_gn_iteration(sosf_system *sys, int iteration, int *canbreak);
/* the caller will keep calling sosf_gn_iteration whatever `canbreak` says (benchmark loops): lets every iteration
 * prefetch the next accumulate (sos_ba_set_prefetch); optimize() decides per iteration by itself */
int sosf_set_pipeline(sosf_system *sys, int on);
/* the next sosf_prepare / sosf_optimize packs and uploads the window again although the graph did not change (measurements of the
 * per-keyframe cost: in a running system every keyframe changes the graph, FS/FullSystem.cpp:814-838) */
int sosf_invalidate_pack(sosf_system *sys);
/* setting_minOptIterations (util/settings.cpp:77, default 1): optimize() does not leave its loop on the convergence test before
 * this many iterations (FS/FullSystemOptimize.cpp:411) */
int sosf_set_min_opt_iterations(sosf_system *sys, int iterations);
/* Host threads of the facade's per-keyframe graph walks -- the record walk of the pack (FS/FullSystemOptimize.cpp:316-329 + the snapshot
 * records) and the consumer of linearizeAll(true) (:125-182, which the reference runs on its IndexThreadReduce workers): contiguous
 * ranges laid out by a prefix sum, so results do not depend on the count.  Process-wide; 1 = serial; default min(4, hardware threads / 2). */
int sosf_set_host_threads(int n);
int sosf_get_host_threads(void);
/* Device-side step of the Gauss-Newton loop (default on; sos_ba_gn_devstep_begin in include/sos_slam.h): the host solves and hands x
 * over, the device forms the frames' new poses, the n^2 precalc records and the deltas inside the launch of the back-substitution.
 * 0: the host computes and stages them as in round 1 (the two differ in the last bits of SE3::exp).  Used when steps are always
 * accepted, up to 17 keyframes; also with the IMU branch of the solve, exchange hooks or a communicator (x and the frame states are
 * replicated over the ranks). */
int sosf_set_device_step(sosf_system *sys, int on);
/* Device-resident Gauss-Newton loop (sos_ba_gn_resident_*): solveSystemF, the frame half of doStepFromBackup and
 * setPrecalcValues run on the device, the host only decides whether to continue.  OFF by default: the (4 + 8 n)-dimensional
 * LDL^T is a chain of ~100 dependent pivots, which one compute unit walks in ~45 us where a host core needs 16 (DESIGN.md,
 * "host out of the loop: measured"); with it off the host solves (blocked LDL^T).  Never taken while IMU factors, callback
 * hooks or an RCCL communicator are attached, with setting_forceAceptStep off, or for windows of more than 16 keyframes. */
int sosf_set_resident(sosf_system *sys, int on);
/* How the last Gauss-Newton iteration actually ran (the two switches above are requests; the conditions listed with them decide):
 * 0 = step on the host, 1 = step on the device, 2 = device-resident loop, 3 = energy-checked step (setting_forceAceptStep off), -1 = none yet. */
int sosf_get_loop_mode(sosf_system *sys, int *mode);
int sosf_counts(sosf_system *sys, int *nFrames, int *nPoints, int *nResiduals);

int sosf_get_frame(sosf_system *sys, int idx, double *camToWorld12, double *state10, double *state_zero10,
                   float *frameEnergyTH);
int sosf_get_calib(sosf_system *sys, double *value_scaled4);
/* per point, in allPoints order: idepth, idepth_hessian, maxRelBaseline, numGoodResiduals (any may be NULL) */
int sosf_get_points(sosf_system *sys, float *idepth, float *idepth_hessian, float *maxRelBaseline,
                    int32_t *numGoodResiduals);
/* per point still in the window, same order: the running index under which it was added (sosf_add_points) */
int sosf_get_point_ids(sosf_system *sys, int32_t *addIdx);
/* per residual, in packing order (points -> residualsAll): state_state, isActive; removed = dropped by
 * the final linearizeAll(true) */
int sosf_get_residuals(sosf_system *sys, int32_t *state_state, int32_t *isActive, int32_t *removed);
/* identity of every residual of the current graph, same order as sosf_get_residuals: the running index under which
 * its point was added and the frameID of its target keyframe (the "active index set" of the parity tests) */
int sosf_get_residual_ids(sosf_system *sys, int32_t *pointAddIdx, int32_t *targetFrameID);
int sosf_get_lastX(sosf_system *sys, double *x);
int sosf_get_stats(sosf_system *sys, int *resInA, int *resInL, int *resInM);
/* inspection: with keep on, every solveSystemF keeps what it assembled from the device's accumulation before the solve --
 * H_top = HA + HL with the calibration / frame priors of the L stitch (OB/AccumulatedTopHessian.cpp:292-300), b_top, and
 * the Schur side H_sc, b_sc, all (4 + 8 n) full symmetric row-major -- i.e. the inputs of OB/EnergyFunctional.cpp:1046-1171 */
int sosf_keep_last_system(sosf_system *sys, int on);
int sosf_get_last_system(sosf_system *sys, double *H_top, double *b_top, double *H_sc, double *b_sc);

/* FullSystem::flagPointsForRemoval restricted to an explicit list + ef->marginalizePointsF
 * (FS/FullSystem.cpp:535-614, 909-912; OB/EnergyFunctional.cpp:891-936): the listed points (by the running
 * index under which they were added) are re-linearized, fixed (fixLinearizationF) and marginalised into
 * HM / bM, or dropped when their idepth_hessian is below setting_minIdepthH_marg. */
int sosf_marginalize_points(sosf_system *sys, const int32_t *pointIdx, int count);
/* ef->dropPointsF (OB/EnergyFunctional.cpp:938-952) for an explicit list (PS_DROP) */
int sosf_drop_points(sosf_system *sys, const int32_t *pointIdx, int count);
/* FullSystem::marginalizeFrame -> ef->marginalizeFrame (FS/FullSystemMarginalize.cpp:143-236,
 * OB/EnergyFunctional.cpp:730-889, IMU off): frame idx must have no points left */
int sosf_marginalize_frame(sosf_system *sys, int frameIdx);

/* Multi-GPU hooks: when set, solveSystemF accumulates the local shard, hands the packed
 * fp32 accumulator buffer (device pointer) to `allreduce` (RCCL sum over ranks, must be complete on return)
 * and stitches from the reduced buffer; setNewFrameEnergyTH asks `nth` for the global order statistic:
 * given the local energies of residuals targeting the newest frame, return the element at index
 * (int)(frac * N_global) of the globally sorted list (or a negative value if the global list is empty). */
typedef void (*sosf_allreduce_fn)(void *user, float *dev_ptr, size_t nfloats);
typedef float (*sosf_nth_fn)(void *user, const float *energies, int count, float frac);
int sosf_set_hooks(sosf_system *sys, sosf_allreduce_fn allreduce, sosf_nth_fn nth, void *user);
/* Companion of the callback exchange for the keyframe-rate fp64 sums: the shard-local M - Msc (and resInM) of
 * marginalizePointsF (OB/EnergyFunctional.cpp:891-936) and the mean |idepth| behind the termination test of
 * doStepFromBackup (FS/FullSystemOptimize.cpp:240-257) must be identical on every rank.  `buf` is a HOST buffer,
 * summed over ranks in place.  With sosf_set_hooks but without this hook sosf_marginalize_points returns SOS_ERR_STATE
 * instead of letting the priors diverge.  Shard-local by design (not exchanged): EnergyFunctional::connectivityMap. */
typedef void (*sosf_allreduce_f64_fn)(void *user, double *host_buf, size_t count);
int sosf_set_allreduce_f64_hook(sosf_system *sys, sosf_allreduce_f64_fn allreduce_f64);
/* Native exchange: attach an RCCL communicator (sos_comm_create) to the system's backend; the all-reduce / all-gather
 * then run on the library's stream inside the fused calls (no callbacks, pipelining stays on).  NULL detaches. */
int sosf_set_comm(sosf_system *sys, sos_comm *comm);

/* ---- CoarseTracker / ScaleOptimizer (FS/CoarseTracker.h:27-48, FS/ScaleOptimizer.h:43-104) -------------
 * The pose LM loop runs on the device (one launch per call); the scale loop on the host around device residual passes. */
typedef struct sosf_tracker sosf_tracker;
/* makeImages of a frame that is not (yet) a keyframe (FS/FullSystem.cpp:650, 1114): returns its image slot */
int sosf_upload_image(sosf_system *sys, const float *image, int *slot_out);
int sosf_release_image(sosf_system *sys, int slot);
/* reserves a free image slot of the system's context without filling it: the caller builds the pyramid there itself
 * (sos_undistort_frame on sosf_ctx(sys): raw camera frame -> undistortion -> pyramid, U/Undistort.cpp:361-458) */
int sosf_alloc_slot(sosf_system *sys, int *slot_out);
int sosf_tracker_create(sosf_system *sys, sosf_tracker **out);
int sosf_tracker_destroy(sosf_tracker *trk);
/* makeK + setCoarseTrackingRef(frameHessians) (FS/FullSystem.cpp:889-890): reference = newest keyframe, points =
 * those whose lastResiduals[0] is IN (centerProjectedTo, HdiF of the last optimize()) */
int sosf_tracker_set_ref(sosf_tracker *trk, int32_t *pc_n_out);
int sosf_tracker_set_ref_raw(sosf_tracker *trk, int npts, const float *u, const float *v, const float *idepth,
                             const float *hdi, int32_t *pc_n_out);
sos_tracker *sosf_tracker_handle(sosf_tracker *trk);
/* trackNewestCoarse (FS/CoarseTracker.cpp:366-552); lastToNew12 / aff2 are in/out */
int sosf_tracker_track(sosf_tracker *trk, int newSlot, float new_ab_exposure, double *lastToNew12, double *aff2,
                       int coarsestLvl, const double *minResForAbort5, double *lastResiduals5, double *flow3, int *ok);
/* FullSystem::optimizeScale(scale_optimizer) (FS/FullSystem.cpp:1117-1177, called from makeKeyFrame :897-903): while the scale is
 * not trapped the seven guesses {0.1, 0.2, 0.5, 1, 2, 5, 10} are optimised -- side by side in one launch -- and the smallest positive
 * error wins; once trapped, one run from trackingRefScale (frameHessians.back()->shell->trackingRef->scale).  state2 = {scaleTrapped,
 * scale_opt_fails} of the reference, kept by the caller; thres = setting_scale_opt_thres.  *new_scale = the accepted scale or -1
 * (rejected: error not in (0, thres)); the caller then does HCalib.setScaleScaledZero(new_scale) if it is positive. */
int sosf_tracker_optimize_scale_kf(sosf_tracker *trk, int stereoSlot, const double *tfmF0ToF1_12, const float *K1_level0,
                                   float trackingRefScale, int coarsestLvl, float thres, int32_t *state2, float *new_scale,
                                   float *scale_error);
/* The LM loop of trackNewestCoarse / pose_estimate runs on the device as one launch (sos_tracker_track, include/sos_slam.h)
 * by default; 0 selects the loop on the host with one device round trip per residual evaluation (same decisions, same
 * arithmetic per pixel; the two differ by the rounding of the 8x8 solve).  last_evals: residual evaluations of the last call. */
int sosf_tracker_set_device_lm(sosf_tracker *trk, int on);
int sosf_tracker_last_evals(sosf_tracker *trk, int *evals);
/* where the last one-launch loop spent its time (sos_tracker_lm_profile, include/sos_slam.h): microseconds [kernel, residual passes,
 * all-gather of the chunk sums, bookkeeping between evaluations, of which 8 x 8 solve, SE3::exp + request, number of evaluations] */
int sosf_tracker_lm_profile(sosf_tracker *trk, int hyp, double *us7);
/* The one-launch loops need all their workgroups resident; their waits are bounded (sos_tracker_set_lm_spin_limit, include/sos_slam.h).
 * When a launch comes back SOS_ERR_TIMEOUT -- other streams kept the device busy -- the facade redoes that call with the host loop
 * around the device passes (same decisions; hypotheses of the batch one by one) and counts it here.  set_lm_spin_limit forwards the
 * bound (rounds of about a microsecond, default 2^18; 0 = give up at the first unmet wait). */
int sosf_tracker_set_lm_spin_limit(sosf_tracker *trk, unsigned rounds);
int sosf_tracker_lm_fallbacks(sosf_tracker *trk, int *count);
/* FullSystem::trackNewCoarse, the hypothesis list (FS/FullSystem.cpp:150-213): slast_2_sprelast and lastF_2_slast as the
 * reference forms them from the frame history, lastF_2_fh_imu12 = the IMU-predicted motion or NULL; posesValid = all
 * three shells have poseValid.  Writes n tries (12 doubles each, row-major R | t); 84 without / 85 with the IMU try. */
int sosf_tracker_make_tries(const double *slast_2_sprelast12, const double *lastF_2_slast12, const double *lastF_2_fh_imu12,
                            int posesValid, int cap, double *tries12, int *n_out);
/* FullSystem::trackNewCoarse, the loop over the tries (FS/FullSystem.cpp:219-283) with the tries evaluated `batch` at a time
 * in one launch (the first try alone); the sequential take-over / early-exit decisions of the reference are replayed in
 * order on the batch results, so the outcome equals the one-by-one loop.  Out: lastF_2_fh, aff_g2l, achievedRes (the new
 * lastCoarseRMSE), flowVecs; info4 = {tryIterations, index of the winning try or -1, tries evaluated, haveOneGood}. */
int sosf_tracker_track_hypotheses(sosf_tracker *trk, int newSlot, float new_ab_exposure, int nTries, const double *tries12,
                                  const double *aff_last2, int coarsestLvl, const double *lastCoarseRMSE5, double reTrackThreshold,
                                  int batch, double *lastF_2_fh12, double *aff2, double *achievedRes5, double *flow3, int *info4);
/* Loop-closure aligner (N4), PoseEstimator::estimate (src/LoopClosure/PoseEstimator.cpp:288-495): set_points3d takes
 * `matched_frame->pts_dso` (xyz AoS; colors[l * n + i]), the camera of the current frame and the matched frame's
 * exposure; pose_estimate runs the LM loop from refToNew12 (in / out) on the pyramid in newSlot and applies the three
 * acceptance tests (affine parameters sane, pose_error < setting_loop_direct_thres, inlier percentage > INNER_PERCENT). */
int sosf_tracker_set_points3d(sosf_tracker *trk, const sos_calib *cam, float matched_ab_exposure, int n, const float *xyz,
                              const float *colors);
int sosf_tracker_pose_estimate(sosf_tracker *trk, int newSlot, float new_ab_exposure, double *refToNew12, int coarsestLvl,
                               float loopDirectThres, int innerPercent, float *poseError, int *inlierPercent, int *ok);
/* LoopHandler::savePose (src/LoopClosure/LoopHandler.cpp:62-76): one line per keyframe, "incoming_id tx ty tz" with
 * std::setprecision(6) default-float formatting (= "%.6g"); t_wc holds n translations */
int sosf_write_poses(const char *path, int n, const int32_t *incoming_id, const double *t_wc);
/* optimizeScale (FS/ScaleOptimizer.cpp:120-230) */
int sosf_tracker_optimize_scale(sosf_tracker *trk, int stereoSlot, const double *tfmF0ToF1_12, const float *K1_level0,
                                float *scale_inout, int coarsestLvl, float *rmse);

/* accumulated wall-clock seconds per phase of sosf_gn_iteration: 0 sos_ba_gn_accumulate (wait + copy), 1 assemble
 * H/b, 2 LDLT, 3 doStepFromBackup (includes 4), 4 setPrecalcValues, 5 sos_ba_gn_step + post-processing (includes 6),
 * 6 host mirrors + setNewFrameEnergyTH, 7 backupState */
int sosf_get_timing(double *phases8, int reset);

/* the facade's dense symmetric solve (pivoted LDL^T standing in for Eigen's ldlt().solve, OB/EnergyFunctional.cpp:1148);
 * which = 0: blocked production variant, 1: unblocked reference variant.  Exposed for the CPU test-suite. */
int sosf_ldlt_solve(const double *A, const double *b, double *x, int n, int which);
/* the same system solved as the cached visual-inertial solve does (sos_imu.cpp): the leading m unknowns eliminated by a partial
 * factorisation with pivots from the leading block only, forward pass, the trailing (n - m) block -- its Schur complement -- solved
 * on its own, backward pass.  Exposed for the CPU test-suite. */
int sosf_ldlt_partial_solve(const double *A, const double *b, double *x, int n, int m);
/* the prior algebra of EnergyFunctional::marginalizeFrame in its visual form (OB/EnergyFunctional.cpp:788-858) on its own: (HM, bM) of
 * dimension 4 + 8 n, the keyframe idx with its pose prior -> the prior of dimension 4 + 8 (n - 1).  What sosf_marginalize_frame runs on
 * the system's prior; exposed for the CPU test-suite. */
/* the per-keyframe host math that feeds the device, on frames built for the occasion (no system, no device): setEvalPT(evalPT, state_zero) +
 * setState(state) -> PRE_camToWorld (n x 12), FrameFramePrecalc::set of every pair (n x n records, index host + n target), setAdjointsF
 * (adHost / adTarget: n x n row-major 8 x 8) and setDeltaF's adHTdeltaF (n x n x 8).  Any output may be NULL.  Exposed for the CPU
 * test-suite. */
int sosf_host_frame_math(int n, const double *evalPT12, const double *state_zero10, const double *state10, const float *ab_exposure,
                         const double *calib_value4, const double *calib_value_zero4, double *camToWorld12, sos_precalc *precalc, double *adHost,
                         double *adTarget, float *adHTdeltaF);
/* the decision of FullSystem::flagFramesForMarginalization (FS/FullSystemMarginalize.cpp:54-141) on plain arrays, window order: frameID, the
 * points a keyframe still hosts (active + immature) / has lost (marginalised + dropped), refToFh[0] of fromToVecExposure(newest -> keyframe),
 * distanceLL[h * n + t] of the pair (h, t); flagged[h] is set to 1 where the reference flags.  What sosf_flag_frames_for_marginalization
 * runs on the system's own numbers; exposed for the CPU test-suite. */
int sosf_flag_frames(int n, const int32_t *frameID, const int32_t *pointsIn, const int32_t *pointsOut, const double *refToFh0, const float *distanceLL,
                     uint8_t *flagged);
/* FullSystem::setNewFrameEnergyTH (FS/FullSystemOptimize.cpp:84-124) on a list of energies (state_NewEnergyWithOutlier >= 0 of the residuals
 * towards the newest keyframe): the element at index (int)(frameEnergyTHN * count) -- a float product -- through the threshold formula;
 * 12 * 12 * patternNum for an empty list.  Exposed for the CPU test-suite. */
int sosf_new_frame_energy_th(const float *energies, int count, float frameEnergyTHN, float facMedian, float constWeight, float overall, float *th);
/* the visual solve of solveSystemF from its pieces (OB/EnergyFunctional.cpp:1069-1148): H_top with the priors of the L stitch in, b_top, H_sc,
 * b_sc, the prior (HM, bM), delta = getStitchedDeltaF(), lambda -> x (dimension 4 + 8 n).  Reads the UPPER triangles of H_top / H_sc / HM
 * for the matrix (what the device delivers), all of HM for bM + HM delta.  What every Gauss-Newton iteration runs between
 * sos_ba_gn_accumulate and the step; exposed for the CPU test-suite. */
int sosf_solve_system(int n, const double *H_top, const double *b_top, const double *H_sc, const double *b_sc, const double *HM, const double *bM,
                      const double *delta, double lambda, double *x);
int sosf_marginalize_frame_prior(int n, int idx, const double *HM, const double *bM, const double *prior8, const double *delta_prior8, double *HM_out,
                                 double *bM_out);

/* ---- the frame-rate loop: FullSystem::addActiveFrame -> trackNewestCoarse -> traceNewCoarse -> keyframe decision -> makeKeyFrame
 * (FS/FullSystem.cpp:616-766, 311-361, 783-931, 375-531, 1071-1097), visual part, in C++ (csrc/host/sos_sequence.cpp).  The object
 * owns a CoarseTracker, a PixelSelector and the immature points of every keyframe (FrameHessian::immaturePoints).  The IMU / stereo
 * branches of makeKeyFrame: sosf_sequence_enable_imu / _enable_stereo + sosf_add_active_frame_ex, declared behind the IMU records below. */
typedef struct sosf_sequence sosf_sequence;
typedef struct sosf_sequence_params {
  sos_trace_params trace;
  sos_activate_params activate;
  sos_pixsel_params pixsel;
  float desiredPointDensity;   /* setting_desiredPointDensity (util/settings.cpp:66) */
  float immatureDensity;       /* setting_desiredImmatureDensity (:65) */
  float minTraceQuality;       /* setting_minTraceQuality = 3 */
  int32_t kfEvery;             /* > 0: every kfEvery-th frame is a keyframe (test sequences); 0: the decision of FS/FullSystem.cpp:709-732 */
  int32_t maxOptIterations;    /* setting_maxOptIterations = 6 */
  int32_t patternPadding;      /* 2 */
  float kfGlobalWeight, maxShiftWeightT, maxShiftWeightR, maxShiftWeightRT, maxAffineWeight; /* util/settings.cpp:36-42 */
} sosf_sequence_params;
typedef struct sosf_frame_result {
  int32_t trackingOk, isKeyframe;
  double refToNew[12];         /* lastF_2_fh of trackNewCoarse */
  double camToWorld[12];       /* shell->camToWorld = trackingRef->camToWorld * camToTrackingRef */
  double aff[2];               /* aff_g2l */
  double trackResiduals[5];    /* achievedRes */
  double flow[3];              /* lastFlowIndicators */
  /* keyframes only */
  int32_t nActivated, nDeletedImmature, nPointsBeforeOpt, iterations;
  float rmse;
  int32_t nOutliersRemoved, nMargPoints, nDroppedPoints, nNewImmature, nMargFrames;
  int32_t margFrameIDs[8];
  double margCamToWorld[8 * 12];
  float newScale, scaleError;  /* stereo branch: FullSystem::optimizeScale of this keyframe (newScale -1: rejected / not run) */
} sosf_frame_result;
int sosf_sequence_create(sosf_system *sys, const sosf_sequence_params *prm, const uint8_t *randomPattern /* w*h, as sos_pixsel_create */,
                         sosf_sequence **out);
int sosf_sequence_destroy(sosf_sequence *seq);
/* the window the initialiser hands over is in the system (sosf_add_frame* / sosf_add_points / sosf_add_residuals): optimize(),
 * removeOutliers, setCoarseTrackingRef, makeNewTraces on every keyframe (FS/FullSystem.cpp:933-1069 tail) */
int sosf_sequence_bootstrap(sosf_sequence *seq, float *rmse, int *iterations);
/* FullSystem::addActiveFrame for the frame whose pyramid is in image slot `slot` (sos_undistort_frame / sosf_upload_image).
 * T_init12: initial refToNew (NULL: the motion model).  A frame that does not become a keyframe leaves its slot to the caller
 * (sosf_release_image); a keyframe's slot belongs to the window until the keyframe is marginalised. */
int sosf_add_active_frame(sosf_sequence *seq, int slot, int frameID, float ab_exposure, const double *T_init12, sosf_frame_result *out);
int sosf_sequence_immature_count(sosf_sequence *seq, int frameID, int *count);
int sosf_sequence_get_immature(sosf_sequence *seq, int frameID, int capacity, sos_immature *out, float *type);

/* ---- candidate selection of FullSystem::activatePointsMT (FS/FullSystem.cpp:375-470) with CoarseDistanceMap
 * (FS/CoarseTracker.cpp:766-954): host logic around sos_immature_activate.  The loop is inherently ordered (every
 * accepted candidate is inserted into the distance map before the next one is tested), so it stays on the host.
 *   currentMinActDist update of :377-399 */
float sosf_next_min_act_dist(float currentMinActDist, int nPoints, float desiredPointDensity);
/* makeDistanceMap + the candidate loop.  w1 x h1 = level-1 image size; per keyframe f (idx order, `newest` skipped):
 * KRKi[f] (3x3 row-major) = K[1] * R(newest <- f) * Ki[0], Kt[f] = K[1] * t(newest <- f), computed by the caller
 * exactly as FS/FullSystem.cpp:410-415 / FS/CoarseTracker.cpp:806-809.  Active points (u, v, idepth_scaled, host
 * idx) in frames -> points order; candidates in the order the reference visits them (frames, then
 * host->immaturePoints), with cand_type = ImmaturePoint::my_type and hostFlagged[f] = flaggedForMarginalization.
 * decision[i]: SOSF_SEL_KEEP (stays immature), SOSF_SEL_DELETE, SOSF_SEL_OPTIMIZE (goes to sos_immature_activate,
 * in this order).  distFinal (optional, w1*h1): fwdWarpedIDDistFinal after the loop. */
#define SOSF_SEL_KEEP 0
#define SOSF_SEL_OPTIMIZE 1
#define SOSF_SEL_DELETE (-1)
int sosf_activate_select(int w1, int h1, int nFrames, int newest, const float *KRKi, const float *Kt, int nActive,
                         const float *act_u, const float *act_v, const float *act_idepth_scaled, const int32_t *act_host,
                         float currentMinActDist, float minTraceQuality, int nCand, const sos_immature *cand,
                         const int32_t *cand_host, const float *cand_type, const uint8_t *hostFlagged, int8_t *decision,
                         float *distFinal);

/* ---- IMU / spline factor assembly on the backend boundary ---------------------------------------
 * The host-side fp64 block that sits inside EnergyFunctional::solveSystemF between accumulate and solve:
 * getImuHessian / getImuHessianCurrentFrame (OB/EnergyFunctional.cpp:288-494), FrameHessian::getImuHi
 * (FS/HessianBlocks.cpp:178-225), expandHbtoFitImu (:256-286) and the IMU branch of solveSystemF (:1053-1171: expanded
 * system, marginalisation prior, Schur complement, spline constraint rows, removal of unconstrained states, Jacobi-scaled
 * LDLT, split of x into pose / scale / IMU steps).  State layout per keyframe, as in the reference: 8 (pose, a, b) + 6
 * (accelerometer, gyroscope bias) + 15 (spline: linear rot 3, quadratic trans 3 rot 3, cubic trans 3 rot 3) = 29; full
 * dimension CPARS + 1 (scale) + 29 n. */
#define SOSF_IMU_DIM(n) (4 + 1 + 29 * (n))
typedef struct sosf_imu_settings {
  double weight_imu[36];       /* setting_weight_imu, 6x6 row-major */
  double weight_imu_bias[36];  /* setting_weight_imu_bias */
  double gravity[3];           /* setting_gravity */
  double rot_imu_cam[9];       /* setting_rot_imu_cam, row-major */
  double maxImuInterval;       /* setting_maxImuInterval */
  int32_t enable_scale_opt;    /* setting_enable_scale_opt */
  int32_t pad;
} sosf_imu_settings;
typedef struct sosf_imu_calib {   /* the IMU part of CalibHessian (FS/HessianBlocks.h:439-445) */
  double scale, scale_zero;
  int32_t scale_trapped, imu_initialized;
} sosf_imu_calib;
typedef struct sosf_imu_frame {
  double timestamp;            /* shell->timestamp */
  double camToWorld[12];       /* PRE_camToWorld: R row-major | t */
  double evalPT_R[9];          /* get_camToWorld_evalPT().rotationMatrix() */
  double state_imu[21];        /* FrameHessian::state_imu (unscaled) */
  double state_imu_zero[21];
  int32_t trackingRefIsPrev;   /* shell->trackingRef == the previous keyframe's shell */
  int32_t n_imu;
  const double *imu;           /* n_imu x 7: timestamp, acc xyz, gyro xyz (FrameHessian::imu_data) */
} sosf_imu_frame;
/* getImuHi for one IMU sample at relative time tt <= 0: JsTW (6), JfTW (29 x 6 row-major), Hss, Hff (29 x 29), Hfs (29) */
int sosf_imu_get_Hi(const sosf_imu_settings *S, const sosf_imu_calib *C, const sosf_imu_frame *f, double tt, double *JsTW,
                    double *JfTW, double *Hss, double *Hff, double *Hfs);
/* getImuHessian: H (dim x dim), b (dim), the stacked spline constraints J_cst (n_cst x dim, caller provides 6 n rows),
 * r_cst, and spline_valid per frame */
int sosf_imu_hessian(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *frames, double *H, double *b,
                     double *J_cst, double *r_cst, int32_t *n_cst, int32_t *spline_valid);
/* expandHbtoFitImu: (4 + 8 n) -> SOSF_IMU_DIM(n) */
int sosf_imu_expand(int n, const double *H, const double *b, double *He, double *be);
/* the IMU branch of solveSystemF from the accumulated H_top = HA + HL, b_top, H_sc, b_sc (dimension 4 + 8 n), the
 * marginalisation prior HM / bM (dimension SOSF_IMU_DIM(n)) and delta = getStitchedDeltaF(): x (4 + 8 n), scale_step,
 * step_imu (n x 21) */
int sosf_imu_solve(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *frames, const double *H_top,
                   const double *b_top, const double *H_sc, const double *b_sc, const double *HM, const double *bM,
                   const double *delta, double lambda, double *x, double *scale_step, double *step_imu);
/* The same solve in two calls, so that what does not need the device's H / b runs while the accumulation is in flight: _prepare takes
 * the records, the prior and delta (kept by pointer until _finish, which must follow on the same thread with nothing changed in
 * between), _finish the stitched system.  With the scale trapped (first-estimate Jacobians: getImuHi at state_imu_zero / scale_zero /
 * evalPT, FS/HessianBlocks.cpp:178-225) everything of the KKT matrix except the visual block is constant over the iterations of one
 * optimize(): the IMU states and constraint multipliers are eliminated once and the factor is kept (thread-local) for as long as the
 * inputs it rests on -- settings, lambda, timestamps, linearisation points, the whole of HM -- compare equal value by value
 * (prior_id != 0 is the caller's name for the VALUES of HM: a call with the same name, the same pointer and the same diagonal skips
 * the comparison of the other dim^2 values -- the facade names its own prior by a counter of its writes; 0 = always compared); an
 * iteration then costs the right-hand sides, the (4 + 1 + 8 n)-dimensional border solve and two substitutions.  Results agree with
 * the literal form to rounding.  sosf_imu_solve_mode(0) selects the literal form for every call, (1) the cached one (default;
 * SOS_IMU_CACHE=0 in the environment = 0), any other value only reads; returns the previous mode.  sosf_imu_solve_stats: solves on
 * a kept factor / factor rebuilds / literal-form solves of the calling thread. */
int sosf_imu_solve_prepare(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *frames, const double *HM,
                           const double *bM, const double *delta, double lambda, uint64_t prior_id);
int sosf_imu_solve_finish(const double *H_top, const double *b_top, const double *H_sc, const double *b_sc, double *x, double *scale_step,
                          double *step_imu);
int sosf_imu_solve_mode(int mode);
/* between _prepare and _finish: 1 = the kept factor will serve (it reads only the upper triangles of H_top / H_sc), 0 = the literal form
 * (reads them whole), -1 = nothing prepared */
int sosf_imu_solve_prepared_form(void);
int sosf_imu_solve_stats(int32_t *kept, int32_t *rebuilt, int32_t *literal, int reset);

/* EnergyFunctional::marginalizeFrame with IMU enabled (OB/EnergyFunctional.cpp:730-889): the IMU factors linking keyframe
 * idx to its neighbours are folded into the prior (getImuHessianCurrentFrame of idx + 1 and, if idx > 0, of idx, linearised
 * at delta: HM += margWeightFac * HM_change, bM += margWeightFac * (bM_change - HM_change * delta2)), the keyframe's 29
 * states are moved to the end, its pose prior added, the 15 spline states dropped when no valid spline constrains them,
 * and the block is eliminated by a Jacobi-scaled Schur complement.  HM / bM: SOSF_IMU_DIM(n) in, SOSF_IMU_DIM(n - 1) out
 * (written to HM_out / bM_out); delta = getStitchedDeltaF() (4 + 8 n); prior8 / delta_prior8 = fh->prior, fh->delta_prior. */
int sosf_imu_marginalize_frame(const sosf_imu_settings *S, const sosf_imu_calib *C, int n, const sosf_imu_frame *frames, int idx,
                               const double *delta, const double *prior8, const double *delta_prior8, double margWeightFac,
                               const double *HM, const double *bM, double *HM_out, double *bM_out);
/* ---- VIO front-end around the assembly (FS/HessianBlocks.cpp:225-429; called from FullSystem::addActiveFrame :693-706,
 * makeKeyFrame :806-882 and optimize, FS/FullSystemOptimize.cpp:462-478).  Host fp64.  sosf_imu_shell = the fields of
 * FrameShell they touch; biases and spline coefficients live in sosf_imu_frame::state_imu (unscaled, 21). */
typedef struct sosf_imu_shell {
  double timestamp;
  double camToWorld[12];   /* shell->camToWorld, R row-major | t */
  double velInWorld[3];
} sosf_imu_shell;
/* FrameHessian::propagateImuState(last_shell, last_imu_bias, HCalib) (:357-404): gyroscope integration from the last
 * shell's rotation, least-squares fit of the quadratic / cubic spline coefficients to the frame's IMU samples, then
 * setImuStateScaled + setImuStateZero and the velocity.  last_imu_bias6 = the previous frame's imu_bias (scaled: ba, bg).
 * The accelerometer fit has a zero design column (the linear term is carried by velInWorld); the reference relies on how
 * Eigen's dynamic-size inverse treats the singular normal matrix, which is reproduced (see lu_inverse in sos_imu.cpp). */
int sosf_imu_propagate_state(const sosf_imu_settings *S, const sosf_imu_calib *C, sosf_imu_frame *f, sosf_imu_shell *shell,
                             const sosf_imu_shell *last_shell, const double *last_imu_bias6);
/* FrameHessian::updateVel(last_shell) (:406-412) */
int sosf_imu_update_vel(const sosf_imu_frame *f, sosf_imu_shell *shell, const sosf_imu_shell *last_shell);
/* FrameHessian::initializeImu(frame_hessians, HCalib) (:253-355) on exactly five keyframes (frames[4] = the newest = the base
 * frame; frames[i].camToWorld = PRE_camToWorld, shells[i] = its shell): cubic spline through the poses of frames 1..3,
 * velocities and spline states of all five, gyroscope bias as the mean prediction error over the IMU samples of frames
 * 2..4, and (unless setting_enable_scale_opt) the metric scale as a one-parameter least squares; *ok = 0 when that scale
 * is negative ("IMU initialization failed").  Writes state_imu / state_imu_zero, velInWorld, calib->scale / scale_zero /
 * imu_initialized. */
int sosf_imu_initialize(const sosf_imu_settings *S, sosf_imu_calib *calib, sosf_imu_frame *frames5, sosf_imu_shell *shells5, int *ok);
/* CalibHessian::tryTrapScale() (:414-429): scale_queue10 / scale_queue_i are the caller's copy of CalibHessian's ring
 * (initially LinSpaced(10, -10, -100), index 0); thres = setting_scale_trap_thres */
int sosf_imu_try_trap_scale(sosf_imu_calib *calib, double *scale_queue10, int32_t *scale_queue_i, double thres);

/* Switches the facade's solveSystemF to the IMU branch (S != NULL) or back (S == NULL).  The records are caller-owned and
 * must outlive the system's iterations: frames[i] belongs to keyframe idx i (its camToWorld / evalPT_R are refreshed by
 * the facade before every solve; state_imu and calib->scale are stepped after it, as doStepFromBackup does with unit
 * step factors); HM / bM: the marginalisation prior in the expanded dimension SOSF_IMU_DIM(n), read at every solve (with
 * first-estimate Jacobians the solve keeps a factor of the constant part of its system, sosf_imu_solve_prepare; a caller-owned HM
 * is compared WHOLE against the copy the factor was built from, so values rewritten in place are noticed without another call).
 * HM = bM = NULL: the facade keeps the expanded prior itself, as EnergyFunctional does with setting_enable_imu: it starts from
 * expandHbtoFitImu of the current (visual) prior, insertFrame grows it by 29 states (OB/EnergyFunctional.cpp:666-677),
 * marginalizePointsF adds the expanded M - Msc (:928-932), marginalizeFrame runs the IMU form (:733-889) before it drops the
 * keyframe.  frames[i] must describe keyframe idx i whenever a solve or a frame marginalisation runs: the caller appends a
 * record when it adds a keyframe (calling sosf_set_imu again with NULL priors only renews the pointers); a frame
 * marginalisation erases record idx from the caller's array in place (the later records move down by one), so the array
 * stays aligned when several keyframes leave in one sosf_marginalize_flagged_frames.  The IMU samples of the leaving
 * keyframe go in front of those of its successor (FS/FullSystemMarginalize.cpp:226-228: the successor's factor then spans the
 * whole interval): the facade rewrites n_imu / imu of record idx + 1 to a merged list it owns (valid while that record refers to
 * it; released when no record does, or with sosf_set_imu(NULL)).  A caller that rebuilds its records from its own sample
 * storage afterwards has to carry the same hand-over there.  sosf_get_imu_prior copies the expanded prior out. */
int sosf_set_imu(sosf_system *sys, const sosf_imu_settings *S, sosf_imu_calib *calib, sosf_imu_frame *frames, const double *HM,
                 const double *bM);
int sosf_get_imu_prior(sosf_system *sys, double *HM, double *bM, int *dim);
/* scale_step and step_imu (n x 21) of the last solve */
int sosf_get_imu_step(sosf_system *sys, double *scale_step, double *step_imu);

/* ---- the IMU / stereo branches of the frame-rate loop (FS/FullSystem.cpp:800-807 setImuData + propagateImuState, :841-849 initializeImu
 * at the fifth keyframe, FS/FullSystemOptimize.cpp:437-479 shell poses / updateVel / setImuStateZero / tryTrapScale, :878-886, :897-903
 * optimizeScale, FS/FullSystemMarginalize.cpp:226-228 sample hand-over).  The sequence keeps per keyframe the FrameShell fields, the 21
 * IMU states with their linearisation point and the samples, and the IMU part of CalibHessian; the facade keeps the expanded prior. */
typedef struct sosf_frame_extra {
  double timestamp;     /* shell->timestamp */
  int32_t n_imu;        /* samples since the previous frame (FullSystem::addActiveFrame's new_imu_data) */
  int32_t stereoSlot;   /* image slot of the stereo partner, -1: none (released by the caller after the call) */
  const double *imu;    /* n_imu x 7: timestamp, acc xyz, gyro xyz */
} sosf_frame_extra;
/* after the bootstrap window is in the system, before sosf_sequence_bootstrap: timestamps / samples of its keyframes in window order */
int sosf_sequence_enable_imu(sosf_sequence *seq, const sosf_imu_settings *S, int nBoot, const double *timestamps, const int32_t *n_imu,
                             const double *const *imu);
int sosf_sequence_enable_stereo(sosf_sequence *seq, const double *tfmF0ToF1_12, float scaleOptThres /* setting_scale_opt_thres */);
int sosf_sequence_bootstrap_ex(sosf_sequence *seq, int stereoSlot /* of the newest bootstrap keyframe, -1: none */, float *rmse, int *iterations);
int sosf_add_active_frame_ex(sosf_sequence *seq, int slot, int frameID, float ab_exposure, const double *T_init12, const sosf_frame_extra *extra,
                             sosf_frame_result *out);
int sosf_sequence_get_imu(sosf_sequence *seq, int frameID, double *state21, double *zero21, double *vel3);
int sosf_sequence_get_imu_calib(sosf_sequence *seq, sosf_imu_calib *out);
int sosf_sequence_get_scale_state(sosf_sequence *seq, int32_t *state2 /* FullSystem::scaleTrapped, scale_opt_fails */);
/* for the parity tests: what the graph looks like BETWEEN the stages of the last makeKeyFrame.  which = 0 keyframes flagged for
 * marginalisation (frameID), 1 activated points (host frameID, u, v; insertion order), 2 residuals after optimize() (host frameID, u, v,
 * target frameID), 3 points after flagPointsForRemoval (host frameID, u, v); *count = doubles available */
int sosf_sequence_set_snapshots(sosf_sequence *seq, int on);
int sosf_sequence_get_snapshot(sosf_sequence *seq, int which, int capacity, double *out, int *count);

/* direct access to the underlying context / backend handles (tracker tests share the frame store) */
sos_ctx *sosf_ctx(sosf_system *sys);
sos_ba *sosf_ba(sosf_system *sys);
int sosf_frame_slot(sosf_system *sys, int frameIdx);

#ifdef __cplusplus
}
#endif
#endif
