// sos_host.hpp -- C++ host facade keeping the reference's class surface around the HIP backend.
//
//   CalibHessian / FrameHessian / PointHessian      FS/HessianBlocks.h:136-649
//   FrameFramePrecalc                               FS/HessianBlocks.h:109-134, .cpp:431-461
//   PointFrameResidual                              FS/Residuals.h:49-93
//   EFFrame / EFPoint / EFResidual, EnergyFunctional  OB/EnergyFunctionalStructs.h:43-134, OB/EnergyFunctional.h:52-154
//   FullSystem (backend-facing subset)              FS/FullSystem.h:118-288
//
// Graph mutation, FEJ bookkeeping, priors, the marginalisation prior HM/bM, the <= (4+8n)-dim fp64
// solve and frame marginalisation stay on the host (they are O(n^2..n^3) on tiny dense matrices); the
// per-residual / per-point loops run on the device through the C-ABI of include/sos_slam.h.
// IMU / spline factors (OB/EnergyFunctional.cpp:256-494) are out of scope of this round.
#pragma once

#include <cstdint>
#include <list>
#include <map>
#include <vector>

#include "../../../include/sos_slam.h"
#include "../../../include/sos_slam_host.h"
#include "sos_math.hpp"

namespace sos {

typedef std::vector<double> VecX;
typedef std::vector<double> MatXX;  // row-major, dim x dim

struct AffLight {  // util/NumType.h:149-171
  double a = 0, b = 0;
  AffLight() {}
  AffLight(double a_, double b_) : a(a_), b(b_) {}
  static void fromToVecExposure(float exposureF, float exposureT, AffLight g2F, AffLight g2T, double *out2);
};

struct FrameHessian;
struct PointHessian;
struct PointFrameResidual;
struct EFFrame;
struct EFPoint;
struct EFResidual;
class EnergyFunctional;

enum ResState { IN = 0, OOB = 1, OUTLIER = 2 };
enum EFPointStatus { PS_GOOD = 0, PS_MARGINALIZE, PS_DROP };

struct CalibHessian {  // FS/HessianBlocks.h:426-553
  double value_zero[4], value_scaled[4], value[4], step[4], value_backup[4], value_minus_value_zero[4];
  float value_scaledf[4], value_scaledi[4];
  CalibHessian();
  void setValue(const double *v);
  void setValueScaled(const double *vs);
  float fxl() const { return value_scaledf[0]; }
  float fyl() const { return value_scaledf[1]; }
  float cxl() const { return value_scaledf[2]; }
  float cyl() const { return value_scaledf[3]; }
  sos_calib toCalib() const;
};

struct FrameHessian;
struct FrameFramePrecalc {  // FS/HessianBlocks.h:109-134
  sos_precalc dev;  // PRE_KRKiTll, PRE_KtTll, PRE_RTll_0, PRE_tTll_0, PRE_aff_mode, PRE_b0_mode
  float PRE_RTll[9], PRE_RKiTll[9], PRE_tTll[3];
  float distanceLL;
  const FrameHessian *hostPtr = nullptr, *targetPtr = nullptr;
  int hostEvalVersion = -1, targetEvalVersion = -1;  // the evalPT (FEJ) products only change with setEvalPT
  void set(const FrameHessian *host, const FrameHessian *target, const CalibHessian *HCalib);
};

struct FrameHessian {
  EFFrame *efFrame = nullptr;
  int frameID = -1;
  int idx = 0;
  int slot = -1;  // device image slot (sos_ctx frame store)
  float frameEnergyTH = 8 * 8 * SOS_PATTERN_NUM;
  float ab_exposure = 1;
  bool flaggedForMarginalization = false;
  int numImmature = 0;  // immaturePoints.size(): the immature points live with the caller (records of sos_immature)
  std::vector<PointHessian *> pointHessians, pointHessiansMarginalized, pointHessiansOut;
  // the residuals that observe a point IN this frame (target == this): marginalizeFrame drops exactly these (FS/FullSystemMarginalize.cpp:148-176)
  // without walking every point of the window.  Kept by PointFrameResidual::registerTarget / its destructor.
  std::vector<PointFrameResidual *> targetedBy;
  SE3 camToWorld_evalPT;
  SE3 worldToCam_evalPT;  // cached inverse
  int evalVersion = 0;    // bumped by setEvalPT
  double state_zero[10], state_scaled[10], state[10], step[10], state_backup[10];
  SE3 PRE_camToWorld, PRE_worldToCam;
  std::vector<FrameFramePrecalc> targetPrecalc;

  FrameHessian();
  ~FrameHessian();
  AffLight aff_g2l() const { return AffLight(state_scaled[6], state_scaled[7]); }
  AffLight aff_g2l_0() const { return AffLight(state_zero[6] * SOS_SCALE_A, state_zero[7] * SOS_SCALE_B); }
  void setState(const double *s);
  void setStateZero(const double *s);
  void setEvalPT(const SE3 &camToWorld, const double *s);
  void getPrior(double *p10, float affineOptModeA, float affineOptModeB) const;
};

struct PointHessian {  // FS/HessianBlocks.h:556-649
  EFPoint *efPoint = nullptr;
  float color[SOS_PATTERN_NUM], weights[SOS_PATTERN_NUM];
  float u = 0, v = 0;
  FrameHessian *host = nullptr;
  bool hasDepthPrior = false;
  float idepth_scaled = 0, idepth_zero_scaled = 0, idepth_zero = 0, idepth = 0, step = 0, idepth_backup = 0;
  float idepth_hessian = 0, maxRelBaseline = 0;
  int numGoodResiduals = 0;
  std::vector<PointFrameResidual *> residuals;
  std::pair<PointFrameResidual *, ResState> lastResiduals[2];
  bool wasMarginalized = false;  // flagPointsForRemoval: PS_MARGINALIZE (vs PS_DROP)
  int packIdx = -1;  // index in the last device snapshot (allPoints order)
  int userIdx = -1;  // running index in which the point was added through the flat API
  ~PointHessian();
  void setIdepth(float id) { idepth = id; idepth_scaled = SOS_SCALE_IDEPTH * id; }
  void setIdepthZero(float id) { idepth_zero = id; idepth_zero_scaled = SOS_SCALE_IDEPTH * id; }
};

struct PointFrameResidual {  // FS/Residuals.h:49-93
  EFResidual *efResidual = nullptr;
  ResState state_state = IN;
  double state_energy = 0;
  ResState state_NewState = OUTLIER;
  double state_NewEnergy = 0, state_NewEnergyWithOutlier = -1;
  PointHessian *point = nullptr;
  FrameHessian *host = nullptr, *target = nullptr;
  bool isNew = true;
  float centerProjectedTo[3] = {0, 0, 0};
  int packIdx = -1;
  int idxInTarget = -1;  // position in target->targetedBy (-1: not registered, or the target frame is gone)
  void registerTarget();
  ~PointFrameResidual();
  void resetOOB() {
    state_NewEnergy = state_energy = 0;
    state_NewState = OUTLIER;
    state_state = IN;
  }
};

struct EFResidual {  // OB/EnergyFunctionalStructs.h:43-81
  PointFrameResidual *data;
  int hostIDX = 0, targetIDX = 0;
  EFPoint *point;
  EFFrame *host, *target;
  int idxInAll = 0;
  uint64_t connKey = 0;  // (host frameID << 32) + target frameID, kept here: the target's EFFrame may be gone when the residual is dropped
  std::pair<int, int> *connEntry = nullptr;  // connectivityMap[connKey], looked up once at insertion
  bool isLinearized = false;
  bool isActiveAndIsGoodNEW = false;
  bool isActive() const { return isActiveAndIsGoodNEW; }
  EFResidual(PointFrameResidual *org, EFPoint *p, EFFrame *h, EFFrame *t) : data(org), point(p), host(h), target(t) {}
};

struct EFPoint {  // OB/EnergyFunctionalStructs.h:83-114
  PointHessian *data;
  float priorF = 0, deltaF = 0;
  int idxInPoints = 0;
  EFFrame *host;
  std::vector<EFResidual *> residualsAll;
  float HdiF = 0, bdSumF = 0;
  EFPointStatus stateFlag = PS_GOOD;
  EFPoint(PointHessian *d, EFFrame *h, float idepthFixPrior);
};

struct EFFrame {  // OB/EnergyFunctionalStructs.h:116-134
  double prior[8], delta_prior[8], delta[8];
  std::vector<EFPoint *> points;
  FrameHessian *data;
  int idx = 0;
  int frameID = -1;
  EFFrame(FrameHessian *d, float modeA, float modeB);
  void takeData(float modeA, float modeB);
};

class EnergyFunctional {  // OB/EnergyFunctional.h:52-154
 public:
  EnergyFunctional(sos_ctx *ctx, const sos_params &prm);
  ~EnergyFunctional();

  EFResidual *insertResidual(PointFrameResidual *r);
  EFFrame *insertFrame(FrameHessian *fh, CalibHessian *HCalib);
  EFPoint *insertPoint(PointHessian *ph);
  void dropResidual(EFResidual *r);
  int marginalizeFrame(EFFrame *fh);
  void removePoint(EFPoint *pt);
  int marginalizePointsF();
  void dropPointsF();
  int solveSystemF(int iteration, double lambda, CalibHessian *HCalib, bool deferResubstitute = false);
  double calcMEnergyF();
  double calcLEnergyF_MT();
  void makeIDX();
  void setDeltaF(CalibHessian *HCalib, bool points = true);
  MatXX scrHA, scrHsc;  // solveSystemF scratch
  bool keepSystem = false;  // sosf_keep_last_system: solveSystemF keeps a copy of what it assembled
  MatXX keptH, keptHsc;
  VecX keptb, keptbsc;
  VecX scrbA, scrbsc, scrPriorRhs;
  void setAdjointsF(CalibHessian *HCalib);

  // device snapshot management
  // sos_ba_set_window from the current graph (when dirty).  active != nullptr: the same walk collects the residuals that are not
  // linearised and resets them (FS/FullSystemOptimize.cpp:316-329)
  int packWindow(std::vector<PointFrameResidual *> *active = nullptr);
  int pushState(CalibHessian *HCalib, bool adjoints, bool points = true);  // sos_ba_set_state
  bool pointsOnDeviceCurrent = false;  // the device's idepth / idepth_zero / deltaF equal the host mirrors
  std::vector<sos_point> packPts;      // record vectors of packWindow / pushState, kept between keyframes
  std::vector<sos_resid> packRes;
  std::vector<sos_precalc> scrPrecalc;
  std::vector<float> scrId, scrIdz, scrDl;
  sos_ba *ba = nullptr;
  sos_ctx *ctx = nullptr;
  bool packDirty = true;
  bool structDirty = true;                 // ... by more than dropped residuals / points since the last pack
  std::vector<int32_t> droppedSincePack;   // snapshot indices of the residuals dropped since the last pack
  bool syncDropsToDevice();

  std::vector<EFFrame *> frames;
  int nPoints = 0, nFrames = 0, nResiduals = 0;
  MatXX HM;
  VecX bM;
  int resInA = 0, resInL = 0, resInM = 0;
  VecX lastX;
  std::map<uint64_t, std::pair<int, int>> connectivityMap;
  std::vector<EFPoint *> allPoints;
  std::vector<EFResidual *> allResiduals;  // packing order
  std::vector<float> pointStep;            // last resubstitute result, packing order

  // multi-GPU hooks (see include/sos_slam_host.h)
  void (*allreduceHook)(void *, float *, size_t) = nullptr;
  // IMU / spline factors (sos_imu.cpp): when set, solveSystemF takes the IMU branch of OB/EnergyFunctional.cpp:1053-1171
  // with these caller-owned records (poses refreshed here every solve, states and scale stepped after it) and the prior
  // in the expanded dimension
  const sosf_imu_settings *imuSettings = nullptr;
  sosf_imu_calib *imuCalib = nullptr;
  sosf_imu_frame *imuFrames = nullptr;
  const double *imuHM = nullptr, *imuBM = nullptr;
  // setting_enable_imu with the prior kept here: HM / bM in the expanded dimension CPARS + 1 + 29 n, grown by insertFrame
  // (OB/EnergyFunctional.cpp:666-677), fed by marginalizePointsF through expandHbtoFitImu (:928-932) and reduced by the IMU
  // form of marginalizeFrame (:733-889).  The 8-per-keyframe HM / bM above stay maintained as the visual-only prior.
  bool imuOwnPrior = false;
  MatXX HMi;
  uint64_t imuCallerPriorName = 0;  // counts sosf_set_imu: the name of a caller-owned prior
  uint64_t imuPriorVersion = 1;  // counts the writes of HMi: the name under which the cached IMU solve keeps its factor (sos_imu.cpp)
  VecX bMi;
  void imuAdoptPrior();  // HMi / bMi = expandHbtoFitImu(HM, bM)
  std::list<std::vector<double>> imuMergedSamples;  // sample lists merged by marginalizeFrame, kept while a record points at them
  double imuScaleStep = 0;
  std::vector<double> imuStep;
  float (*nthHook)(void *, const float *, int, float) = nullptr;
  void (*allreduceF64Hook)(void *, double *, size_t) = nullptr;  // host fp64 sum over ranks (keyframe-rate exchanges)
  void *hookUser = nullptr;
  bool commAttached = false;  // sos_ba_set_comm holds an RCCL communicator: sos_ba_set_window is then a collective
  bool multiRank() const { return commAttached || allreduceHook != nullptr; }
  // sum of a host fp64 buffer over all ranks: native communicator, else the fp64 hook; SOS_ERR_STATE when the system runs
  // on callback hooks without one (the fp32 hook cannot carry the prior update)
  int allreduceF64(double *buf, size_t count);

  sos_params prm;
  float cDeltaF[4];
  double cPrior[4];
  std::vector<double> adHost, adTarget;    // n*n*64
  std::vector<float> adHostF, adTargetF, adHTdeltaF;
  bool EFAdjointsValid = false, EFIndicesValid = false, EFDeltaValid = false;

 private:
  VecX getStitchedDeltaF() const;
  std::vector<EFPoint *> allPointsToMarg;
};

// backend-facing part of FullSystem (FS/FullSystem.h:118-288)
class FullSystem {
 public:
  FullSystem(const sos_params &prm, int device, void *stream);
  ~FullSystem();
  bool ok() const { return ctx != nullptr && ef != nullptr; }

  FrameHessian *addFrame(const double *camToWorld12, const double *state10, float ab_exposure, int frameID,
                         float frameEnergyTH, const float *image, int haveSlot = -1);
  PointHessian *addPoint(const sos_point &p);
  PointFrameResidual *addResidual(PointHessian *ph, FrameHessian *target, const sos_resid &r);

  float optimize(int mnumOptIts, int *iterations);            // FS/FullSystemOptimize.cpp:305-489
  int prepare();                                              // :316-344
  bool gnIteration(int iteration, bool mayContinue = false);  // :358-413
  sos_comm *comm = nullptr;     // RCCL communicator attached to the backend (multi-GPU), not owned
  bool pipelineAlways = false;  // flat API: the caller iterates regardless of canbreak
  // device-resident Gauss-Newton loop (sos_ba_gn_resident_*): the solve, the frame step and the precalc records on the device
  bool residentAllowed = false;  // sosf_set_resident (off by default: measured slower than the host solve)
  bool residentActive = false;
  // device-side step of the fused loop (sos_ba_gn_devstep_begin): the device derives poses / precalc / deltas from x; the host
  // keeps stepping its own states but neither computes nor stages the n^2 precalc records inside the loop
  bool devStepAllowed = true, devStepActive = false;
  // Lazy point mirrors of the device-step loop: inside optimize() (and in a pipelined flat-API loop) the per-iteration point part of
  // doStepFromBackup (FS/FullSystemOptimize.cpp:207-213) and of backupState (:260-269) runs on flat arrays in snapshot order -- same
  // float operations in the same order -- and the PointHessian objects are brought up to date once, by flushPointMirrors(), before
  // anything reads them (it is called wherever residentFlush() is, and by every member that reads or writes the mirrors).
  bool inOptimizeLoop = false, pointMirrorsStale = false, flatStepped = false;
  std::vector<float> flatIdepth, flatBackup;
  std::vector<int> flatOrder;  // snapshot index of the j-th point in the order frames -> pointHessians (the reference's summation order)
  bool beginLazyPointMirrors();
  void flushPointMirrors();
  bool devStepUsable() const;
  int devStepBegin();
  int residentSeq = 0;          // sequence number of the iteration whose results the host has consumed
  int residentQueued = 0;       // ... and of the last one enqueued
  bool residentUsable() const;
  int residentBegin();
  bool residentConsume(int seq);            // waits for iteration seq, refreshes the host mirrors, returns canbreak
  int residentFlush();                      // leaves the loop: mirrors of points / thresholds brought up to date
  int minOptIterations = 1;     // setting_minOptIterations (util/settings.cpp:77)
  bool forceAcceptStep = true;  // setting_forceAceptStep (util/settings.cpp:117); false: energy-checked steps with loadSateBackup
  int stepsRejected = 0;        // rejected steps of the last optimize()
  int lastLoopMode = -1;        // how the last Gauss-Newton iteration ran: 0 host step, 1 device-side step, 2 device-resident loop, 3 energy-checked
  void loadSateBackup();                                      // FS/FullSystemOptimize.cpp:271-287
  bool gnIterationChecked(int iteration, double &lastE, double &lastEL, double &lastEM);  // :358-413, forceAceptStep off
  void setPrecalcValues(bool points = true);                  // FS/FullSystem.cpp:1099-1107
  bool hostPrecalcStale = false;  // the device-side step moved the states: targetPrecalc is refreshed on the next read
  void ensureHostPrecalc();
  void removeOutliers();                                      // FS/FullSystemOptimize.cpp:507-526
  int marginalizePoints(const std::vector<PointHessian *> &pts, bool alreadyDetached = false);  // flagPointsForRemoval core + marginalizePointsF
  void flagFramesForMarginalization();                        // FS/FullSystemMarginalize.cpp:53-133
  int addResidualsToNewestFrame();                            // FS/FullSystem.cpp:818-832
  PointHessian *addActivatedPoint(const sos_point &p, uint32_t inMask);  // FS/FullSystemOptPoint.cpp:151-185
  int flagPointsForRemoval(int *nMarg, int *nDrop);           // FS/FullSystem.cpp:535-614 + :909, :912
  int marginalizeFlaggedFrames(int cap, int32_t *frameIDs, double *camToWorld12, int *count);  // :926-931
  int dropPoints(const std::vector<PointHessian *> &pts);
  int marginalizeFrame(FrameHessian *fh);                     // FS/FullSystemMarginalize.cpp:143-236

  sos_params prm;
  sos_ctx *ctx = nullptr;
  EnergyFunctional *ef = nullptr;
  CalibHessian HCalib;
  std::vector<FrameHessian *> frameHessians;
  std::vector<PointHessian *> userPoints;  // in the order they were added through the flat API
  bool isLost = false;
  int lastError = 0;
  bool slotUsed[SOS_MAX_SLOTS];

 private:
  double linearizeAll(bool fixLinearization);                 // :125-182
  void setNewFrameEnergyTH();                                 // :84-124
  void setNewFrameEnergyTH(std::vector<float> &energiesOfNewestFrame);
  void applyRes();                                            // :79-83
  void backupState();                                         // :260-269
  float backupSumNID = 0, backupNumID = 0;
  double prepareEnergy = 0, prepareEnergyL = 0, prepareEnergyM = 0;
  int rcAcc(int rc) { if (rc != SOS_OK && lastError == SOS_OK) lastError = rc; return rc; }
  bool doStepFromBackup(float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD,
                        bool pointsOnDevice = false, bool precalcOnDevice = false);  // :185-257
  void solveSystem(int iteration, double lambda);             // :491-497
  std::vector<PointFrameResidual *> activeResiduals;
  std::vector<uint8_t> h_newState;
  std::vector<float> h_newEnergy, h_newEnergyWO, h_center, newestE;
};

// FS/CoarseTracker.h:27-48 + FS/ScaleOptimizer.h:43-104: host control flow (LM loops, SE3 updates, 8x8 / scalar
// solves) around the device primitives sos_tracker_calc_res / calc_gs / calc_res_scale / calc_gs_scale.
class CoarseTracker {
 public:
  CoarseTracker(sos_ctx *ctx, const sos_params &prm);
  ~CoarseTracker();
  bool ok() const { return trk != nullptr; }
  void makeK(const CalibHessian *HCalib);                          // FS/ScaleOptimizer.cpp:95-118
  // makeCoarseDepthL0 inputs collected as FS/CoarseTracker.cpp:62-79 does
  int setCoarseTrackingRef(const std::vector<FrameHessian *> &frameHessians);   // :232-242
  int setCoarseTrackingRefRaw(const FrameHessian *lastRef, int npts, const float *u, const float *v, const float *idepth,
                              const float *hdi);
  bool trackNewestCoarse(int newSlot, float new_ab_exposure, SE3 &lastToNew_out, AffLight &aff_g2l_out, int coarsestLvl,
                         const double *minResForAbort5, double *lastResiduals5);   // :366-552
  float optimizeScale(int stereoSlot, const SE3 &tfmF0ToF1, const float *K1_level0, float &scale, int coarsestLvl);  // FS/ScaleOptimizer.cpp:120-230
  // FullSystem::optimizeScale (FS/FullSystem.cpp:1117-1177): until the scale is trapped the seven guesses {0.1 ... 10} are optimised
  // (side by side in one launch of the device loop) and the one with the smallest positive error wins; afterwards one run from
  // the tracking reference's scale.  state = the function's scaleTrapped / static scale_opt_fails.  Returns the new scale, or -1
  // when the error is not below `thres` (setting_scale_opt_thres)
  struct ScaleOptState {
    int scaleTrapped = 0, fails = 0;
  };
  float optimizeScaleKF(int stereoSlot, const SE3 &tfmF0ToF1, const float *K1_level0, float trackingRefScale, int coarsestLvl, float thres,
                        ScaleOptState &state, float *scale_error_out);
  // n initial scales -> n optimised scales and their optimizeScale() return values
  int optimizeScaleHyp(int stereoSlot, const SE3 &tfmF0ToF1, const float *K1_level0, int n, float *scales, float *errors, int coarsestLvl);
  // the pose hypotheses of FullSystem::trackNewCoarse (FS/FullSystem.cpp:150-213): IMU prediction (optional), constant /
  // double / half / zero motion, zero motion from the keyframe, then 26 rotation signs x 3 magnitudes around the constant-
  // motion guess; one identity try when a pose is not valid
  static void makeTrackTries(const SE3 &slast_2_sprelast, const SE3 &lastF_2_slast, const SE3 *lastF_2_fh_imu, bool posesValid,
                             std::vector<SE3> &tries);
  // the loop over the hypotheses, FS/FullSystem.cpp:219-262.  The tries are evaluated `batch` at a time in one launch of the
  // device loop (the first one alone: it usually wins) and the reference's sequential decisions -- each try must be at
  // least as good as the best so far on every level it reaches, the loop stops at the first result below
  // lastCoarseRMSE[0] * setting_reTrackThreshold -- are replayed on the results in order; a try that the sequential loop
  // would have aborted on a coarse level is cut at that level from the list of levels it finished.
  struct TrackResult {
    SE3 lastF_2_fh;
    AffLight aff_g2l;
    double achievedRes[5];
    double flowVecs[3];
    int tryIterations = 0, chosen = -1, evaluated = 0;
    bool haveOneGood = false;
  };
  int trackHypotheses(int newSlot, float new_ab_exposure, const std::vector<SE3> &tries, const AffLight &aff_last_2_l, int coarsestLvl,
                      const double *lastCoarseRMSE5, double reTrackThreshold, int batch, TrackResult &out);
  bool finishTrack(const sos_track_hyp &h, int nvis, SE3 &lastToNew_out, AffLight &aff_g2l_out, double *lastResiduals5);
  float new_ab_exposure_last = 1;
  bool deviceLM = true;  // trackNewestCoarse / poseEstimate as one launch (sos_tracker_track); false: the LM loop on the host
  int lastEvals = 0;     // residual evaluations of the last trackNewestCoarse
  int lmFallbacks = 0;   // launches of the device loop that came back SOS_ERR_TIMEOUT and were redone with the host loop
  void scaleCoarseDepthL0(float scale);
  // loop-closure aligner (src/LoopClosure/PoseEstimator.cpp): template = 3-D points of the matched keyframe with one colour per
  // level; estimate() = the same Levenberg-Marquardt loop with zero reference affine parameters, no abort thresholds and
  // PoseEstimator's three acceptance tests (:455-485)
  int setPoints3d(const sos_calib &cam, float matched_ab_exposure, int n, const float *xyz, const float *colors);
  bool poseEstimate(int newSlot, float new_ab_exposure, SE3 &refToNew, int coarsestLvl, float loopDirectThres, int innerPercent,
                    float *poseError, int *inlierPercent);

  sos_tracker *trk = nullptr;
  sos_ctx *ctx = nullptr;
  sos_params prm;
  int levels = 1;
  float fx[SOS_PYR_LEVELS], fy[SOS_PYR_LEVELS], cx[SOS_PYR_LEVELS], cy[SOS_PYR_LEVELS], Ki[SOS_PYR_LEVELS][9];
  sos_calib calib;
  int pc_n[SOS_PYR_LEVELS];
  // outputs the reference exposes as fields (FS/CoarseTracker.h:41-48)
  int refFrameID = -1;
  float ref_ab_exposure = 1;
  AffLight lastRef_aff_g2l;
  double lastFlowIndicators[3] = {1000, 1000, 1000};
  double firstCoarseRMSE = -1;
  int lastInners[SOS_PYR_LEVELS] = {0};
  int loopPoints = 0;
};

}  // namespace sos

// the FullSystem behind a facade handle (for the other translation units of the facade)
sos::FullSystem *sosf_system_full(sosf_system *s);
