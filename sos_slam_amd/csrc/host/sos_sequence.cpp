// sos_sequence.cpp -- the frame-rate loop of FullSystem in C++ (visual part):
//
//   FullSystem::addActiveFrame      FS/FullSystem.cpp:616-766   track -> traceNewCoarse -> keyframe decision -> deliverTrackedFrame
//   FullSystem::traceNewCoarse      :311-361                    every immature point of every keyframe against the new frame
//   FullSystem::makeKeyFrame        :783-931                    flagFramesForMarginalization, insertFrame, residuals of the old points,
//                                                               activatePointsMT, optimize, removeOutliers, setCoarseTrackingRef,
//                                                               flagPointsForRemoval + marginalizePointsF, makeNewTraces, marginalizeFrame
//   FullSystem::activatePointsMT    :375-531                    currentMinActDist, distance map + candidate loop (sosf_activate_select),
//                                                               optimizeImmaturePoint (sos_immature_activate), removal from immaturePoints
//   FullSystem::makeNewTraces       :1071-1097                  PixelSelector::makeMaps + new ImmaturePoint per selected pixel
//
// Every stage is a call that already existed (facade C++ classes, device C-ABI); what is new is the loop that strings them together
// and the per-keyframe containers of the immature points (FrameHessian::immaturePoints).  The IMU / stereo branches of makeKeyFrame
// (:800-807, 841-849, 878-903; FS/FullSystemOptimize.cpp:437-479) are strung in behind sosf_sequence_enable_imu / _enable_stereo and
// sosf_add_active_frame_ex.  Out of scope: the initialiser (CoarseInitializer) -- the first window is handed over (sosf_sequence_bootstrap).
//
// PENDING_FIRST_GPU_RUN: written while GPU access was withdrawn (round 3); tests/test_gpu_sequence_driver.py is its acceptance test (green under tests/emu
// in round 4, where it also drives the visual rolling-window tests).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "sos_host.hpp"

using namespace sos;

struct sosf_sequence {
  FullSystem *fs = nullptr;
  CoarseTracker *ct = nullptr;
  sos_pixsel *sel = nullptr;
  sosf_sequence_params prm;
  int currentPotential = 3;          // PixelSelector::currentPotential
  float currentMinActDist = 2.0f;    // FullSystem::currentMinActDist
  // FrameHessian::immaturePoints (+ ImmaturePoint::my_type) per keyframe id, in the reference's container order
  std::map<int, std::vector<sos_immature>> imm;
  std::map<int, std::vector<float>> immType;
  bool haveLastRel = false;
  SE3 lastRel;                       // last frame's refToNew (constant-motion guess when every frame is a keyframe)
  std::vector<SE3> trackHist;        // camToWorld of the last tracked frames
  int framesSinceKF = 0;
  double lastCoarseRMSE0 = -1;
  // ---- visual-inertial branch (setting_enable_imu): per keyframe the FrameShell fields, the 21 IMU states (unscaled) with their
  // linearisation point and FrameHessian::imu_data; the IMU part of CalibHessian; FullSystem::imu_data (samples since the last keyframe)
  struct VioFrame {
    double ts = 0, c2w[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, vel[3] = {0, 0, 0}, state[21] = {0}, zero[21] = {0};
    std::vector<double> imu;  // n x 7
    int trackRefId = -1;      // frameID of shell->trackingRef (coarseTracker->lastRef when the frame was tracked, FS/FullSystem.cpp:296; -1: none)
  };
  bool vio = false;
  sosf_imu_settings S;
  sosf_imu_calib cal = {1.0 / 200.0, 1.0 / 200.0, 0, 0};
  std::map<int, VioFrame> vf;
  std::vector<sosf_imu_frame> winRecs;  // the records the facade's solve reads (one per keyframe of the window, renewed before use)
  std::vector<double> pendingImu;
  double scaleQueue[10] = {-10, -20, -30, -40, -50, -60, -70, -80, -90, -100};  // CalibHessian's ring, LinSpaced(10, -10, -100)
  int32_t scaleQi = 0;
  int nKfTotal = 0;
  // ---- stereo branch (FullSystem::optimizeScale on the stereo partner of every keyframe)
  bool stereo = false;
  SE3 stereoTfm;
  float scaleOptThres = 0;
  CoarseTracker::ScaleOptState scaleState;
  // ---- what the parity tests look at BETWEEN the stages of makeKeyFrame (sosf_sequence_set_snapshots): 0 keyframes flagged for
  // marginalisation (frameID), 1 activated points (host frameID, u, v) in insertion order, 2 residuals after optimize() (host frameID,
  // u, v, target frameID), 3 points after flagPointsForRemoval (host frameID, u, v)
  bool snapshots = false;
  std::vector<double> snap[4];
};

namespace {

inline float f32(double v) { return (float)v; }

// KRKi, Kt, aff of FullSystem::traceNewCoarse (FS/FullSystem.cpp:326-336), exposures as given
void host_to_frame(const float K4[4], const SE3 &host_c2w, const SE3 &frame_c2w, const AffLight &hostAff, float hostExp, const AffLight &frameAff,
                   float frameExp, float *KRKi9, float *Kt3, float *aff2) {
  const SE3 T = frame_c2w.inverse() * host_c2w;
  float R[9], t[3];
  for (int i = 0; i < 9; i++) R[i] = f32(T.R[i]);
  for (int i = 0; i < 3; i++) t[i] = f32(T.t[i]);
  const float K[9] = {K4[0], 0, K4[2], 0, K4[1], K4[3], 0, 0, 1};
  const float Ki[9] = {1.0f / K4[0], 0, -K4[2] / K4[0], 0, 1.0f / K4[1], -K4[3] / K4[1], 0, 0, 1};
  float KR[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) KR[3 * i + j] = K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j] + K[3 * i + 2] * R[6 + j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) KRKi9[3 * i + j] = KR[3 * i] * Ki[j] + KR[3 * i + 1] * Ki[3 + j] + KR[3 * i + 2] * Ki[6 + j];
  for (int i = 0; i < 3; i++) Kt3[i] = K[3 * i] * t[0] + K[3 * i + 1] * t[1] + K[3 * i + 2] * t[2];
  double a2[2];
  AffLight::fromToVecExposure(hostExp, frameExp, hostAff, frameAff, a2);
  aff2[0] = f32(a2[0]);
  aff2[1] = f32(a2[1]);
}

AffLight frame_aff(const FrameHessian *fh) { return AffLight(fh->state[6] * SOS_SCALE_A, fh->state[7] * SOS_SCALE_B); }

// FullSystem::makeNewTraces (FS/FullSystem.cpp:1071-1097) for the keyframe `fh`
int make_new_traces(sosf_sequence *q, FrameHessian *fh) {
  sos_ctx *ctx = q->fs->ctx;
  int32_t nsel = 0, pot = q->currentPotential;
  int rc = sos_pixsel_make_maps(q->sel, fh->slot, q->prm.immatureDensity, 1, 1.0f, &pot, nullptr, &nsel);
  if (rc != SOS_OK) return -1;
  q->currentPotential = pot;
  int32_t cnt = 0;
  const int cap = 1 << 16;
  std::vector<int32_t> u(cap), v(cap);
  std::vector<float> ty(cap);
  rc = sos_pixsel_list(q->sel, q->prm.patternPadding, cap, u.data(), v.data(), ty.data(), &cnt);
  if (rc != SOS_OK) return -1;
  const int k = std::min<int>(cnt, cap);
  std::vector<sos_immature> rec(k);
  if (k > 0 && sos_immature_init(ctx, &q->prm.trace, fh->slot, k, u.data(), v.data(), rec.data()) != SOS_OK) return -1;
  std::vector<sos_immature> &dst = q->imm[fh->frameID];
  std::vector<float> &dty = q->immType[fh->frameID];
  dst.clear();
  dty.clear();
  for (int i = 0; i < k; i++)
    if (std::isfinite(rec[i].energyTH)) {  // if (!std::isfinite(impt->energyTH)) delete impt; (:1090-1093)
      dst.push_back(rec[i]);
      dty.push_back(ty[i]);
    }
  fh->numImmature = (int)dst.size();
  return (int)dst.size();
}

// FullSystem::traceNewCoarse: ph->traceOn(fh, KRKi, Kt, aff, ...) for every immature point of every keyframe
int trace_new_coarse(sosf_sequence *q, int slot, const SE3 &c2w, const AffLight &aff, float ab_exposure) {
  FullSystem *fs = q->fs;
  float K4[4] = {fs->HCalib.fxl(), fs->HCalib.fyl(), fs->HCalib.cxl(), fs->HCalib.cyl()};
  for (FrameHessian *host : fs->frameHessians) {
    std::vector<sos_immature> &v = q->imm[host->frameID];
    if (v.empty()) continue;
    float KRKi[9], Kt[3], a2[2];
    host_to_frame(K4, host->PRE_camToWorld, c2w, frame_aff(host), host->ab_exposure, aff, ab_exposure, KRKi, Kt, a2);
    const int rc = sos_immature_trace(fs->ctx, &q->prm.trace, slot, (int)v.size(), v.data(), KRKi, Kt, a2);
    if (rc != SOS_OK) return rc;
  }
  return SOS_OK;
}

// FullSystem::activatePointsMT (FS/FullSystem.cpp:375-531); flaggedOld = flags of the keyframes before the new one was inserted
int activate_points(sosf_sequence *q, int *nActivated, int *nDeleted) {
  FullSystem *fs = q->fs;
  const int n = (int)fs->frameHessians.size(), newest = n - 1;
  q->currentMinActDist = sosf_next_min_act_dist(q->currentMinActDist, fs->ef->nPoints, q->prm.desiredPointDensity);
  const float fx = fs->HCalib.fxl(), fy = fs->HCalib.fyl(), cx = fs->HCalib.cxl(), cy = fs->HCalib.cyl();
  // level-1 projection into the newest keyframe: K[1] R(newest <- f) Ki[0], K[1] t  (:410-415; FS/CoarseTracker.cpp:806-809)
  const float K1[9] = {fx * 0.5f, 0, (cx + 0.5f) / 2 - 0.5f, 0, fy * 0.5f, (cy + 0.5f) / 2 - 0.5f, 0, 0, 1};
  const float Ki0[9] = {1.0f / fx, 0, -cx / fx, 0, 1.0f / fy, -cy / fy, 0, 0, 1};
  std::vector<float> KRKi((size_t)9 * n), Kt((size_t)3 * n);
  const SE3 newestInv = fs->frameHessians[newest]->PRE_camToWorld.inverse();
  for (int f = 0; f < n; f++) {
    const SE3 T = newestInv * fs->frameHessians[f]->PRE_camToWorld;
    float R[9], t[3], KR[9];
    for (int i = 0; i < 9; i++) R[i] = f32(T.R[i]);
    for (int i = 0; i < 3; i++) t[i] = f32(T.t[i]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) KR[3 * i + j] = K1[3 * i] * R[j] + K1[3 * i + 1] * R[3 + j] + K1[3 * i + 2] * R[6 + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) KRKi[9 * f + 3 * i + j] = KR[3 * i] * Ki0[j] + KR[3 * i + 1] * Ki0[3 + j] + KR[3 * i + 2] * Ki0[6 + j];
    for (int i = 0; i < 3; i++) Kt[3 * f + i] = K1[3 * i] * t[0] + K1[3 * i + 1] * t[1] + K1[3 * i + 2] * t[2];
  }
  // active points in frames -> points order
  std::vector<float> au, av, aid;
  std::vector<int32_t> ah;
  for (FrameHessian *fh : fs->frameHessians)
    for (EFPoint *p : fh->efFrame->points) {
      au.push_back(p->data->u); av.push_back(p->data->v); aid.push_back(p->data->idepth_scaled); ah.push_back(fh->idx);
    }
  // candidates: every immature point of every keyframe but the newest, in container order
  std::vector<sos_immature> cand;
  std::vector<int32_t> candHost;
  std::vector<float> candType;
  std::vector<std::pair<int, int>> src;  // (frameID, index in its container)
  std::vector<uint8_t> hostFlagged(n, 0);
  for (int f = 0; f < n; f++) hostFlagged[f] = fs->frameHessians[f]->flaggedForMarginalization ? 1 : 0;
  for (int f = 0; f < newest; f++) {
    const int fid = fs->frameHessians[f]->frameID;
    const std::vector<sos_immature> &v = q->imm[fid];
    const std::vector<float> &ty = q->immType[fid];
    for (size_t j = 0; j < v.size(); j++) {
      cand.push_back(v[j]); candHost.push_back(f); candType.push_back(ty[j]); src.emplace_back(fid, (int)j);
    }
  }
  const int nc = (int)cand.size();
  std::vector<int8_t> dec(nc ? nc : 1, 0);
  int rc = sosf_activate_select(fs->prm.w / 2, fs->prm.h / 2, n, newest, KRKi.data(), Kt.data(), (int)au.size(), au.data(), av.data(), aid.data(), ah.data(),
                                q->currentMinActDist, q->prm.minTraceQuality, nc, cand.data(), candHost.data(), candType.data(), hostFlagged.data(),
                                dec.data(), nullptr);
  if (rc != SOS_OK) return rc;
  std::vector<int> todo;
  for (int i = 0; i < nc; i++)
    if (dec[i] == SOSF_SEL_OPTIMIZE) todo.push_back(i);
  // optimizeImmaturePoint over the chosen candidates
  std::vector<sos_activation> act(todo.size());
  if (!todo.empty()) {
    std::vector<sos_immature> pts(todo.size());
    std::vector<int32_t> ho(todo.size()), slots(n);
    for (size_t k = 0; k < todo.size(); k++) { pts[k] = cand[todo[k]]; ho[k] = candHost[todo[k]]; }
    std::vector<sos_pair_tfm> pairs((size_t)n * n);
    for (int h = 0; h < n; h++)
      for (int t = 0; t < n; t++) {
        const FrameHessian *H = fs->frameHessians[h], *Tg = fs->frameHessians[t];
        const SE3 T = Tg->PRE_camToWorld.inverse() * H->PRE_camToWorld;
        sos_pair_tfm &o = pairs[(size_t)h + (size_t)n * t];
        for (int i = 0; i < 9; i++) o.R[i] = f32(T.R[i]);
        for (int i = 0; i < 3; i++) o.t[i] = f32(T.t[i]);
        double a2[2];
        AffLight::fromToVecExposure(H->ab_exposure, Tg->ab_exposure, frame_aff(H), frame_aff(Tg), a2);
        o.aff[0] = f32(a2[0]); o.aff[1] = f32(a2[1]);
        o.pad[0] = o.pad[1] = 0;
      }
    for (int f = 0; f < n; f++) slots[f] = fs->frameHessians[f]->slot;
    const sos_calib cal = fs->HCalib.toCalib();
    rc = sos_immature_activate(fs->ctx, &q->prm.activate, &cal, n, slots.data(), pairs.data(), (int)todo.size(), pts.data(), ho.data(), act.data());
    if (rc != SOS_OK) return rc;
  }
  std::vector<char> gone(nc ? nc : 1, 0);
  for (int i = 0; i < nc; i++)
    if (dec[i] == SOSF_SEL_DELETE) gone[i] = 1;
  int activated = 0;
  for (size_t k = 0; k < todo.size(); k++) {
    const int j = todo[k];
    const sos_activation &a = act[k];
    if (a.status == SOS_ACT_ACTIVATED) {  // FS/FullSystemOptPoint.cpp:151-185: the PointHessian with its IN residuals
      sos_point p;
      std::memset(&p, 0, sizeof(p));
      p.u = cand[j].u; p.v = cand[j].v;
      p.idepth_scaled = p.idepth_zero_scaled = a.idepth;
      std::memcpy(p.color, cand[j].color, sizeof(p.color));
      std::memcpy(p.weights, cand[j].weights, sizeof(p.weights));
      p.host = candHost[j];
      if (!fs->addActivatedPoint(p, a.inMask)) return SOS_ERR_STATE;
      if (q->snapshots) {
        q->snap[1].push_back(fs->frameHessians[candHost[j]]->frameID);
        q->snap[1].push_back(p.u);
        q->snap[1].push_back(p.v);
      }
      gone[j] = 1;
      activated++;
    } else if (a.status == SOS_ACT_DELETE || cand[j].lastTraceStatus == SOS_IPS_OOB) {  // :493-500
      gone[j] = 1;
    }
  }
  // removal with the reference's swap-from-the-back compaction per host (:519-531)
  int nGone = 0;
  std::map<int, std::vector<char>> alive;
  for (int i = 0; i < nc; i++)
    if (gone[i]) {
      nGone++;
      std::vector<char> &a = alive[src[i].first];
      if (a.empty()) a.assign(q->imm[src[i].first].size(), 1);
      a[src[i].second] = 0;
    }
  for (auto &kv : alive) {
    std::vector<sos_immature> &v = q->imm[kv.first];
    std::vector<float> &ty = q->immType[kv.first];
    std::vector<char> &a = kv.second;
    size_t i = 0;
    while (i < v.size()) {
      if (!a[i]) {
        v[i] = v.back(); ty[i] = ty.back(); a[i] = a.back();
        v.pop_back(); ty.pop_back(); a.pop_back();
        continue;
      }
      i++;
    }
  }
  for (FrameHessian *fh : fs->frameHessians) fh->numImmature = (int)q->imm[fh->frameID].size();
  if (nActivated) *nActivated = activated;
  if (nDeleted) *nDeleted = nGone - activated;
  return SOS_OK;
}

// FullSystem::optimizeScale on the stereo partner in `stereoSlot`; HCalib.setScaleScaledZero when it is accepted (FS/FullSystem.cpp:1117-1177)
int optimize_scale(sosf_sequence *q, int stereoSlot, float *newScale, float *scaleError) {
  FullSystem *fs = q->fs;
  const float K1[4] = {fs->HCalib.fxl(), fs->HCalib.fyl(), fs->HCalib.cxl(), fs->HCalib.cyl()};
  const float refScale = (float)(200.0 * q->cal.scale);  // shell->trackingRef->scale = HCalib.getScaleScaled() of the last optimize
  float err = 0;
  const float ns = q->ct->optimizeScaleKF(stereoSlot, q->stereoTfm, K1, refScale, q->ct->levels - 1, q->scaleOptThres, q->scaleState, &err);
  if (newScale) *newScale = ns;
  if (scaleError) *scaleError = err;
  if (ns > 0) q->cal.scale = q->cal.scale_zero = (double)(1.0f / 200.0f) * (double)ns;
  return SOS_OK;
}

// ---- visual-inertial helpers: records / shells of a keyframe from the sequence's own state
const double kImuScale[21] = {100, 100, 100, 1, 1, 1, 100, 100, 100, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000, 1000};  // SCALE_BA, BG, SL_ROT, SQ_TRANS, SQ_ROT, SC_TRANS, SC_ROT

// prevId = frameID of the keyframe that stands before `fid` in the window NOW (-1: none): spline_valid needs shell->trackingRef to be
// exactly that keyframe's shell (OB/EnergyFunctional.cpp:318, :350), which stops holding once a keyframe between the two was marginalised
sosf_imu_frame vio_record(sosf_sequence *q, int fid, int prevId) {
  const sosf_sequence::VioFrame &v = q->vf[fid];
  sosf_imu_frame f;
  std::memset(&f, 0, sizeof(f));
  f.timestamp = v.ts;
  std::memcpy(f.camToWorld, v.c2w, sizeof(v.c2w));
  std::memcpy(f.evalPT_R, v.c2w, sizeof(double) * 9);
  std::memcpy(f.state_imu, v.state, sizeof(v.state));
  std::memcpy(f.state_imu_zero, v.zero, sizeof(v.zero));
  f.trackingRefIsPrev = (prevId >= 0 && v.trackRefId == prevId) ? 1 : 0;
  f.n_imu = (int32_t)(v.imu.size() / 7);
  f.imu = v.imu.empty() ? nullptr : v.imu.data();
  return f;
}
sosf_imu_shell vio_shell(sosf_sequence *q, int fid) {
  const sosf_sequence::VioFrame &v = q->vf[fid];
  sosf_imu_shell sh;
  sh.timestamp = v.ts;
  std::memcpy(sh.camToWorld, v.c2w, sizeof(v.c2w));
  std::memcpy(sh.velInWorld, v.vel, sizeof(v.vel));
  return sh;
}
// sosf_set_imu(sys, S, calib, records, NULL, NULL): the facade keeps the expanded prior; the records are renewed from the sequence's state
void vio_push(sosf_sequence *q) {
  FullSystem *fs = q->fs;
  q->winRecs.clear();
  int prevId = -1;
  for (FrameHessian *fh : fs->frameHessians) {
    q->winRecs.push_back(vio_record(q, fh->frameID, prevId));
    prevId = fh->frameID;
  }
  EnergyFunctional *ef = fs->ef;
  ef->imuSettings = &q->S; ef->imuCalib = &q->cal; ef->imuFrames = q->winRecs.data(); ef->imuHM = nullptr; ef->imuBM = nullptr;
  if (!ef->imuOwnPrior) ef->imuAdoptPrior();
}
void vio_pull(sosf_sequence *q) {  // the states the solve stepped (doStepFromBackup with unit step factors)
  FullSystem *fs = q->fs;
  for (size_t i = 0; i < fs->frameHessians.size() && i < q->winRecs.size(); i++)
    std::memcpy(q->vf[fs->frameHessians[i]->frameID].state, q->winRecs[i].state_imu, sizeof(double) * 21);
}
int vio_update_vel(sosf_sequence *q, int fid, int lastFid) {  // FrameHessian::updateVel(last_shell)
  const sosf_imu_frame rec = vio_record(q, fid, lastFid);
  sosf_imu_shell sh = vio_shell(q, fid);
  const sosf_imu_shell shl = vio_shell(q, lastFid);
  const int rc = sosf_imu_update_vel(&rec, &sh, &shl);
  if (rc == SOS_OK) std::memcpy(q->vf[fid].vel, sh.velInWorld, sizeof(sh.velInWorld));
  return rc;
}

// FullSystem::makeKeyFrame for the tracked frame in `slot` (FS/FullSystem.cpp:783-931)
int make_keyframe(sosf_sequence *q, int slot, int frameID, int trackRefId, const SE3 &c2w, const AffLight &aff, float ab_exposure,
                  const sosf_frame_extra *extra, sosf_frame_result *out) {
  FullSystem *fs = q->fs;
  for (FrameHessian *fh : fs->frameHessians) fh->numImmature = (int)q->imm[fh->frameID].size();
  fs->flagFramesForMarginalization();  // :798
  if (q->snapshots) {
    for (int w = 0; w < 4; w++) q->snap[w].clear();
    for (FrameHessian *fh : fs->frameHessians)
      if (fh->flaggedForMarginalization) q->snap[0].push_back(fh->frameID);
  }
  if (q->vio) {  // fh->setImuData(imu_data); propagateImuState(allKeyFramesHistory.back(), coarseTracker->lastRef->imu_bias), :800-807
    sosf_sequence::VioFrame &v = q->vf[frameID];
    v = sosf_sequence::VioFrame();
    v.ts = extra ? extra->timestamp : 0.0;
    v.trackRefId = trackRefId;
    c2w.to12(v.c2w);
    v.imu.swap(q->pendingImu);
    q->pendingImu.clear();
    if (q->cal.imu_initialized) {
      const int last = fs->frameHessians.back()->frameID;
      double bias6[6];
      for (int i = 0; i < 6; i++) bias6[i] = kImuScale[i] * q->vf[last].state[i];
      sosf_imu_frame rec = vio_record(q, frameID, last);
      sosf_imu_shell sh = vio_shell(q, frameID);
      const sosf_imu_shell shl = vio_shell(q, last);
      const int rci = sosf_imu_propagate_state(&q->S, &q->cal, &rec, &sh, &shl, bias6);
      if (rci != SOS_OK) return rci;
      std::memcpy(v.state, rec.state_imu, sizeof(v.state));
      std::memcpy(v.zero, rec.state_imu_zero, sizeof(v.zero));
      std::memcpy(v.vel, sh.velInWorld, sizeof(v.vel));
    }
    q->nKfTotal++;
  }
  double c2w12[12], st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  c2w.to12(c2w12);
  st[6] = aff.a / SOS_SCALE_A;
  st[7] = aff.b / SOS_SCALE_B;
  FrameHessian *fh = fs->addFrame(c2w12, st, ab_exposure, frameID, 8 * 8 * SOS_PATTERN_NUM, nullptr, slot);  // :809-816
  if (!fh) return SOS_ERR_STATE;
  q->imm[frameID].clear();
  q->immType[frameID].clear();
  fs->addResidualsToNewestFrame();  // :818-832
  int rc = activate_points(q, &out->nActivated, &out->nDeletedImmature);  // :837
  if (rc != SOS_OK) return rc;
  out->nPointsBeforeOpt = fs->ef->nPoints;
  const bool imuInitNow = q->vio && q->nKfTotal == 5;
  if (imuInitNow) {  // imu initialization on the first five keyframes, :841-849
    const int n5 = (int)fs->frameHessians.size();
    if (n5 != 5) return SOS_ERR_STATE;  // the window must still hold them
    sosf_imu_frame recs[5];
    sosf_imu_shell shs[5];
    for (int i = 0; i < 5; i++) {
      const int fid = fs->frameHessians[i]->frameID;
      recs[i] = vio_record(q, fid, i > 0 ? fs->frameHessians[i - 1]->frameID : -1);
      fs->frameHessians[i]->PRE_camToWorld.to12(recs[i].camToWorld);
      shs[i] = vio_shell(q, fid);
    }
    int ok = 0;
    rc = sosf_imu_initialize(&q->S, &q->cal, recs, shs, &ok);
    if (rc != SOS_OK) return rc;
    if (!ok) return SOS_ERR_STATE;  // "IMU initialization failed"
    for (int i = 0; i < 5; i++) {
      sosf_sequence::VioFrame &v = q->vf[fs->frameHessians[i]->frameID];
      std::memcpy(v.state, recs[i].state_imu, sizeof(v.state));
      std::memcpy(v.zero, recs[i].state_imu_zero, sizeof(v.zero));
      std::memcpy(v.vel, shs[i].velInWorld, sizeof(v.vel));
    }
    q->cal.imu_initialized = 1;
  }
  const bool imuOn = q->vio && q->cal.imu_initialized;
  int its = 0;
  if (imuOn) vio_push(q);
  out->rmse = fs->optimize(q->prm.maxOptIterations, &its);  // :853
  out->iterations = its;
  if (fs->lastError != SOS_OK) return fs->lastError;
  if (imuOn) vio_pull(q);
  if (q->snapshots)
    for (FrameHessian *fh : fs->frameHessians)
      for (EFPoint *p : fh->efFrame->points)
        for (EFResidual *r : p->residualsAll) {
          q->snap[2].push_back(fh->frameID);
          q->snap[2].push_back(p->data->u);
          q->snap[2].push_back(p->data->v);
          q->snap[2].push_back(r->target->frameID);
        }
  if (q->vio) {
    const int nw = (int)fs->frameHessians.size();
    for (FrameHessian *fh : fs->frameHessians) fh->PRE_camToWorld.to12(q->vf[fh->frameID].c2w);  // shell->camToWorld = PRE_camToWorld, FS/FullSystemOptimize.cpp:437-443
    if (imuOn && nw >= 2) {  // :459-479
      const int last = fs->frameHessians[nw - 1]->frameID, prev = fs->frameHessians[nw - 2]->frameID;
      if ((rc = vio_update_vel(q, last, prev)) != SOS_OK) return rc;
      std::memcpy(q->vf[last].zero, q->vf[last].state, sizeof(double) * 21);
      if (q->S.enable_scale_opt) q->cal.scale_trapped = 1;
      if (!q->cal.scale_trapped) {
        sosf_imu_try_trap_scale(&q->cal, q->scaleQueue, &q->scaleQi, 1e-4);
        if (q->cal.scale_trapped)
          for (FrameHessian *fh : fs->frameHessians) std::memcpy(q->vf[fh->frameID].zero, q->vf[fh->frameID].state, sizeof(double) * 21);
      }
    }
  }
  {
    const int before = fs->ef->nPoints;
    fs->removeOutliers();  // :875
    out->nOutliersRemoved = before - fs->ef->nPoints;
  }
  if (imuInitNow) {  // reset imu states for imu initialization, :878-886
    for (size_t i = 0; i < fs->frameHessians.size(); i++) {
      const int fid = fs->frameHessians[i]->frameID;
      std::memcpy(q->vf[fid].zero, q->vf[fid].state, sizeof(double) * 21);
      if (i > 0 && (rc = vio_update_vel(q, fid, fs->frameHessians[i - 1]->frameID)) != SOS_OK) return rc;
    }
  }
  q->ct->makeK(&fs->HCalib);  // :887-895
  rc = q->ct->setCoarseTrackingRef(fs->frameHessians);
  if (rc != SOS_OK) return rc;
  if (q->stereo && extra && extra->stereoSlot >= 0) {  // scale optimization, :897-903
    if ((rc = optimize_scale(q, extra->stereoSlot, &out->newScale, &out->scaleError)) != SOS_OK) return rc;
  }
  rc = fs->flagPointsForRemoval(&out->nMargPoints, &out->nDroppedPoints);  // :908-912
  if (rc != SOS_OK) return rc;
  if (q->snapshots)
    for (FrameHessian *fh : fs->frameHessians)
      for (EFPoint *p : fh->efFrame->points) {
        q->snap[3].push_back(fh->frameID);
        q->snap[3].push_back(p->data->u);
        q->snap[3].push_back(p->data->v);
      }
  out->nNewImmature = make_new_traces(q, fs->frameHessians.back());  // :915
  if (out->nNewImmature < 0) return SOS_ERR_HIP;
  // :923-927 -- the keyframes that leave take their immature points with them
  std::vector<int> flaggedIDs;
  for (FrameHessian *f : fs->frameHessians)
    if (f->flaggedForMarginalization) flaggedIDs.push_back(f->frameID);
  int cnt = 0;
  std::vector<int> win;
  for (FrameHessian *f : fs->frameHessians) win.push_back(f->frameID);
  if (q->vio && q->cal.imu_initialized) vio_push(q);  // the IMU form of marginalizeFrame reads the records of the window as it is now
  rc = fs->marginalizeFlaggedFrames(8, out->margFrameIDs, out->margCamToWorld, &cnt);
  if (rc != SOS_OK) return rc;
  out->nMargFrames = cnt;
  for (int k = 0; k < cnt && q->vio; k++) {  // the samples of a leaving keyframe go in front of its successor's (FS/FullSystemMarginalize.cpp:226-228)
    const int fid = out->margFrameIDs[k];
    const auto it = std::find(win.begin(), win.end(), fid);
    if (it == win.end()) continue;
    if (it + 1 != win.end()) {
      std::vector<double> &nxt = q->vf[*(it + 1)].imu, &cur = q->vf[fid].imu;
      nxt.insert(nxt.begin(), cur.begin(), cur.end());
    }
    win.erase(it);
    q->vf.erase(fid);
  }
  for (int id : flaggedIDs) {
    q->imm.erase(id);
    q->immType.erase(id);
  }
  return SOS_OK;
}

}  // namespace

extern "C" int sosf_sequence_create(sosf_system *s, const sosf_sequence_params *prm, const uint8_t *randomPattern, sosf_sequence **out) {
  if (!s || !prm || !randomPattern || !out) return SOS_ERR_ARG;
  *out = nullptr;
  FullSystem *fs = sosf_system_full(s);
  sosf_sequence *q = new sosf_sequence();
  q->fs = fs;
  q->prm = *prm;
  q->ct = new CoarseTracker(fs->ctx, fs->prm);
  if (!q->ct->ok() || sos_pixsel_create(fs->ctx, &prm->pixsel, randomPattern, &q->sel) != SOS_OK) {
    delete q->ct;
    delete q;
    return SOS_ERR_HIP;
  }
  *out = q;
  return SOS_OK;
}

extern "C" int sosf_sequence_destroy(sosf_sequence *q) {
  if (!q) return SOS_OK;
  if (q->vio && q->fs && q->fs->ef && q->fs->ef->imuSettings == &q->S) {  // the facade must not keep pointers into this object
    EnergyFunctional *ef = q->fs->ef;
    ef->imuSettings = nullptr; ef->imuCalib = nullptr; ef->imuFrames = nullptr;
    ef->imuMergedSamples.clear();
    ef->imuOwnPrior = false;  // (as sosf_set_imu(sys, NULL, ...) leaves it)
  }
  if (q->sel) sos_pixsel_destroy(q->sel);
  delete q->ct;
  delete q;
  return SOS_OK;
}

// the window the initialiser would hand over is in the system already (sosf_add_frame / _points / _residuals): first optimize(),
// removeOutliers, tracking reference, and makeNewTraces on every bootstrap keyframe
extern "C" int sosf_sequence_enable_imu(sosf_sequence *q, const sosf_imu_settings *S, int nBoot, const double *timestamps, const int32_t *n_imu,
                                        const double *const *imu) {
  if (!q || !S || nBoot < 0 || (nBoot && !timestamps)) return SOS_ERR_ARG;
  FullSystem *fs = q->fs;
  if ((int)fs->frameHessians.size() != nBoot) return SOS_ERR_STATE;
  q->vio = true;
  q->S = *S;
  q->vf.clear();
  for (int i = 0; i < nBoot; i++) {  // the shells of the window the initialiser handed over: poses as they are, velocities and states zero
    sosf_sequence::VioFrame &v = q->vf[fs->frameHessians[i]->frameID];
    v.ts = timestamps[i];
    v.trackRefId = i > 0 ? fs->frameHessians[i - 1]->frameID : -1;  // firstFrame->trackingRef = 0, newFrame->trackingRef = firstFrame (FS/FullSystem.cpp:1054-1062)
    fs->frameHessians[i]->PRE_camToWorld.to12(v.c2w);
    if (n_imu && imu && n_imu[i] > 0 && imu[i]) v.imu.assign(imu[i], imu[i] + (size_t)7 * n_imu[i]);
  }
  q->nKfTotal = nBoot;
  return SOS_OK;
}

extern "C" int sosf_sequence_enable_stereo(sosf_sequence *q, const double *tfmF0ToF1_12, float scaleOptThres) {
  if (!q || !tfmF0ToF1_12) return SOS_ERR_ARG;
  q->stereo = true;
  q->stereoTfm = SE3::from12(tfmF0ToF1_12);
  q->scaleOptThres = scaleOptThres;
  return SOS_OK;
}

extern "C" int sosf_sequence_get_imu(sosf_sequence *q, int frameID, double *state21, double *zero21, double *vel3) {
  if (!q) return SOS_ERR_ARG;
  const auto it = q->vf.find(frameID);
  if (it == q->vf.end()) return SOS_ERR_STATE;
  if (state21) std::memcpy(state21, it->second.state, sizeof(double) * 21);
  if (zero21) std::memcpy(zero21, it->second.zero, sizeof(double) * 21);
  if (vel3) std::memcpy(vel3, it->second.vel, sizeof(double) * 3);
  return SOS_OK;
}

extern "C" int sosf_sequence_set_snapshots(sosf_sequence *q, int on) {
  if (!q) return SOS_ERR_ARG;
  q->snapshots = on != 0;
  return SOS_OK;
}
extern "C" int sosf_sequence_get_snapshot(sosf_sequence *q, int which, int capacity, double *out, int *count) {
  if (!q || which < 0 || which > 3 || capacity < 0 || !count) return SOS_ERR_ARG;
  const std::vector<double> &v = q->snap[which];
  *count = (int)v.size();
  if (out && !v.empty()) std::memcpy(out, v.data(), sizeof(double) * std::min<size_t>(v.size(), (size_t)capacity));
  return SOS_OK;
}
extern "C" int sosf_sequence_get_scale_state(sosf_sequence *q, int32_t *state2) {
  if (!q || !state2) return SOS_ERR_ARG;
  state2[0] = q->scaleState.scaleTrapped;
  state2[1] = q->scaleState.fails;
  return SOS_OK;
}

extern "C" int sosf_sequence_get_imu_calib(sosf_sequence *q, sosf_imu_calib *out) {
  if (!q || !out) return SOS_ERR_ARG;
  *out = q->cal;
  return SOS_OK;
}

extern "C" int sosf_sequence_bootstrap_ex(sosf_sequence *q, int stereoSlot, float *rmse, int *iterations) {
  if (!q) return SOS_ERR_ARG;
  FullSystem *fs = q->fs;
  int its = 0;
  const float r = fs->optimize(q->prm.maxOptIterations, &its);
  if (rmse) *rmse = r;
  if (iterations) *iterations = its;
  if (fs->lastError != SOS_OK) return fs->lastError;
  fs->removeOutliers();
  q->ct->makeK(&fs->HCalib);
  int rc = q->ct->setCoarseTrackingRef(fs->frameHessians);
  if (rc != SOS_OK) return rc;
  if (q->stereo && stereoSlot >= 0 && (rc = optimize_scale(q, stereoSlot, nullptr, nullptr)) != SOS_OK) return rc;
  for (FrameHessian *fh : fs->frameHessians)
    if (make_new_traces(q, fh) < 0) return SOS_ERR_HIP;
  q->haveLastRel = false;
  q->trackHist.clear();
  q->framesSinceKF = 0;
  return SOS_OK;
}
extern "C" int sosf_sequence_bootstrap(sosf_sequence *q, float *rmse, int *iterations) { return sosf_sequence_bootstrap_ex(q, -1, rmse, iterations); }

extern "C" int sosf_sequence_immature_count(sosf_sequence *q, int frameID, int *count) {
  if (!q || !count) return SOS_ERR_ARG;
  auto it = q->imm.find(frameID);
  *count = it == q->imm.end() ? 0 : (int)it->second.size();
  return SOS_OK;
}

extern "C" int sosf_sequence_get_immature(sosf_sequence *q, int frameID, int capacity, sos_immature *out, float *type) {
  if (!q || capacity < 0) return SOS_ERR_ARG;
  auto it = q->imm.find(frameID);
  if (it == q->imm.end()) return SOS_OK;
  const int k = std::min<int>(capacity, (int)it->second.size());
  if (out && k) std::memcpy(out, it->second.data(), sizeof(sos_immature) * k);
  if (type && k) std::memcpy(type, q->immType[frameID].data(), sizeof(float) * k);
  return SOS_OK;
}

// FullSystem::addActiveFrame for a frame whose pyramid is in `slot` (sos_undistort_frame / sosf_upload_image made it)
extern "C" int sosf_add_active_frame(sosf_sequence *q, int slot, int frameID, float ab_exposure, const double *T_init12, sosf_frame_result *out) {
  return sosf_add_active_frame_ex(q, slot, frameID, ab_exposure, T_init12, nullptr, out);
}
// ... with what the IMU / stereo branches need: the frame's timestamp, the IMU samples since the last frame (FullSystem::imu_data is
// extended by them, :625, and handed to the next keyframe, :802-803), the image slot of the stereo partner
extern "C" int sosf_add_active_frame_ex(sosf_sequence *q, int slot, int frameID, float ab_exposure, const double *T_init12, const sosf_frame_extra *extra,
                                        sosf_frame_result *out) {
  if (!q || !out || slot < 0 || slot >= SOS_MAX_SLOTS) return SOS_ERR_ARG;
  if (q->vio && !extra) return SOS_ERR_ARG;
  std::memset(out, 0, sizeof(*out));
  out->newScale = -1;
  FullSystem *fs = q->fs;
  if (fs->frameHessians.empty()) return SOS_ERR_STATE;
  if (q->vio && extra->n_imu > 0 && extra->imu) q->pendingImu.insert(q->pendingImu.end(), extra->imu, extra->imu + (size_t)7 * extra->n_imu);
  FrameHessian *ref = fs->frameHessians.back();
  const SE3 refPose = ref->PRE_camToWorld;
  // ---- initial guess of refToNew: the caller's, else the motion model (FS/FullSystem.cpp:163-200 tries constant motion first; with
  // frames between the keyframes "no motion since the last frame" is inside the tracker's basin and does not feed tracking error back)
  SE3 T;
  if (T_init12) T = SE3::from12(T_init12);
  else if (q->prm.kfEvery == 1 && q->haveLastRel) T = q->lastRel;
  else if (!q->trackHist.empty()) T = q->trackHist.back().inverse() * refPose;
  else if (q->haveLastRel) T = q->lastRel;
  AffLight aff = frame_aff(ref);  // aff_g2l of the tracking reference as the start (lastF->aff_g2l, :203-206)
  const double minRes[5] = {NAN, NAN, NAN, NAN, NAN};
  double lastRes[5] = {0, 0, 0, 0, 0};
  const bool ok = q->ct->trackNewestCoarse(slot, ab_exposure, T, aff, q->ct->levels - 1, minRes, lastRes);
  out->trackingOk = ok ? 1 : 0;
  T.to12(out->refToNew);
  out->aff[0] = aff.a; out->aff[1] = aff.b;
  for (int i = 0; i < 5; i++) out->trackResiduals[i] = lastRes[i];
  for (int i = 0; i < 3; i++) out->flow[i] = q->ct->lastFlowIndicators[i];
  if (!ok) return SOS_OK;  // isLost is the caller's decision (:698-703)
  q->lastRel = T;
  q->haveLastRel = true;
  const SE3 c2w = refPose * T.inverse();  // shell->camToWorld = trackingRef->camToWorld * camToTrackingRef
  c2w.to12(out->camToWorld);
  q->trackHist.push_back(c2w);
  if (q->trackHist.size() > 2) q->trackHist.erase(q->trackHist.begin());
  int rc = trace_new_coarse(q, slot, c2w, aff, ab_exposure);  // :311-361
  if (rc != SOS_OK) return rc;
  // ---- keyframe decision (:709-732), or every kfEvery-th frame
  bool needKF;
  q->framesSinceKF++;
  if (q->prm.kfEvery > 0) needKF = q->framesSinceKF >= q->prm.kfEvery;
  else {
    double refToFh[2];
    AffLight::fromToVecExposure(q->ct->ref_ab_exposure, ab_exposure, q->ct->lastRef_aff_g2l, aff, refToFh);
    const float wh = (float)(fs->prm.w + fs->prm.h);
    needKF = q->prm.kfGlobalWeight * q->prm.maxShiftWeightT * sqrtf((float)out->flow[0]) / wh +
                     q->prm.kfGlobalWeight * q->prm.maxShiftWeightR * sqrtf((float)out->flow[1]) / wh +
                     q->prm.kfGlobalWeight * q->prm.maxShiftWeightRT * sqrtf((float)out->flow[2]) / wh +
                     q->prm.kfGlobalWeight * q->prm.maxAffineWeight * fabsf(logf((float)refToFh[0])) > 1 ||
             2 * q->ct->firstCoarseRMSE < lastRes[0];
  }
  if (!needKF) {  // makeNonKeyFrame (:768-781): traced, nothing else; the frame's pyramid is released by the caller
    out->isKeyframe = 0;
    return SOS_OK;
  }
  q->framesSinceKF = 0;
  out->isKeyframe = 1;
  return make_keyframe(q, slot, frameID, ref->frameID, c2w, aff, ab_exposure, extra, out);
}
