// sos_host.cpp -- implementation of the C++ host facade (see sos_host.hpp) and its flat C entry points
// (include/sos_slam_host.h).  Compiled with -ffp-contract=off: the fp32 host arithmetic that feeds the
// C-ABI (precalc, adHTdeltaF, xAd) follows the same convention as the device kernels.
#include "sos_host.hpp"
#include "sos_pool.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "../../../include/sos_slam_host.h"

#include <chrono>

namespace sos {

// wall-clock phase timers of the GN iteration (sosf_get_timing): 0 accumulate+stitch (device, D2H),
// 1 assemble+solve (host), 2 resubstitute, 3 step+precalc (host), 4 pushState, 5 linearize, 6 applyRes
double g_phase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct PhaseTimer {
  int k;
  double t0;
  explicit PhaseTimer(int k_) : k(k_), t0(now_s()) {}
  ~PhaseTimer() { g_phase[k] += now_s() - t0; }
};

// reference defaults that are not part of sos_params (util/settings.cpp:47-77)
static const float setting_initialRotPrior = 1e11f;
static const float setting_initialTransPrior = 1e10f;
static const float setting_initialAffBPrior = 1e14f;
static const float setting_initialAffAPrior = 1e14f;
static const float setting_thOptIterations = 1.2f;
static const float setting_minIdepthH_marg = 50;

// ------------------------------------------------------------------------------------------------
void AffLight::fromToVecExposure(float exposureF, float exposureT, AffLight g2F, AffLight g2T, double *out) {
  if (exposureF == 0 || exposureT == 0) exposureT = exposureF = 1;
  const double a = std::exp(g2T.a - g2F.a) * exposureT / exposureF;
  const double b = g2T.b - a * g2F.b;
  out[0] = a;
  out[1] = b;
}

CalibHessian::CalibHessian() {
  for (int i = 0; i < 4; i++) value_zero[i] = value_scaled[i] = value[i] = step[i] = value_backup[i] = value_minus_value_zero[i] = 0;
  for (int i = 0; i < 4; i++) value_scaledf[i] = value_scaledi[i] = 0;
}
void CalibHessian::setValue(const double *v) {  // FS/HessianBlocks.h:476-491
  for (int i = 0; i < 4; i++) value[i] = v[i];
  value_scaled[0] = SOS_SCALE_F * v[0];
  value_scaled[1] = SOS_SCALE_F * v[1];
  value_scaled[2] = SOS_SCALE_C * v[2];
  value_scaled[3] = SOS_SCALE_C * v[3];
  for (int i = 0; i < 4; i++) value_scaledf[i] = (float)value_scaled[i];
  value_scaledi[0] = 1.0f / value_scaledf[0];
  value_scaledi[1] = 1.0f / value_scaledf[1];
  value_scaledi[2] = -value_scaledf[2] / value_scaledf[0];
  value_scaledi[3] = -value_scaledf[3] / value_scaledf[1];
  for (int i = 0; i < 4; i++) value_minus_value_zero[i] = value[i] - value_zero[i];
}
void CalibHessian::setValueScaled(const double *vs) {  // FS/HessianBlocks.h:493-506
  for (int i = 0; i < 4; i++) value_scaled[i] = vs[i];
  for (int i = 0; i < 4; i++) value_scaledf[i] = (float)value_scaled[i];
  value[0] = (1.0f / SOS_SCALE_F) * vs[0];
  value[1] = (1.0f / SOS_SCALE_F) * vs[1];
  value[2] = (1.0f / SOS_SCALE_C) * vs[2];
  value[3] = (1.0f / SOS_SCALE_C) * vs[3];
  for (int i = 0; i < 4; i++) value_minus_value_zero[i] = value[i] - value_zero[i];
  value_scaledi[0] = 1.0f / value_scaledf[0];
  value_scaledi[1] = 1.0f / value_scaledf[1];
  value_scaledi[2] = -value_scaledf[2] / value_scaledf[0];
  value_scaledi[3] = -value_scaledf[3] / value_scaledf[1];
}
sos_calib CalibHessian::toCalib() const {
  sos_calib c;
  c.fxl = value_scaledf[0]; c.fyl = value_scaledf[1]; c.cxl = value_scaledf[2]; c.cyl = value_scaledf[3];
  c.fxli = value_scaledi[0]; c.fyli = value_scaledi[1]; c.cxli = value_scaledi[2]; c.cyli = value_scaledi[3];
  return c;
}

// ------------------------------------------------------------------------------------------------
FrameHessian::FrameHessian() {
  for (int i = 0; i < 10; i++) state_zero[i] = state_scaled[i] = state[i] = step[i] = state_backup[i] = 0;
}
FrameHessian::~FrameHessian() {
  for (PointFrameResidual *r : targetedBy) r->idxInTarget = -1;  // (they outlive this frame in the containers of removed points of other frames)
  targetedBy.clear();
  for (PointHessian *p : pointHessians) delete p;
  for (PointHessian *p : pointHessiansMarginalized) delete p;
  for (PointHessian *p : pointHessiansOut) delete p;
}
PointHessian::~PointHessian() {
  for (PointFrameResidual *r : residuals) delete r;
}
void PointFrameResidual::registerTarget() {
  if (!target || idxInTarget >= 0) return;
  idxInTarget = (int)target->targetedBy.size();
  target->targetedBy.push_back(this);
}
PointFrameResidual::~PointFrameResidual() {
  if (target && idxInTarget >= 0) {
    std::vector<PointFrameResidual *> &v = target->targetedBy;
    v[idxInTarget] = v.back();
    v[idxInTarget]->idxInTarget = idxInTarget;
    v.pop_back();
  }
}
static int g_evalCounter = 0;  // unique id of every evalPT ever set: keys the cached FEJ products
void FrameHessian::setState(const double *s) {  // FS/HessianBlocks.h:217-230
  for (int i = 0; i < 10; i++) state[i] = s[i];
  for (int i = 0; i < 3; i++) state_scaled[i] = SOS_SCALE_XI_TRANS * state[i];
  for (int i = 3; i < 6; i++) state_scaled[i] = SOS_SCALE_XI_ROT * state[i];
  state_scaled[6] = SOS_SCALE_A * state[6];
  state_scaled[7] = SOS_SCALE_B * state[7];
  state_scaled[8] = SOS_SCALE_A * state[8];
  state_scaled[9] = SOS_SCALE_B * state[9];
  PRE_camToWorld = SE3::exp(state_scaled) * camToWorld_evalPT;
  PRE_worldToCam = PRE_camToWorld.inverse();
}
void FrameHessian::setStateZero(const double *s) {  // FS/HessianBlocks.cpp:66-101 (nullspaces: dead code downstream)
  for (int i = 0; i < 10; i++) state_zero[i] = s[i];
}
void FrameHessian::setEvalPT(const SE3 &c2w, const double *s) {  // FS/HessianBlocks.h:246-251
  camToWorld_evalPT = c2w;
  worldToCam_evalPT = c2w.inverse();
  evalVersion = ++g_evalCounter;
  setState(s);
  setStateZero(s);
}
void FrameHessian::getPrior(double *p, float modeA, float modeB) const {  // FS/HessianBlocks.h:280-302
  for (int i = 0; i < 10; i++) p[i] = 0;
  if (frameID == 0) {
    p[0] = p[1] = p[2] = setting_initialTransPrior;
    p[3] = p[4] = p[5] = setting_initialRotPrior;
    p[6] = setting_initialAffAPrior;
    p[7] = setting_initialAffBPrior;
  } else {
    p[6] = modeA < 0 ? setting_initialAffAPrior : modeA;
    p[7] = modeB < 0 ? setting_initialAffBPrior : modeB;
  }
  p[8] = setting_initialAffAPrior;
  p[9] = setting_initialAffBPrior;
}

void FrameFramePrecalc::set(const FrameHessian *host, const FrameHessian *target, const CalibHessian *HCalib) {
  if (hostEvalVersion != host->evalVersion || targetEvalVersion != target->evalVersion) {
    const SE3 leftToLeft_0 = target->worldToCam_evalPT * host->camToWorld_evalPT;
    for (int i = 0; i < 9; i++) dev.PRE_RTll_0[i] = (float)leftToLeft_0.R[i];
    for (int i = 0; i < 3; i++) dev.PRE_tTll_0[i] = (float)leftToLeft_0.t[i];
    hostEvalVersion = host->evalVersion; targetEvalVersion = target->evalVersion;
  }
  const SE3 leftToLeft = target->PRE_worldToCam * host->PRE_camToWorld;
  for (int i = 0; i < 9; i++) PRE_RTll[i] = (float)leftToLeft.R[i];
  for (int i = 0; i < 3; i++) PRE_tTll[i] = (float)leftToLeft.t[i];
  distanceLL = (float)std::sqrt(leftToLeft.t[0] * leftToLeft.t[0] + leftToLeft.t[1] * leftToLeft.t[1] + leftToLeft.t[2] * leftToLeft.t[2]);
  // K, K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1] (value_scaledi)
  const float K[9] = {HCalib->value_scaledf[0], 0, HCalib->value_scaledf[2], 0, HCalib->value_scaledf[1], HCalib->value_scaledf[3], 0, 0, 1};
  const float Ki[9] = {HCalib->value_scaledi[0], 0, HCalib->value_scaledi[2], 0, HCalib->value_scaledi[1], HCalib->value_scaledi[3], 0, 0, 1};
  float KR[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) KR[3 * i + j] = K[3 * i] * PRE_RTll[j] + K[3 * i + 1] * PRE_RTll[3 + j] + K[3 * i + 2] * PRE_RTll[6 + j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      dev.PRE_KRKiTll[3 * i + j] = KR[3 * i] * Ki[j] + KR[3 * i + 1] * Ki[3 + j] + KR[3 * i + 2] * Ki[6 + j];
      PRE_RKiTll[3 * i + j] = PRE_RTll[3 * i] * Ki[j] + PRE_RTll[3 * i + 1] * Ki[3 + j] + PRE_RTll[3 * i + 2] * Ki[6 + j];
    }
  for (int i = 0; i < 3; i++) dev.PRE_KtTll[i] = K[3 * i] * PRE_tTll[0] + K[3 * i + 1] * PRE_tTll[1] + K[3 * i + 2] * PRE_tTll[2];
  double aff[2];
  AffLight::fromToVecExposure(host->ab_exposure, target->ab_exposure, host->aff_g2l(), target->aff_g2l(), aff);
  dev.PRE_aff_mode[0] = (float)aff[0];
  dev.PRE_aff_mode[1] = (float)aff[1];
  dev.PRE_b0_mode = (float)host->aff_g2l_0().b;
  dev.pad = 0;
}

// ------------------------------------------------------------------------------------------------
EFPoint::EFPoint(PointHessian *d, EFFrame *h, float idepthFixPrior) : data(d), host(h) {
  priorF = d->hasDepthPrior ? idepthFixPrior * SOS_SCALE_IDEPTH * SOS_SCALE_IDEPTH : 0;  // OB/EnergyFunctionalStructs.cpp:66-72
  deltaF = d->idepth - d->idepth_zero;
}
EFFrame::EFFrame(FrameHessian *d, float modeA, float modeB) : data(d) { takeData(modeA, modeB); }
void EFFrame::takeData(float modeA, float modeB) {  // OB/EnergyFunctionalStructs.cpp:47-64
  double p[10];
  data->getPrior(p, modeA, modeB);
  for (int i = 0; i < 8; i++) {
    prior[i] = p[i];
    delta[i] = data->state[i] - data->state_zero[i];
    delta_prior[i] = data->state[i];  // getPriorZero() == 0
  }
  frameID = data->frameID;
}

EnergyFunctional::EnergyFunctional(sos_ctx *c, const sos_params &p) : ctx(c), prm(p) {
  HM.assign(SOS_CPARS * SOS_CPARS, 0.0);
  bM.assign(SOS_CPARS, 0.0);
  for (int i = 0; i < 4; i++) { cDeltaF[i] = 0; cPrior[i] = 0; }
  sos_ba_create(ctx, &prm, &ba);
}
EnergyFunctional::~EnergyFunctional() {
  for (EFFrame *f : frames) {
    for (EFPoint *p : f->points) {
      for (EFResidual *r : p->residualsAll) { r->data->efResidual = nullptr; delete r; }
      p->data->efPoint = nullptr;
      delete p;
    }
    f->data->efFrame = nullptr;
    delete f;
  }
  if (ba) sos_ba_destroy(ba);
}

// one (host, target) pair of setAdjointsF (OB/EnergyFunctional.cpp:51-84): AH, AT row-major 8 x 8
static void adjoint_pair(const FrameHessian *host, const FrameHessian *target, double *AH, double *AT) {
  const SE3 worldToTarget = target->camToWorld_evalPT.inverse();
  double Ad[36];
  worldToTarget.Adj(Ad);
  for (int i = 0; i < 64; i++) AH[i] = AT[i] = 0;
  for (int i = 0; i < 8; i++) AH[9 * i] = AT[9 * i] = 1;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { AH[8 * i + j] = Ad[6 * j + i]; AT[8 * i + j] = -Ad[6 * j + i]; }
  double aff[2];
  AffLight::fromToVecExposure(host->ab_exposure, target->ab_exposure, host->aff_g2l_0(), target->aff_g2l_0(), aff);
  const float a0 = (float)aff[0];
  AT[8 * 6 + 6] = -a0; AH[8 * 6 + 6] = a0; AT[8 * 7 + 7] = -1; AH[8 * 7 + 7] = a0;
  for (int j = 0; j < 8; j++) {
    for (int i = 0; i < 3; i++) { AH[8 * i + j] *= SOS_SCALE_XI_TRANS; AT[8 * i + j] *= SOS_SCALE_XI_TRANS; }
    for (int i = 3; i < 6; i++) { AH[8 * i + j] *= SOS_SCALE_XI_ROT; AT[8 * i + j] *= SOS_SCALE_XI_ROT; }
    AH[8 * 6 + j] *= SOS_SCALE_A; AT[8 * 6 + j] *= SOS_SCALE_A;
    AH[8 * 7 + j] *= SOS_SCALE_B; AT[8 * 7 + j] *= SOS_SCALE_B;
  }
}
// one pair of setDeltaF (:168-176): adHTdeltaF = delta_host^T adHostF + delta_target^T adTargetF, in float, summed left to right
static void ad_ht_delta_pair(const FrameHessian *host, const FrameHessian *target, const float *AHf, const float *ATf, float *out8) {
  float dh[8], dt[8];
  for (int i = 0; i < 8; i++) {
    dh[i] = (float)(host->state[i] - host->state_zero[i]);
    dt[i] = (float)(target->state[i] - target->state_zero[i]);
  }
  for (int j = 0; j < 8; j++) {
    float s1 = 0, s2 = 0;
    for (int i = 0; i < 8; i++) { s1 += dh[i] * AHf[8 * i + j]; s2 += dt[i] * ATf[8 * i + j]; }
    out8[j] = s1 + s2;
  }
}

void EnergyFunctional::setAdjointsF(CalibHessian *) {  // OB/EnergyFunctional.cpp:42-103
  const int n = nFrames;
  adHost.assign((size_t)n * n * 64, 0.0);
  adTarget.assign((size_t)n * n * 64, 0.0);
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++)
      adjoint_pair(frames[h]->data, frames[t]->data, &adHost[(size_t)(h + t * n) * 64], &adTarget[(size_t)(h + t * n) * 64]);
  for (int i = 0; i < 4; i++) cPrior[i] = prm.initialCalibHessian;
  adHostF.resize(adHost.size());
  adTargetF.resize(adTarget.size());
  for (size_t i = 0; i < adHost.size(); i++) { adHostF[i] = (float)adHost[i]; adTargetF[i] = (float)adTarget[i]; }
  EFAdjointsValid = true;
}

void EnergyFunctional::setDeltaF(CalibHessian *HCalib, bool points) {  // OB/EnergyFunctional.cpp:163-194
  const int n = nFrames;
  adHTdeltaF.assign((size_t)n * n * 8, 0.f);
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) {
      const size_t idx = (size_t)(h + t * n);
      ad_ht_delta_pair(frames[h]->data, frames[t]->data, &adHostF[64 * idx], &adTargetF[64 * idx], &adHTdeltaF[8 * idx]);
    }
  for (int i = 0; i < 4; i++) cDeltaF[i] = (float)HCalib->value_minus_value_zero[i];
  for (EFFrame *f : frames) {
    for (int i = 0; i < 8; i++) {
      f->delta[i] = f->data->state[i] - f->data->state_zero[i];
      f->delta_prior[i] = f->data->state[i];
    }
    if (points)  // skipped while the device owns the point state inside the GN loop (mirrors are refreshed after the step)
      for (EFPoint *p : f->points) p->deltaF = p->data->idepth - p->data->idepth_zero;
  }
  EFDeltaValid = true;
}

EFResidual *EnergyFunctional::insertResidual(PointFrameResidual *r) {  // OB/EnergyFunctional.cpp:644-656
  EFResidual *efr = new EFResidual(r, r->point->efPoint, r->host->efFrame, r->target->efFrame);
  efr->idxInAll = (int)r->point->efPoint->residualsAll.size();
  r->point->efPoint->residualsAll.push_back(efr);
  efr->connKey = (((uint64_t)efr->host->frameID) << 32) + ((uint64_t)efr->target->frameID);
  efr->connEntry = &connectivityMap[efr->connKey];  // (node addresses of a std::map are stable and entries are never erased)
  efr->connEntry->first++;
  nResiduals++;
  r->efResidual = efr;
  packDirty = structDirty = true;
  return efr;
}

EFFrame *EnergyFunctional::insertFrame(FrameHessian *fh, CalibHessian *HCalib) {  // :658-695 (IMU off)
  EFFrame *eff = new EFFrame(fh, prm.affineOptModeA, prm.affineOptModeB);
  eff->idx = (int)frames.size();
  frames.push_back(eff);
  nFrames++;
  fh->efFrame = eff;
  const int ndim = SOS_CPARS + 8 * nFrames, odim = ndim - 8;
  MatXX HMn((size_t)ndim * ndim, 0.0);
  VecX bMn(ndim, 0.0);
  for (int i = 0; i < odim; i++) {
    for (int j = 0; j < odim; j++) HMn[(size_t)i * ndim + j] = HM[(size_t)i * odim + j];
    bMn[i] = bM[i];
  }
  HM.swap(HMn);
  bM.swap(bMn);
  if (imuOwnPrior) {  // step = 29, ndim = CPARS + 1 + 29 nFrames, :666-677
    const int nd = SOSF_IMU_DIM(nFrames), od = nd - 29;
    MatXX Hn((size_t)nd * nd, 0.0);
    VecX bn(nd, 0.0);
    for (int i = 0; i < od; i++) {
      for (int j = 0; j < od; j++) Hn[(size_t)i * nd + j] = HMi[(size_t)i * od + j];
      bn[i] = bMi[i];
    }
    HMi.swap(Hn);
    bMi.swap(bn);
    imuPriorVersion++;
  }
  EFIndicesValid = EFAdjointsValid = EFDeltaValid = false;
  setAdjointsF(HCalib);
  makeIDX();
  for (EFFrame *fh2 : frames) {
    connectivityMap[(((uint64_t)eff->frameID) << 32) + ((uint64_t)fh2->frameID)] = std::make_pair(0, 0);
    if (fh2 != eff) connectivityMap[(((uint64_t)fh2->frameID) << 32) + ((uint64_t)eff->frameID)] = std::make_pair(0, 0);
  }
  packDirty = structDirty = true;
  return eff;
}

EFPoint *EnergyFunctional::insertPoint(PointHessian *ph) {  // :697-708
  EFPoint *efp = new EFPoint(ph, ph->host->efFrame, prm.idepthFixPrior);
  efp->idxInPoints = (int)ph->host->efFrame->points.size();
  ph->host->efFrame->points.push_back(efp);
  nPoints++;
  ph->efPoint = efp;
  EFIndicesValid = false;
  packDirty = structDirty = true;
  return efp;
}

void EnergyFunctional::dropResidual(EFResidual *r) {  // :710-728
  EFPoint *p = r->point;
  p->residualsAll[r->idxInAll] = p->residualsAll.back();
  p->residualsAll[r->idxInAll]->idxInAll = r->idxInAll;
  p->residualsAll.pop_back();
  // (the reference reads r->target->frameID here, also when marginalizeFrame has just deleted that EFFrame: FS/FullSystemMarginalize.cpp:146-176)
  r->connEntry->first--;
  nResiduals--;
  r->data->efResidual = nullptr;
  if (r->data->packIdx >= 0) droppedSincePack.push_back(r->data->packIdx);  // the device snapshot still holds it (sos_ba_kill_residuals)
  r->data->packIdx = -1;
  delete r;
  packDirty = true;
}

void EnergyFunctional::removePoint(EFPoint *p) {  // :954-969
  for (EFResidual *r : std::vector<EFResidual *>(p->residualsAll)) dropResidual(r);
  EFFrame *h = p->host;
  h->points[p->idxInPoints] = h->points.back();
  h->points[p->idxInPoints]->idxInPoints = p->idxInPoints;
  h->points.pop_back();
  nPoints--;
  p->data->efPoint = nullptr;
  EFIndicesValid = false;
  packDirty = true;
  delete p;
}

void EnergyFunctional::makeIDX() {  // :1186-1202
  for (size_t i = 0; i < frames.size(); i++) frames[i]->idx = (int)i;
  allPoints.clear();
  for (EFFrame *f : frames)
    for (EFPoint *p : f->points) allPoints.push_back(p);
  // (r->hostIDX / r->targetIDX of the reference are only read when the window is packed: packWindow takes them from the frames there,
  // instead of every makeIDX -- insertFrame, dropPointsF, marginalizePointsF, marginalizeFrame: four walks over all residuals per keyframe)
  EFIndicesValid = true;
}

VecX EnergyFunctional::getStitchedDeltaF() const {  // :1204-1210
  VecX d(SOS_CPARS + nFrames * 8);
  for (int i = 0; i < 4; i++) d[i] = (double)cDeltaF[i];
  for (int h = 0; h < nFrames; h++)
    for (int i = 0; i < 8; i++) d[SOS_CPARS + 8 * h + i] = frames[h]->delta[i];
  return d;
}

int EnergyFunctional::allreduceF64(double *buf, size_t count) {
  if (commAttached) return sos_ba_allreduce_f64(ba, buf, count);
  if (allreduceF64Hook) { allreduceF64Hook(hookUser, buf, count); return SOS_OK; }
  return allreduceHook ? SOS_ERR_STATE : SOS_OK;
}

int EnergyFunctional::packWindow(std::vector<PointFrameResidual *> *active) {
  // graph edits are shard-local (linearizeAll(true) / removeOutliers drop residuals on some ranks only), and
  // sos_ba_set_window agrees on capacities with a collective once a communicator is attached: every rank repacks
  if (!packDirty && !commAttached) {
    if (active)  // FS/FullSystemOptimize.cpp:316-329 on the unchanged graph
      for (EFPoint *p : allPoints)
        for (PointFrameResidual *r : p->data->residuals)
          if (!r->efResidual->isLinearized) {
            active->push_back(r);
            r->resetOOB();
          }
    return SOS_OK;
  }
  const double tpk0 = now_s();
  makeIDX();
  const int n = nFrames;
  std::vector<int32_t> slots(n);
  for (int i = 0; i < n; i++) slots[i] = frames[i]->data->slot;
  // ONE walk over the graph: the activeResiduals list with resetOOB (when asked for), the point and the residual records.  The
  // record vectors are members: no allocation (and no page faults) per keyframe
  std::vector<sos_point> &pts = packPts;
  std::vector<sos_resid> &res = packRes;
  const size_t P = allPoints.size();
  pts.resize(P);
  // The walk is split into contiguous point ranges over the helper threads (sos_pool.hpp; the reference runs its per-residual loops on
  // IndexThreadReduce workers as well).  Phase A: residual count of every range; a prefix sum gives every range its record slots;
  // phase B: the ranges write their records / allResiduals entries / packIdx in place and collect their part of the active list, which
  // is concatenated in range order -- byte for byte what the serial loop produces, for any thread count.
  const int parts = (int)std::min<size_t>(P ? P : 1, (size_t)4 * (size_t)WalkPool::get().threads());
  std::vector<size_t> partBegin((size_t)parts + 1, 0), partRes((size_t)parts + 1, 0);
  for (int q = 0; q <= parts; q++) partBegin[(size_t)q] = P * (size_t)q / (size_t)parts;
  WalkPool::get().run(parts, [&](int q) {
    size_t c = 0;
    for (size_t k = partBegin[(size_t)q]; k < partBegin[(size_t)q + 1]; k++) c += allPoints[k]->residualsAll.size();
    partRes[(size_t)q + 1] = c;
  });
  for (int q = 0; q < parts; q++) partRes[(size_t)q + 1] += partRes[(size_t)q];
  const size_t Rtot = partRes[(size_t)parts];
  res.resize(Rtot);
  allResiduals.resize(Rtot);
  // (the active list of a range is written straight into the caller's vector at the range's record offset -- an upper bound of its
  // position, a point has as many PointFrameResiduals as EFResiduals -- and the ranges are closed up afterwards)
  const size_t act0 = active ? active->size() : 0;
  if (active) active->resize(act0 + Rtot);
  std::vector<size_t> partAct((size_t)parts, 0);
  PointFrameResidual **actBase = active ? active->data() + act0 : nullptr;
  WalkPool::get().run(parts, [&](int q) {
    size_t w = partRes[(size_t)q];
    PointFrameResidual **act = actBase ? actBase + w : nullptr;
    size_t na = 0;
    const size_t actCap = partRes[(size_t)q + 1] - w;
    for (size_t k = partBegin[(size_t)q]; k < partBegin[(size_t)q + 1]; k++) {
      EFPoint *p = allPoints[k];
      PointHessian *ph = p->data;
      if (active)
        for (PointFrameResidual *r : ph->residuals)
          if (!r->efResidual->isLinearized) {
            if (na < actCap) act[na] = r;
            na++;
            r->resetOOB();
          }
      ph->packIdx = (int)k;
      sos_point &o = pts[k];
      o.u = ph->u; o.v = ph->v;
      o.idepth_scaled = ph->idepth_scaled;
      o.idepth_zero_scaled = ph->idepth_zero_scaled;
      std::memcpy(o.color, ph->color, sizeof(o.color));
      std::memcpy(o.weights, ph->weights, sizeof(o.weights));
      o.priorF = p->priorF;
      o.deltaF = p->deltaF;
      o.host = p->host->idx;
      o.pad = 0;
      for (EFResidual *r : p->residualsAll) {
        sos_resid qq;
        qq.point = (int)k;
        qq.host = r->hostIDX = r->host->idx;
        qq.target = r->targetIDX = r->target->idx;
        qq.flags = (r->isActive() ? SOS_RF_ACTIVE : 0u) | (r->isLinearized ? SOS_RF_LINEARIZED : 0u) | (r->data->isNew ? SOS_RF_ISNEW : 0u);
        qq.state_state = (int)r->data->state_state;
        qq.state_energy = (float)r->data->state_energy;
        r->data->packIdx = (int)w;
        res[w] = qq;
        allResiduals[w] = r;
        w++;
      }
    }
    partAct[(size_t)q] = na;
  });
  if (active) {
    size_t wpos = 0;
    for (int q = 0; q < parts; q++) {
      if (partAct[(size_t)q] > partRes[(size_t)q + 1] - partRes[(size_t)q]) return SOS_ERR_STATE;  // a point with more PointFrameResiduals than EFResiduals: not a graph this facade builds
      if (wpos != partRes[(size_t)q]) std::memmove(actBase + wpos, actBase + partRes[(size_t)q], sizeof(PointFrameResidual *) * partAct[(size_t)q]);
      wpos += partAct[(size_t)q];
    }
    active->resize(act0 + wpos);
  }
  const double tq = now_s();
  int rc = sos_ba_set_window(ba, n, slots.data(), (int)pts.size(), pts.data(), (int)res.size(), res.data(), nullptr, nullptr);
  if (getenv("SOS_TIMING")) fprintf(stderr, "[packWindow] graph walk + records %.0f us, sos_ba_set_window %.0f us\n", (tq - tpk0) * 1e6, (now_s() - tq) * 1e6);
  if (rc == SOS_OK) {
    packDirty = structDirty = false;
    droppedSincePack.clear();
  }
  pointsOnDeviceCurrent = rc == SOS_OK;
  pointStep.assign(pts.size(), 0.f);
  return rc;
}

// The graph only LOST residuals since the last pack (linearizeAll(true), removeOutliers): the snapshot stays, the dropped residuals
// are marked dead on the device.  Returns false when a full pack is needed instead.
bool EnergyFunctional::syncDropsToDevice() {
  if (!packDirty) return true;
  if (structDirty || commAttached) return false;
  if (!droppedSincePack.empty()) {
    if (sos_ba_kill_residuals(ba, droppedSincePack.data(), (int)droppedSincePack.size()) != SOS_OK) return false;
    droppedSincePack.clear();
  }
  return true;  // packDirty stays set: the next optimize() packs the graph as it is then
}

int EnergyFunctional::pushState(CalibHessian *HCalib, bool adjoints, bool points) {
  const int n = nFrames;
  std::vector<sos_precalc> &pc = scrPrecalc;
  pc.resize((size_t)n * n);
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) pc[(size_t)(h + n * t)] = frames[h]->data->targetPrecalc[t].dev;
  const sos_calib c = HCalib->toCalib();
  // points = false: the device's point values are the host's (they were uploaded by the pack, or stepped on both sides by the
  // fused iterations with the same fp32 operation): nothing to send
  if (!points && pointsOnDeviceCurrent)
    return sos_ba_set_state(ba, &c, pc.data(), adHTdeltaF.data(), cDeltaF, adjoints ? adHost.data() : nullptr,
                            adjoints ? adTarget.data() : nullptr, nullptr, nullptr, nullptr);
  std::vector<float> &id = scrId, &idz = scrIdz, &dl = scrDl;
  id.resize(allPoints.size()); idz.resize(allPoints.size()); dl.resize(allPoints.size());
  for (size_t k = 0; k < allPoints.size(); k++) {
    id[k] = allPoints[k]->data->idepth_scaled;
    idz[k] = allPoints[k]->data->idepth_zero_scaled;
    dl[k] = allPoints[k]->deltaF;
  }
  const int rc = sos_ba_set_state(ba, &c, pc.data(), adHTdeltaF.data(), cDeltaF, adjoints ? adHost.data() : nullptr,
                                  adjoints ? adTarget.data() : nullptr, id.data(), idz.data(), dl.data());
  pointsOnDeviceCurrent = rc == SOS_OK;
  return rc;
}

// The visual solve from its pieces (OB/EnergyFunctional.cpp:1069-1148, IMU off): H = H_top with the priors of the L stitch in (upper
// triangle; destroyed), b likewise; the prior around delta (bM + HM delta: a full matrix-vector product), (1 + lambda) on the diagonal,
// H_sc * (1.0f / (1 + lambda)) (a double quotient), Jacobi scaling by (diagonal + 10)^-1/2, LDL^T.  Only the upper triangles (col >=
// row) of H, H_sc and HM enter the matrix: the fused device call delivers just that half, and the LDL^T reads just that half (Eigen's
// LDLT likewise reads one triangle of HFinal_top).  Pure host arithmetic: also reachable through sosf_solve_system for the CPU tests.
// bM + HM delta (OB/EnergyFunctional.cpp:1048): depends on nothing the device delivers -- solveSystemF forms it while the accumulation
// is in flight and hands it in as `priorRhs`
static void prior_rhs(const MatXX &HM, const VecX &bM, const VecX &delta, int dim, VecX &out) {
  out.resize(dim);
  for (int i = 0; i < dim; i++) {
    double s = bM[i];
    for (int j = 0; j < dim; j++) s += HM[(size_t)i * dim + j] * delta[j];
    out[i] = s;
  }
}
static void solve_visual_system(MatXX &H, VecX &b, const MatXX &Hsc, const VecX &bsc, const MatXX &HM, const VecX &bM, const VecX &delta, double lambda,
                                VecX &x, double t_sol0, const VecX *priorRhs = nullptr) {
  const int dim = (int)b.size();
  if (priorRhs) {
    for (int i = 0; i < dim; i++) b[i] += (*priorRhs)[i];
  } else {
    for (int i = 0; i < dim; i++) {
      double s = bM[i];
      for (int j = 0; j < dim; j++) s += HM[(size_t)i * dim + j] * delta[j];
      b[i] += s;
    }
  }
  const double isc = 1.0f / (1 + lambda);
  for (int i = 0; i < dim; i++) b[i] -= bsc[i];
  VecX S(dim);
  for (int i = 0; i < dim; i++) {
    const size_t o = (size_t)i * dim + i;
    const double hii = (H[o] + HM[o]) * (1 + lambda) - Hsc[o] * isc;
    S[i] = 1.0 / std::sqrt(hii + 10);
  }
  for (int i = 0; i < dim; i++) {
    double *hr = &H[(size_t)i * dim];
    const double *mr = &HM[(size_t)i * dim], *sr = &Hsc[(size_t)i * dim];
    const double si = S[i];
    hr[i] = ((hr[i] + mr[i]) * (1 + lambda) - sr[i] * isc) * (si * si);
    for (int j = i + 1; j < dim; j++) hr[j] = ((hr[j] + mr[j]) - sr[j] * isc) * (si * S[j]);
    b[i] *= si;
  }
  const double t1 = now_s();
  g_phase[1] += t1 - t_sol0;
  ldlt_solve(H, b, x, dim, H.data());  // in place: H is this call's scratch (its upper triangle, diagonal included, is what was just formed)
  for (int i = 0; i < dim; i++) x[i] *= S[i];
  g_phase[2] += now_s() - t1;
}

int EnergyFunctional::solveSystemF(int, double lambda, CalibHessian *HCalib, bool deferResubstitute) {  // :1029-1184, IMU off
  lambda = 1e-5;
  const int n = nFrames, dim = SOS_CPARS + 8 * n;
  const size_t dd = (size_t)dim * dim;
  MatXX &HA = scrHA, &Hsc = scrHsc, HL;  // scratch members: no per-iteration allocation / zero fill
  VecX &bA = scrbA, &bsc = scrbsc, bL;
  HA.resize(dd); Hsc.resize(dd); bA.resize(dim); bsc.resize(dim);
  const VecX delta = getStitchedDeltaF();
  // the IMU branch (OB/EnergyFunctional.cpp:1053-1171), first half: everything that does not need the device's H / b -- the IMU factors
  // at the current states, the prior's right-hand side, and with first-estimate Jacobians the forward pass through the kept factor of
  // the IMU states and multipliers (sos_imu.cpp) -- runs while the accumulation is in flight on the device (enqueued by the previous
  // step, by sosf_prepare, or just before this is called)
  double imuPrepT = 0;  // (spent inside the accumulate phase's wall time: taken out of it below)
  auto imuPrepare = [&]() -> int {
    const double t_pre0 = now_s();
    for (int h = 0; h < n; h++) {
      frames[h]->data->PRE_camToWorld.to12(imuFrames[h].camToWorld);
      std::memcpy(imuFrames[h].evalPT_R, frames[h]->data->camToWorld_evalPT.R, sizeof(double) * 9);
    }
    const int rcp = sosf_imu_solve_prepare(imuSettings, imuCalib, n, imuFrames, imuOwnPrior ? HMi.data() : imuHM, imuOwnPrior ? bMi.data() : imuBM,
                                           delta.data(), lambda, imuOwnPrior ? imuPriorVersion : (uint64_t)0);  // a caller-owned prior is compared whole
    imuPrepT = now_s() - t_pre0;
    g_phase[1] += imuPrepT;
    return rcp;
  };
  // SOS_IMU_OVERLAP=1: the first half runs BEHIND the enqueue of the accumulation (sos_ba_gn_accumulate_begin / sos_ba_accumulate_local) so
  // that it overlaps the device also when nothing prefetched the accumulation.  Opt-in until that ordering has passed the GPU suite on
  // an MI355X (it has run under tests/emu only); the default is the order of the last GPU-verified build: first half, then accumulate.
  static const bool imuOverlap = getenv("SOS_IMU_OVERLAP") && atoi(getenv("SOS_IMU_OVERLAP")) != 0;
  if (imuSettings && !imuOverlap) {
    const int rcp = imuPrepare();
    if (rcp != SOS_OK) return rcp;
    imuPrepT = 0;  // outside the accumulate phase's wall time in this order
  }
  const bool havePriorRhs = !imuSettings;
  if (havePriorRhs) {  // (the accumulation was prefetched by the previous step: this runs in its shadow, not between H / b and x)
    const double tpr = now_s();
    prior_rhs(HM, bM, delta, dim, scrPriorRhs);
    g_phase[1] += now_s() - tpr;
  }
  double t_acc0 = now_s();
  if (allreduceHook) {  // shard-local sums -> RCCL all-reduce of the packed fp32 blocks -> identical stitch on every rank
    HL.assign(dd, 0.0);
    bL.assign(dim, 0.0);
    float *dev = nullptr;
    size_t nfl = 0;
    int rc = sos_ba_accumulate_local(ba);
    if (rc == SOS_OK && imuSettings && imuOverlap) rc = imuPrepare();
    if (rc == SOS_OK) rc = sos_ba_acc_buffer(ba, &dev, &nfl);
    if (rc == SOS_OK) rc = sos_ctx_synchronize(ctx);
    if (rc != SOS_OK) return rc;
    allreduceHook(hookUser, dev, nfl);
    rc = sos_ba_stitch(ba, HA.data(), bA.data(), HL.data(), bL.data(), Hsc.data(), bsc.data(), &resInA, &resInL);
    if (rc != SOS_OK) return rc;
    for (size_t i = 0; i < dd; i++) HA[i] += HL[i];
    for (int i = 0; i < dim; i++) bA[i] += bL[i];
  } else {  // HA := HL_top + HA_top already summed by the library
    if (imuSettings && imuOverlap) {
      int rcp = sos_ba_gn_accumulate_begin(ba);  // (a no-op when the previous step prefetched it)
      if (rcp == SOS_OK) rcp = imuPrepare();
      if (rcp != SOS_OK) return rcp;
    }
    const int rc = sos_ba_gn_accumulate(ba, HA.data(), bA.data(), Hsc.data(), bsc.data(), &resInA, &resInL);
    if (rc != SOS_OK) return rc;  // H / b were not delivered: nothing below may consume them
  }
  g_phase[0] += now_s() - t_acc0 - imuPrepT;
  double t_sol0 = now_s();
  MatXX &H = HA;
  VecX &b = bA;
  // priors of the L stitch (usePrior = true), OB/AccumulatedTopHessian.cpp:292-300
  for (int i = 0; i < 4; i++) {
    H[(size_t)i * dim + i] += cPrior[i];
    b[i] += cPrior[i] * (double)cDeltaF[i];
  }
  for (int h = 0; h < n; h++)
    for (int i = 0; i < 8; i++) {
      H[(size_t)(4 + 8 * h + i) * dim + 4 + 8 * h + i] += frames[h]->prior[i];
      b[4 + 8 * h + i] += frames[h]->prior[i] * frames[h]->delta_prior[i];
    }
  if (keepSystem) {  // inspection (sosf_get_last_system): the assembled H_top / b_top (priors in) and H_sc / b_sc, mirrored
    keptH = H; keptHsc = Hsc; keptb = b; keptbsc = bsc;
    for (int i = 0; i < dim; i++)
      for (int j = i + 1; j < dim; j++) {
        keptH[(size_t)j * dim + i] = keptH[(size_t)i * dim + j];
        keptHsc[(size_t)j * dim + i] = keptHsc[(size_t)i * dim + j];
      }
  }
  if (imuSettings) {  // setting_enable_imu && HCalib->imu_initialized: the IMU branch, OB/EnergyFunctional.cpp:1053-1171
    if (sosf_imu_solve_prepared_form() != 1)  // the fused device call delivers the upper triangle only; the literal form reads H whole
      for (int i = 0; i < dim; i++)
        for (int j = i + 1; j < dim; j++) {
          H[(size_t)j * dim + i] = H[(size_t)i * dim + j];
          Hsc[(size_t)j * dim + i] = Hsc[(size_t)i * dim + j];
        }
    VecX x(dim);
    imuStep.assign((size_t)21 * n, 0.0);
    const int rcf = sosf_imu_solve_finish(H.data(), b.data(), Hsc.data(), bsc.data(), x.data(), &imuScaleStep, imuStep.data());
    if (rcf != SOS_OK) return rcf;
    lastX = x;
    for (int i = 0; i < 4; i++) HCalib->step[i] = -x[i];
    for (EFFrame *h : frames) {
      for (int i = 0; i < 8; i++) h->data->step[i] = -x[SOS_CPARS + 8 * h->idx + i];
      h->data->step[8] = h->data->step[9] = 0;
    }
    // IMU and scale update of doStepFromBackup with unit step factors (FS/FullSystemOptimize.cpp:218-230)
    imuCalib->scale += imuScaleStep;
    for (int h = 0; h < n; h++)
      for (int k = 0; k < 21; k++) imuFrames[h].state_imu[k] += imuStep[(size_t)21 * h + k];
    g_phase[2] += now_s() - t_sol0;
    if (deferResubstitute) return SOS_OK;
    pointStep.resize(allPoints.size());
    const int rcr = sos_ba_resubstitute(ba, x.data(), pointStep.data());
    if (rcr != SOS_OK) return rcr;
    for (size_t k = 0; k < allPoints.size(); k++) allPoints[k]->data->step = pointStep[k];
    return SOS_OK;
  }
  VecX x;
  solve_visual_system(H, b, Hsc, bsc, HM, bM, delta, lambda, x, t_sol0, havePriorRhs ? &scrPriorRhs : nullptr);
  lastX = x;
  // resubstituteF_MT, :496-524
  for (int i = 0; i < 4; i++) HCalib->step[i] = -x[i];
  for (EFFrame *h : frames) {
    for (int i = 0; i < 8; i++) h->data->step[i] = -x[SOS_CPARS + 8 * h->idx + i];
    h->data->step[8] = h->data->step[9] = 0;
  }
  if (deferResubstitute) return SOS_OK;  // done by sos_ba_gn_step together with the next linearisation
  pointStep.resize(allPoints.size());
  const int rcr = sos_ba_resubstitute(ba, x.data(), pointStep.data());
  if (rcr != SOS_OK) return rcr;
  for (size_t k = 0; k < allPoints.size(); k++) allPoints[k]->data->step = pointStep[k];
  return SOS_OK;
}

double EnergyFunctional::calcMEnergyF() {  // :553-561
  const VecX delta = getStitchedDeltaF();
  const int dim = (int)delta.size();
  double e = 0;
  for (int i = 0; i < dim; i++) {
    double s = 2 * bM[i];
    for (int j = 0; j < dim; j++) s += HM[(size_t)i * dim + j] * delta[j];
    e += delta[i] * s;
  }
  return e;
}

double EnergyFunctional::calcLEnergyF_MT() {  // :626-642
  double E = 0;
  for (EFFrame *f : frames)
    for (int i = 0; i < 8; i++) E += f->delta_prior[i] * f->prior[i] * f->delta_prior[i];
  float e4 = 0;
  for (int i = 0; i < 4; i++) e4 += cDeltaF[i] * (float)cPrior[i] * cDeltaF[i];
  E += e4;
  double Ed = 0;
  sos_ba_calc_lenergy(ba, &Ed);
  return E + Ed;
}

int EnergyFunctional::marginalizePointsF() {  // :891-936, IMU off
  allPointsToMarg.clear();
  std::vector<int32_t> idx;
  for (EFFrame *f : frames)
    for (EFPoint *p : f->points)
      if (p->stateFlag == PS_MARGINALIZE) {
        p->priorF *= prm.idepthFixPriorMargFac;
        for (EFResidual *r : p->residualsAll)
          if (r->isActive()) connectivityMap[(((uint64_t)r->host->frameID) << 32) + ((uint64_t)r->target->frameID)].second++;
        allPointsToMarg.push_back(p);
        idx.push_back(p->data->packIdx);
      }
  const int dim = SOS_CPARS + 8 * nFrames;
  const size_t dd = (size_t)dim * dim;
  MatXX M(dd, 0.0), Msc(dd, 0.0);
  VecX Mb(dim, 0.0), Mbsc(dim, 0.0);
  int rin = 0;
  if (!idx.empty()) {
    std::vector<float> pr(idx.size());  // priorF changed above: refresh the device copy
    for (size_t k = 0; k < idx.size(); k++) pr[k] = allPointsToMarg[k]->priorF;
    int rcm = sos_ba_update_point_priors(ba, idx.data(), pr.data(), (int)idx.size());
    if (rcm == SOS_OK) rcm = sos_ba_accumulate_marg(ba, idx.data(), (int)idx.size(), M.data(), Mb.data(), Msc.data(), Mbsc.data(), &rin);
    if (rcm != SOS_OK) {  // nothing consumed: the points stay in the window, the prior is untouched
      for (EFPoint *p : allPointsToMarg) p->priorF /= prm.idepthFixPriorMargFac;
      return rcm;
    }
  }
  for (EFPoint *p : allPointsToMarg) removePoint(p);
  int rcx = SOS_OK;
  {  // multi-GPU: every rank marginalised its own shard; the prior update (and resInM) is the sum over ranks
    std::vector<double> upd(dd + dim + 1);
    for (size_t i = 0; i < dd; i++) upd[i] = M[i] - Msc[i];
    for (int i = 0; i < dim; i++) upd[dd + i] = Mb[i] - Mbsc[i];
    upd[dd + dim] = rin;
    rcx = allreduceF64(upd.data(), upd.size());
    if (rcx == SOS_OK) {  // on failure the prior is left untouched instead of diverging between ranks
      for (size_t i = 0; i < dd; i++) HM[i] += prm.margWeightFac * upd[i];
      for (int i = 0; i < dim; i++) bM[i] += prm.margWeightFac * upd[dd + i];
      resInM += (int)upd[dd + dim];
      if (imuOwnPrior) {  // expandHbtoFitImu(H, b); HM += setting_margWeightFac * H, :928-932
        // (scattered through the index map of the expansion instead of through an expanded copy: the entries outside its image are
        // zero there, 2 x 1 MB for 10^4 numbers at twelve keyframes)
        const int nd = SOSF_IMU_DIM(nFrames);
        auto gidx = [](int a) { return a < SOS_CPARS ? a : SOS_CPARS + 1 + 29 * ((a - SOS_CPARS) / 8) + (a - SOS_CPARS) % 8; };
        for (int r = 0; r < dim; r++) {
          double *dst = &HMi[(size_t)gidx(r) * nd];
          const double *src = &upd[(size_t)r * dim];
          for (int c = 0; c < dim; c++) dst[gidx(c)] += prm.margWeightFac * src[c];
          bMi[gidx(r)] += prm.margWeightFac * upd[dd + r];
        }
        imuPriorVersion++;
      }
    }
  }
  EFIndicesValid = false;
  makeIDX();
  return rcx;
}

void EnergyFunctional::dropPointsF() {  // :938-952
  for (EFFrame *f : frames)
    for (int i = 0; i < (int)f->points.size(); i++) {
      EFPoint *p = f->points[i];
      if (p->stateFlag == PS_DROP) {
        removePoint(p);
        i--;
      }
    }
  EFIndicesValid = false;
  makeIDX();
}

void EnergyFunctional::imuAdoptPrior() {
  const int nd = SOSF_IMU_DIM(nFrames);
  HMi.assign((size_t)nd * nd, 0.0);
  bMi.assign(nd, 0.0);
  if (nFrames > 0) sosf_imu_expand(nFrames, HM.data(), bM.data(), HMi.data(), bMi.data());
  imuPriorVersion++;
  imuOwnPrior = true;
}

// The prior algebra of EnergyFunctional::marginalizeFrame in its visual form (OB/EnergyFunctional.cpp:788-858): the keyframe's block
// moved to the end, its pose prior added there, Jacobi scaling, Schur complement on the scaled system, unscaling, symmetrisation.
// Pure host arithmetic on (HM, bM): also reachable through sosf_marginalize_frame_prior for the CPU tests.  false: the block is
// singular (the reference asserts isfinite(hpi) at :844); nothing is written then.
static bool marginalize_frame_prior(const MatXX &HM, const VecX &bM, int nFrames, int idx, const double *prior8, const double *dprior8, MatXX &HMout,
                                    VecX &bMout) {
  const int step = 8, odim = SOS_CPARS + nFrames * step, ndim = odim - step;
  const int io = SOS_CPARS + idx * step;
  // move the frame's block to the end (row/column permutation)
  std::vector<int> perm;
  for (int i = 0; i < odim; i++)
    if (i < io || i >= io + step) perm.push_back(i);
  for (int i = 0; i < step; i++) perm.push_back(io + i);
  MatXX Hp((size_t)odim * odim);
  VecX bp(odim);
  for (int i = 0; i < odim; i++) {
    bp[i] = bM[perm[i]];
    for (int j = 0; j < odim; j++) Hp[(size_t)i * odim + j] = HM[(size_t)perm[i] * odim + perm[j]];
  }
  // add the prior here (instead of to the active part)
  for (int i = 0; i < 8; i++) {
    Hp[(size_t)(ndim + i) * odim + ndim + i] += prior8[i];
    bp[ndim + i] += prior8[i] * dprior8[i];
  }
  VecX SVec(odim), SVecI(odim);
  for (int i = 0; i < odim; i++) {
    SVec[i] = std::sqrt(std::fabs(Hp[(size_t)i * odim + i]) + 10);
    SVecI[i] = 1.0 / SVec[i];
  }
  for (int i = 0; i < odim; i++) {
    for (int j = 0; j < odim; j++) Hp[(size_t)i * odim + j] *= SVecI[i] * SVecI[j];
    bp[i] *= SVecI[i];
  }
  std::vector<double> hpi((size_t)step * step), hpinv;
  for (int i = 0; i < step; i++)
    for (int j = 0; j < step; j++) hpi[(size_t)i * step + j] = Hp[(size_t)(ndim + i) * odim + ndim + j];
  if (!mat_inverse(hpi, hpinv, step)) return false;
  // bli = bottomLeft^T * hpi ; top -= bli * bottomLeft
  MatXX bli((size_t)ndim * step);
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < step; j++) {
      double s = 0;
      for (int k = 0; k < step; k++) s += Hp[(size_t)(ndim + k) * odim + i] * hpinv[(size_t)k * step + j];
      bli[(size_t)i * step + j] = s;
    }
  for (int i = 0; i < ndim; i++) {
    for (int j = 0; j < ndim; j++) {
      double s = 0;
      for (int k = 0; k < step; k++) s += bli[(size_t)i * step + k] * Hp[(size_t)(ndim + k) * odim + j];
      Hp[(size_t)i * odim + j] -= s;
    }
    double s = 0;
    for (int k = 0; k < step; k++) s += bli[(size_t)i * step + k] * bp[ndim + k];
    bp[i] -= s;
  }
  MatXX HMn((size_t)ndim * ndim);
  bMout.assign(ndim, 0.0);
  for (int i = 0; i < ndim; i++) {
    for (int j = 0; j < ndim; j++) HMn[(size_t)i * ndim + j] = Hp[(size_t)i * odim + j] * SVec[i] * SVec[j];
    bMout[i] = bp[i] * SVec[i];
  }
  HMout.assign((size_t)ndim * ndim, 0.0);
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) HMout[(size_t)i * ndim + j] = 0.5 * (HMn[(size_t)i * ndim + j] + HMn[(size_t)j * ndim + i]);
  return true;
}

int EnergyFunctional::marginalizeFrame(EFFrame *fh) {  // :730-889; the IMU form first when the expanded prior lives here
  // both reductions are formed into temporaries and committed together at the end: a failure of either leaves HM / bM, HMi / bMi,
  // the frame list and the caller's IMU records describing the same window
  MatXX Ho;
  VecX bo;
  if (imuOwnPrior) {
    if (!imuSettings || !imuCalib || !imuFrames) return SOS_ERR_STATE;
    for (int h = 0; h < nFrames; h++) {  // the records carry the poses the factors are linearised at
      frames[h]->data->PRE_camToWorld.to12(imuFrames[h].camToWorld);
      std::memcpy(imuFrames[h].evalPT_R, frames[h]->data->camToWorld_evalPT.R, sizeof(double) * 9);
    }
    const VecX delta = getStitchedDeltaF();
    const int nd = SOSF_IMU_DIM(nFrames - 1);
    Ho.resize((size_t)nd * nd);
    bo.resize(nd);
    const int rci = sosf_imu_marginalize_frame(imuSettings, imuCalib, nFrames, imuFrames, fh->idx, delta.data(), fh->prior, fh->delta_prior,
                                               prm.margWeightFac, HMi.data(), bMi.data(), Ho.data(), bo.data());
    if (rci != SOS_OK) return rci;
  }
  MatXX HMn;
  VecX bMn;
  if (!marginalize_frame_prior(HM, bM, nFrames, fh->idx, fh->prior, fh->delta_prior, HMn, bMn)) return SOS_ERR_STATE;  // HM / bM untouched
  HM.swap(HMn);
  bM.swap(bMn);
  if (imuOwnPrior) {
    HMi.swap(Ho);
    bMi.swap(bo);
    imuPriorVersion++;
    // the leaving keyframe's IMU samples go in front of its successor's (FS/FullSystemMarginalize.cpp:226-228): the factor of
    // keyframe idx + 1 then covers the whole interval from idx - 1.  The merged list lives here until the next sosf_set_imu; a
    // caller that hands in fresh records afterwards has to hand in the merged samples (include/sos_slam_host.h)
    if (fh->idx + 1 < nFrames) {
      const sosf_imu_frame &cur = imuFrames[fh->idx];
      sosf_imu_frame &nx = imuFrames[fh->idx + 1];
      imuMergedSamples.emplace_back();
      std::vector<double> &m = imuMergedSamples.back();
      m.reserve((size_t)7 * (cur.n_imu + nx.n_imu));
      if (cur.n_imu > 0) m.insert(m.end(), cur.imu, cur.imu + (size_t)7 * cur.n_imu);
      if (nx.n_imu > 0) m.insert(m.end(), nx.imu, nx.imu + (size_t)7 * nx.n_imu);
      nx.n_imu = cur.n_imu + nx.n_imu;
      nx.imu = m.empty() ? nullptr : m.data();
    }
    for (int h = fh->idx; h + 1 < nFrames; h++) imuFrames[h] = imuFrames[h + 1];  // the records stay aligned with the window
    for (auto it = imuMergedSamples.begin(); it != imuMergedSamples.end();) {  // lists no record points at any more
      bool used = false;
      for (int h = 0; h + 1 < nFrames && !used; h++) used = imuFrames[h].imu == it->data();
      it = used ? std::next(it) : imuMergedSamples.erase(it);
    }
  }
  for (unsigned i = fh->idx; i + 1 < frames.size(); i++) {
    frames[i] = frames[i + 1];
    frames[i]->idx = (int)i;
  }
  frames.pop_back();
  nFrames--;
  fh->data->efFrame = nullptr;
  EFIndicesValid = EFAdjointsValid = EFDeltaValid = false;
  makeIDX();
  packDirty = structDirty = true;
  delete fh;
  return SOS_OK;
}

// ================================================================================================
// FullSystem
// ================================================================================================
FullSystem::FullSystem(const sos_params &p, int device, void *stream) : prm(p) {
  std::memset(slotUsed, 0, sizeof(slotUsed));
  lastError = sos_ctx_create(device, stream, p.w, p.h, &ctx);
  if (lastError != SOS_OK) {
    ctx = nullptr;
    return;
  }
  ef = new EnergyFunctional(ctx, prm);
  if (!ef->ba) {
    delete ef;
    ef = nullptr;
  }
}
FullSystem::~FullSystem() {
  if (ef && residentActive) residentFlush();
  delete ef;
  for (FrameHessian *f : frameHessians) delete f;
  if (ctx) sos_ctx_destroy(ctx);
}

FrameHessian *FullSystem::addFrame(const double *c2w, const double *state10, float ab_exposure, int frameID,
                                   float frameEnergyTH, const float *image, int haveSlot) {
  int slot = -1;
  if (image) {
    for (int s = 0; s < SOS_MAX_SLOTS; s++)
      if (!slotUsed[s]) { slot = s; break; }
    if (slot < 0) return nullptr;
    if (sos_make_pyramid(ctx, slot, image, nullptr) != SOS_OK) return nullptr;  // makeImages, FS/FullSystem.cpp:650
  } else {  // the frame was tracked first: its pyramid already sits in a slot (sosf_upload_image / sos_undistort_frame)
    if (haveSlot < 0 || haveSlot >= SOS_MAX_SLOTS) return nullptr;
    slot = haveSlot;
  }
  slotUsed[slot] = true;
  FrameHessian *fh = new FrameHessian();
  fh->slot = slot;
  fh->ab_exposure = ab_exposure;
  fh->frameID = frameID;
  fh->frameEnergyTH = frameEnergyTH;
  fh->camToWorld_evalPT = SE3::from12(c2w);
  fh->worldToCam_evalPT = fh->camToWorld_evalPT.inverse();
  fh->evalVersion = ++g_evalCounter;
  double sz[10];
  for (int i = 0; i < 10; i++) sz[i] = i < 6 ? 0.0 : state10[i];
  fh->setStateZero(sz);
  fh->setState(state10);
  fh->idx = (int)frameHessians.size();
  frameHessians.push_back(fh);
  ef->insertFrame(fh, &HCalib);  // FS/FullSystem.cpp:814
  setPrecalcValues();            // :816
  return fh;
}

PointHessian *FullSystem::addPoint(const sos_point &p) {
  flushPointMirrors();  // the flat API's lazily stepped points: the objects are current before anything reads, edits or reorders them
  if (p.host < 0 || p.host >= (int)frameHessians.size()) return nullptr;
  PointHessian *ph = new PointHessian();
  ph->host = frameHessians[p.host];
  ph->u = p.u; ph->v = p.v;
  std::memcpy(ph->color, p.color, sizeof(ph->color));
  std::memcpy(ph->weights, p.weights, sizeof(ph->weights));
  ph->setIdepth(p.idepth_scaled * (1.0f / SOS_SCALE_IDEPTH));
  ph->setIdepthZero(p.idepth_zero_scaled * (1.0f / SOS_SCALE_IDEPTH));
  ph->hasDepthPrior = p.priorF > 0;
  ph->lastResiduals[0] = std::make_pair((PointFrameResidual *)nullptr, OOB);
  ph->lastResiduals[1] = std::make_pair((PointFrameResidual *)nullptr, OOB);
  ph->host->pointHessians.push_back(ph);
  ph->userIdx = (int)userPoints.size();
  userPoints.push_back(ph);
  ef->insertPoint(ph);
  return ph;
}

PointFrameResidual *FullSystem::addResidual(PointHessian *ph, FrameHessian *target, const sos_resid &q) {
  PointFrameResidual *r = new PointFrameResidual();
  r->point = ph;
  r->host = ph->host;
  r->target = target;
  r->registerTarget();
  r->state_state = (ResState)q.state_state;
  r->state_energy = q.state_energy;
  r->isNew = (q.flags & SOS_RF_ISNEW) != 0;
  ph->residuals.push_back(r);
  EFResidual *e = ef->insertResidual(r);
  e->isActiveAndIsGoodNEW = (q.flags & SOS_RF_ACTIVE) != 0;
  // lastResiduals: [0] = residual to the newest frame, [1] = the one before (FS/FullSystem.cpp:826-828)
  ph->lastResiduals[1] = ph->lastResiduals[0];
  ph->lastResiduals[0] = std::make_pair(r, IN);
  return r;
}

void FullSystem::ensureHostPrecalc() {  // after iterations through the flat API with the device-side step (sosf_gn_iteration)
  if (hostPrecalcStale) setPrecalcValues(false);
}

void FullSystem::setPrecalcValues(bool points) {  // FS/FullSystem.cpp:1099-1107
  hostPrecalcStale = false;
  for (FrameHessian *fh : frameHessians) {
    fh->targetPrecalc.resize(frameHessians.size());
    for (size_t i = 0; i < frameHessians.size(); i++) fh->targetPrecalc[i].set(fh, frameHessians[i], &HCalib);
  }
  ef->setDeltaF(&HCalib, points);
}

void FullSystem::setNewFrameEnergyTH() {  // FS/FullSystemOptimize.cpp:84-124
  std::vector<float> allResVec;
  allResVec.reserve(activeResiduals.size());
  FrameHessian *newFrame = frameHessians.back();
  for (PointFrameResidual *r : activeResiduals) {
    const float e = h_newEnergyWO[r->packIdx];
    if (e >= 0 && r->target == newFrame) allResVec.push_back(e);
  }
  if (comm) {  // multi-GPU: the statistic is taken over the residuals of all ranks
    int cap = 0, tot = 0;
    sos_ba_newest_capacity(ef->ba, &cap);
    std::vector<float> all((size_t)cap + 1);
    if (sos_ba_gather_energies(ef->ba, allResVec.data(), (int)allResVec.size(), all.data(), &tot) == SOS_OK) {
      all.resize(tot);
      allResVec.swap(all);
    }
  }
  setNewFrameEnergyTH(allResVec);
}

// the threshold from the order statistic (FS/FullSystemOptimize.cpp:104-124): nthValue = the element at index (int)(frameEnergyTHN * count)
// -- a FLOAT product, as the reference forms it -- of the newest frame's energies
static float energy_threshold_from_nth(float nthValue, float facMedian, float constWeight, float overall) {
  const float nthElement = sqrtf(nthValue);
  float th = nthElement * facMedian;
  th = 26.0f * constWeight + th * (1 - constWeight);
  th = th * th;
  th *= overall * overall;
  return th;
}
static float energy_threshold(std::vector<float> &allResVec, float thn, float facMedian, float constWeight, float overall) {
  if (allResVec.empty()) return 12 * 12 * SOS_PATTERN_NUM;
  // (the reference asserts nthIdx < size, FS/FullSystemOptimize.cpp:110; a frameEnergyTHN close to 1 rounds the float product up to size)
  const int nthIdx = std::min((int)(thn * allResVec.size()), (int)allResVec.size() - 1);
  std::nth_element(allResVec.begin(), allResVec.begin() + nthIdx, allResVec.end());
  return energy_threshold_from_nth(allResVec[nthIdx], facMedian, constWeight, overall);
}

void FullSystem::setNewFrameEnergyTH(std::vector<float> &allResVec) {
  FrameHessian *newFrame = frameHessians.back();
  if (ef->nthHook) {  // global order statistic over all shards
    const float nthValue = ef->nthHook(ef->hookUser, allResVec.data(), (int)allResVec.size(), prm.frameEnergyTHN);
    newFrame->frameEnergyTH = nthValue < 0 ? 12 * 12 * SOS_PATTERN_NUM
                                           : energy_threshold_from_nth(nthValue, prm.frameEnergyTHFacMedian, prm.frameEnergyTHConstWeight, prm.overallEnergyTHWeight);
    return;
  }
  newFrame->frameEnergyTH = energy_threshold(allResVec, prm.frameEnergyTHN, prm.frameEnergyTHFacMedian, prm.frameEnergyTHConstWeight, prm.overallEnergyTHWeight);
}

double FullSystem::linearizeAll(bool fix) {  // FS/FullSystemOptimize.cpp:125-182
  const int n = (int)frameHessians.size();
  std::vector<float> th(n);
  for (int i = 0; i < n; i++) th[i] = frameHessians[i]->frameEnergyTH;
  const size_t R = ef->allResiduals.size();
  h_newEnergyWO.resize(R);
  double E = 0;
  if (!fix) {
    lastError = sos_ba_linearize(ef->ba, th.data(), &E, nullptr, nullptr, h_newEnergyWO.data(), nullptr);
    setNewFrameEnergyTH();
    return E;
  }
  const bool tmg = getenv("SOS_TIMING") != nullptr;
  const double tl0 = now_s();
  const sos_resid_final *rec = nullptr;
  const float *pmax = nullptr;
  const int32_t *pcnt = nullptr;
  {
    int cap = 0, cnt = 0;
    sos_ba_newest_capacity(ef->ba, &cap);
    newestE.resize((size_t)cap + 1);
    lastError = sos_ba_linearize_final(ef->ba, th.data(), 0, &E, &rec, &pmax, &pcnt, newestE.data(), &cnt);
    newestE.resize(lastError == SOS_OK ? cnt : 0);
  }
  if (lastError != SOS_OK) return NAN;
  const double tl1 = now_s();
  // ONE walk over the active residuals: r->applyRes(true) inside the reductor (:51), the removal list (:72-73) and the
  // lastResiduals states (:150-156, which only read what this walk has just written for the same residual)
  // (ranges of the active list over the helper threads, sos_pool.hpp: every residual is written by the one range that holds it, a
  // point's two lastResiduals slots are distinct objects, the removal lists are concatenated in range order = the serial loop's list)
  std::vector<PointFrameResidual *> toRemove;
  {
    const size_t nA = activeResiduals.size();
    const int parts = (int)std::min<size_t>(nA ? nA : 1, (size_t)4 * (size_t)WalkPool::get().threads());
    std::vector<std::vector<PointFrameResidual *>> partRemove((size_t)parts);
    WalkPool::get().run(parts, [&](int pq) {
      std::vector<PointFrameResidual *> &rem = partRemove[(size_t)pq];
      for (size_t i = nA * (size_t)pq / (size_t)parts, e = nA * ((size_t)pq + 1) / (size_t)parts; i < e; i++) {
        PointFrameResidual *r = activeResiduals[i];
        const sos_resid_final &q = rec[r->packIdx];
        r->state_NewState = (ResState)q.state_NewState;
        r->state_NewEnergy = q.state_NewEnergy;
        r->state_NewEnergyWithOutlier = q.state_NewEnergyWithOutlier;
        r->centerProjectedTo[0] = q.centerProjectedTo[0];
        r->centerProjectedTo[1] = q.centerProjectedTo[1];
        r->centerProjectedTo[2] = q.centerProjectedTo[2];
        r->state_state = (ResState)q.state_state;
        r->state_energy = q.state_energy;
        r->efResidual->isActiveAndIsGoodNEW = q.active != 0;
        if (!q.active) rem.push_back(r);
        PointHessian *ph = r->point;
        if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].second = r->state_state;
        else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].second = r->state_state;
      }
    });
    for (int pq = 0; pq < parts; pq++) toRemove.insert(toRemove.end(), partRemove[(size_t)pq].begin(), partRemove[(size_t)pq].end());
    // isNew bookkeeping (:55-71), formed per point on the device over its active residuals
    const size_t nP = ef->allPoints.size();
    const int pparts = (int)std::min<size_t>(nP ? nP : 1, (size_t)2 * (size_t)WalkPool::get().threads());
    WalkPool::get().run(pparts, [&](int pq) {
      for (size_t k = nP * (size_t)pq / (size_t)pparts, e = nP * ((size_t)pq + 1) / (size_t)pparts; k < e; k++) {
        PointHessian *p = ef->allPoints[k]->data;
        if (pcnt[k] > 0) {
          if (pmax[k] > p->maxRelBaseline) p->maxRelBaseline = pmax[k];
          p->numGoodResiduals += pcnt[k];
        }
      }
    });
  }
  const double tl2 = now_s();
  setNewFrameEnergyTH(newestE);
  const double tl3 = now_s();
  for (PointFrameResidual *r : toRemove) {  // :158-176
    PointHessian *ph = r->point;
    if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].first = nullptr;
    else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].first = nullptr;
    for (size_t k = 0; k < ph->residuals.size(); k++)
      if (ph->residuals[k] == r) {
        ef->dropResidual(r->efResidual);
        ph->residuals[k] = ph->residuals.back();
        ph->residuals.pop_back();
        delete r;
        break;
      }
  }
  if (tmg) fprintf(stderr, "[linearizeAll(true)] device call + readback %.0f us, residual / point walk %.0f us, threshold %.0f us, removal (%zu) %.0f us\n",
                   (tl1 - tl0) * 1e6, (tl2 - tl1) * 1e6, (tl3 - tl2) * 1e6, toRemove.size(), (now_s() - tl3) * 1e6);
  return E;
}

void FullSystem::applyRes() { sos_ba_apply_res(ef->ba); }

void FullSystem::backupState() {  // :260-269
  std::memcpy(HCalib.value_backup, HCalib.value, sizeof(HCalib.value));
  backupSumNID = 0;
  backupNumID = 0;
  if (pointMirrorsStale) {  // the loop's flat copies (snapshot order = frames -> points, the order of the walk below)
    for (FrameHessian *fh : frameHessians) std::memcpy(fh->state_backup, fh->state, sizeof(fh->state));
    const size_t P = flatIdepth.size();
    flatBackup.resize(P);
    if (P) std::memcpy(flatBackup.data(), flatIdepth.data(), sizeof(float) * P);
    float s = 0;
    for (size_t j = 0; j < P; j++) s += fabsf(flatBackup[(size_t)flatOrder[j]]);  // the walk's order: frames -> pointHessians
    backupSumNID = s;
    backupNumID = (float)P;
  } else {
    for (FrameHessian *fh : frameHessians) {
      std::memcpy(fh->state_backup, fh->state, sizeof(fh->state));
      for (PointHessian *ph : fh->pointHessians) {
        ph->idepth_backup = ph->idepth;
        backupSumNID += fabsf(ph->idepth_backup);  // same order as the loop of doStepFromBackup, FS/FullSystemOptimize.cpp:207-213
        backupNumID++;
      }
    }
  }
  // multi-GPU: the points are sharded, the termination test of doStepFromBackup (sqrtf(sumT) * sumNID) must come out the
  // same on every rank or the ranks leave the loop -- and the collectives of the fused calls -- at different iterations.
  // Benchmark loops (pipelineAlways) ignore canbreak and skip the exchange.
  if (ef->multiRank() && !pipelineAlways) {
    double v[2] = {backupSumNID, backupNumID};
    if (ef->allreduceF64(v, 2) == SOS_OK) {
      backupSumNID = (float)v[0];
      backupNumID = (float)v[1];
    } else {
      lastError = SOS_ERR_STATE;
    }
  }
}

bool FullSystem::doStepFromBackup(float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD, bool pointsOnDevice,
                                  bool precalcOnDevice) {
  double pstepfac[10];
  for (int i = 0; i < 3; i++) pstepfac[i] = stepfacT;
  for (int i = 3; i < 6; i++) pstepfac[i] = stepfacR;
  for (int i = 6; i < 10; i++) pstepfac[i] = stepfacA;
  float sumA = 0, sumB = 0, sumT = 0, sumR = 0, sumID = 0, numID = 0, sumNID = 0;
  double v[4];
  for (int i = 0; i < 4; i++) v[i] = HCalib.value_backup[i] + stepfacC * HCalib.step[i];
  HCalib.setValue(v);
  for (FrameHessian *fh : frameHessians) {
    double s[10];
    for (int i = 0; i < 10; i++) s[i] = fh->state_backup[i] + pstepfac[i] * fh->step[i];
    fh->setState(s);
    sumA += fh->step[6] * fh->step[6];
    sumB += fh->step[7] * fh->step[7];
    sumT += fh->step[0] * fh->step[0] + fh->step[1] * fh->step[1] + fh->step[2] * fh->step[2];
    sumR += fh->step[3] * fh->step[3] + fh->step[4] * fh->step[4] + fh->step[5] * fh->step[5];
    if (pointsOnDevice) continue;  // sumNID / numID come from backupState (they only read idepth_backup)
    ef->pointsOnDeviceCurrent = false;
    for (PointHessian *ph : fh->pointHessians) {
      ph->setIdepth(ph->idepth_backup + stepfacD * ph->step);
      sumID += ph->step * ph->step;
      sumNID += fabsf(ph->idepth_backup);
      numID++;
      ph->setIdepthZero(ph->idepth_backup + stepfacD * ph->step);
    }
  }
  if (pointsOnDevice) { sumNID = backupSumNID; numID = backupNumID; }
  const float nf = (float)frameHessians.size();
  sumA /= nf; sumB /= nf; sumR /= nf; sumT /= nf;
  sumID /= numID; sumNID /= numID;
  (void)sumID;
  ef->EFDeltaValid = false;
  if (precalcOnDevice) {
    hostPrecalcStale = true;  // targetPrecalc (distanceLL, PRE_KRKiTll ...) is refreshed by whoever reads it next (ensureHostPrecalc)
    // the device forms FrameFramePrecalc / adHTdeltaF / cDeltaF itself (k_resub_devstep); what the host's next solve reads are the
    // frame deltas of setDeltaF (OB/EnergyFunctional.cpp:183-190) and cDeltaF (getStitchedDeltaF)
    for (int i = 0; i < 4; i++) ef->cDeltaF[i] = (float)HCalib.value_minus_value_zero[i];
    for (EFFrame *f : ef->frames)
      for (int i = 0; i < 8; i++) {
        f->delta[i] = f->data->state[i] - f->data->state_zero[i];
        f->delta_prior[i] = f->data->state[i];
      }
    ef->EFDeltaValid = true;
  } else {
    PhaseTimer tp(4);
    setPrecalcValues(!pointsOnDevice);
  }
  return sqrtf(sumA) < 0.0005 * setting_thOptIterations && sqrtf(sumB) < 0.00005 * setting_thOptIterations &&
         sqrtf(sumR) < 0.00005 * setting_thOptIterations && sqrtf(sumT) * sumNID < 0.00005 * setting_thOptIterations;
}

void FullSystem::solveSystem(int iteration, double lambda) { rcAcc(ef->solveSystemF(iteration, lambda, &HCalib, false)); }

int FullSystem::prepare() {  // FS/FullSystemOptimize.cpp:316-344
  residentFlush();
  devStepActive = false;
  const bool tmg = getenv("SOS_TIMING") != nullptr;
  const double t0 = now_s();
  // the deltas first (EFPoint::deltaF is part of the point records), then ONE walk over the graph: activeResiduals with
  // resetOOB and the snapshot records
  setPrecalcValues();
  const double t1 = now_s();
  activeResiduals.clear();
  int rc = ef->packWindow(&activeResiduals);
  if (rc) return rc;
  const double t2 = now_s();
  rc = ef->pushState(&HCalib, true, false);  // the points went up with the pack
  if (rc) return rc;
  const double t3 = now_s();
  if (forceAcceptStep) {
    // resetOOB + linearizeAll(false) + applyRes as one launch chain; the threshold statistic comes back with it
    const int n = (int)frameHessians.size();
    std::vector<float> th(n);
    for (int i = 0; i < n; i++) th[i] = frameHessians[i]->frameEnergyTH;
    int cap = 0, cnt = 0;
    sos_ba_newest_capacity(ef->ba, &cap);
    newestE.resize((size_t)cap + 1);
    double E = 0;
    // the Gauss-Newton loop follows: its first accumulate + stitch is enqueued right behind this linearisation (the tile sums of
    // the top Hessian come out of it), unless a callback exchange has to run between accumulate and stitch
    static const bool noPrefetch = getenv("SOS_NO_PREPARE_PREFETCH") != nullptr;  // (tools/first_solve_probe.py: the stored-Jacobian accumulate, for comparison)
    // (2: only the tile sums are formed -- the device-resident loop and a callback exchange enqueue their own accumulate)
    // (a callback exchange gets 0: sos_ba_accumulate_local resets the tile sums and needs J stored, which the fused form skips)
    sos_ba_set_prefetch(ef->ba, (noPrefetch || ef->allreduceHook) ? 0 : residentUsable() ? 2 : 1);
    lastError = sos_ba_linearize_apply(ef->ba, th.data(), 1, &E, newestE.data(), &cnt);
    newestE.resize(cnt);
    setNewFrameEnergyTH(newestE);
    prepareEnergy = E;
  } else {
    sos_ba_reset_oob(ef->ba);
    prepareEnergy = linearizeAll(false);
    if (!forceAcceptStep) {  // lastEnergyL / lastEnergyM of FS/FullSystemOptimize.cpp:335-336, evaluated before the first applyRes
      prepareEnergyL = ef->calcLEnergyF_MT();
      prepareEnergyM = ef->calcMEnergyF();
    }
    applyRes();
  }
  if (tmg) fprintf(stderr, "[prepare] precalc %.0f us, packWindow %.0f us, pushState %.0f us, resetOOB+linearize+apply %.0f us\n", (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (now_s() - t3) * 1e6);
  return lastError;
}

bool FullSystem::gnIteration(int iteration, bool mayContinue) {  // :358-413 with setting_forceAceptStep
  if (residentActive || residentUsable()) {
    flushPointMirrors();
    if (!residentActive) {
      if (residentBegin() != SOS_OK) goto host_path;
      rcAcc(sos_ba_gn_resident_enqueue(ef->ba, &residentQueued));
    }
    // benchmark loops (pipelineAlways) keep the device one iteration ahead of the host
    if (pipelineAlways && residentQueued == residentSeq + 1) rcAcc(sos_ba_gn_resident_enqueue(ef->ba, &residentQueued));
    if (residentQueued == residentSeq) rcAcc(sos_ba_gn_resident_enqueue(ef->ba, &residentQueued));
    const bool cb = residentConsume(residentSeq + 1);
    if (!pipelineAlways) residentFlush();
    lastLoopMode = 2;
    return cb;
  }
host_path:
  if (!devStepActive && devStepUsable()) devStepBegin();  // (prepare() ended the previous one: the host re-uploaded its states)
  if (devStepActive && !pointMirrorsStale && (inOptimizeLoop || pipelineAlways)) beginLazyPointMirrors();
  if (!devStepActive) flushPointMirrors();
  { PhaseTimer tb(7); backupState(); }
  if (rcAcc(ef->solveSystemF(iteration, 1e-1, &HCalib, true)) != SOS_OK) {  // x, frame / calib steps; back-substitution deferred
    isLost = true;  // a failed device call: no step is taken on undelivered H / b
    return true;
  }
  if (devStepActive) {
    // x goes to the device as it is: back-substitution, the frames' new poses, the precalc records and the deltas are formed
    // there inside ONE launch, the linearisation follows without waiting for the host
    bool canbreak;
    lastLoopMode = 1;
    { PhaseTimer t(3); canbreak = doStepFromBackup(1, 1, 1, 1, 1, true, true); }
    PhaseTimer t(5);
    const int n = (int)frameHessians.size();
    std::vector<float> th(n);
    for (int i = 0; i < n; i++) th[i] = frameHessians[i]->frameEnergyTH;
    const sos_calib cal = HCalib.toCalib();
    {
      int cap = 0;
      sos_ba_newest_capacity(ef->ba, &cap);
      newestE.resize((size_t)cap + 1);
    }
    int cnt = 0;
    double E = 0;
    ef->pointStep.resize(ef->allPoints.size());
    const bool more = pipelineAlways || (mayContinue && !(canbreak && iteration >= minOptIterations));
    sos_ba_set_prefetch(ef->ba, (more && !ef->allreduceHook) ? 1 : 0);  // a callback exchange runs between accumulate and stitch
    lastError = sos_ba_gn_step(ef->ba, ef->lastX.data(), 1.0f, &cal, nullptr, nullptr, nullptr, th.data(), 1, &E, newestE.data(), &cnt,
                               ef->pointStep.data());
    if (lastError != SOS_OK) {  // e.g. the device-side frame states were dropped by a state upload in between: no step was taken
      devStepActive = false;
      sos_ba_gn_devstep_end(ef->ba);
      isLost = true;
      return true;
    }
    newestE.resize(cnt);
    PhaseTimer tpost(6);
    if (pointMirrorsStale) {  // ph->setIdepth(ph->idepth_backup + stepfacD * ph->step) on the flat copies; the objects follow in flushPointMirrors
      const size_t P = flatIdepth.size();
      const float *st = ef->pointStep.data(), *bk = flatBackup.data();
      float *id = flatIdepth.data();
      for (size_t k = 0; k < P; k++) id[k] = bk[k] + 1.0f * st[k];
      flatStepped = true;
    } else {
      for (size_t k = 0; k < ef->allPoints.size(); k++) {
        PointHessian *ph = ef->allPoints[k]->data;
        ph->step = ef->pointStep[k];
        ph->setIdepth(ph->idepth_backup + 1.0f * ph->step);
        ph->setIdepthZero(ph->idepth_backup + 1.0f * ph->step);
        ef->allPoints[k]->deltaF = 0;
      }
    }
    setNewFrameEnergyTH(newestE);
    return canbreak;
  }
  // the back-substitution needs x alone: it runs on the device while the host derives the new poses and precalc records
  const bool resubAhead = sos_ba_gn_resub(ef->ba, ef->lastX.data(), 1.0f) == SOS_OK;
  lastLoopMode = 0;
  bool canbreak;
  { PhaseTimer t(3); canbreak = doStepFromBackup(1, 1, 1, 1, 1, true); }
  {
    // resubstitute + point step (device) + new state upload + linearizeAll(false) + applyRes: one round trip
    PhaseTimer t(5);
    const int n = (int)frameHessians.size();
    std::vector<sos_precalc> pc((size_t)n * n);
    for (int h = 0; h < n; h++)
      for (int tt = 0; tt < n; tt++) pc[(size_t)(h + n * tt)] = frameHessians[h]->targetPrecalc[tt].dev;
    std::vector<float> th(n);
    for (int i = 0; i < n; i++) th[i] = frameHessians[i]->frameEnergyTH;
    const sos_calib cal = HCalib.toCalib();
    {
      int cap = 0;
      sos_ba_newest_capacity(ef->ba, &cap);
      newestE.resize((size_t)cap + 1);
    }
    int cnt = 0;
    double E = 0;
    ef->pointStep.resize(ef->allPoints.size());
    // the next iteration's accumulate can be enqueued behind this linearisation when there will be one
    const bool more = pipelineAlways || (mayContinue && !(canbreak && iteration >= minOptIterations));
    sos_ba_set_prefetch(ef->ba, (more && !ef->allreduceHook) ? 1 : 0);
    lastError = sos_ba_gn_step(ef->ba, resubAhead ? nullptr : ef->lastX.data(), 1.0f, &cal, pc.data(), ef->adHTdeltaF.data(), ef->cDeltaF, th.data(),
                               1, &E, newestE.data(), &cnt, ef->pointStep.data());
    newestE.resize(cnt);
    PhaseTimer tpost(6);
    // point part of doStepFromBackup on the host mirrors (FS/FullSystemOptimize.cpp:207-213)
    for (size_t k = 0; k < ef->allPoints.size(); k++) {
      PointHessian *ph = ef->allPoints[k]->data;
      ph->step = ef->pointStep[k];
      ph->setIdepth(ph->idepth_backup + 1.0f * ph->step);
      ph->setIdepthZero(ph->idepth_backup + 1.0f * ph->step);
      ef->allPoints[k]->deltaF = 0;
    }
    setNewFrameEnergyTH(newestE);
  }
  return canbreak;
}

// ------------------------------------------------------------------------------------------------
// device-resident Gauss-Newton loop: what is left for the host is the loop control (canbreak) and its mirrors
// ------------------------------------------------------------------------------------------------
bool FullSystem::devStepUsable() const {
  // x is replicated over the ranks (identical stitch + solve on the all-reduced accumulator) and so are the frame states: the
  // device-side step needs nothing from the exchange.  The IMU branch of the solve (OB/EnergyFunctional.cpp:1053-1171) sits between
  // stitch and back-substitution on the host and delivers the same x vector.
  return devStepAllowed && forceAcceptStep && !ef->keepSystem && (int)frameHessians.size() <= 17 && !ef->allPoints.empty();
}

int FullSystem::devStepBegin() {
  const int n = (int)frameHessians.size();
  std::vector<sos_gn_frame> fr(n);
  for (int i = 0; i < n; i++) {
    const FrameHessian *fh = frameHessians[i];
    fh->camToWorld_evalPT.to12(fr[i].camToWorld_evalPT);
    std::memcpy(fr[i].state, fh->state, sizeof(double) * 10);
    std::memcpy(fr[i].state_zero, fh->state_zero, sizeof(double) * 10);
    std::memset(fr[i].prior, 0, sizeof(fr[i].prior));
    fr[i].ab_exposure = fh->ab_exposure;
    fr[i].pad = 0;
  }
  devStepActive = sos_ba_gn_devstep_begin(ef->ba, fr.data(), HCalib.value, HCalib.value_zero) == SOS_OK;
  return devStepActive ? SOS_OK : SOS_ERR_STATE;
}

bool FullSystem::residentUsable() const {
  // (with a communicator attached the chain carries the exchange itself -- the library enqueues the all-reduce of the packed accumulator
  // and the all-gather of the newest frame's energies on its stream --, so N = 1 and N > 1 run the same loop; a callback exchange
  // (allreduceHook) needs the host between accumulate and stitch and stays on the default loop)
  return residentAllowed && forceAcceptStep && !ef->imuSettings && !ef->allreduceHook && !ef->keepSystem &&
         sos_ba_gn_resident_supported(ef->ba) == 1;
}

int FullSystem::residentBegin() {
  const int n = (int)frameHessians.size();
  std::vector<sos_gn_frame> fr(n);
  for (int i = 0; i < n; i++) {
    const FrameHessian *fh = frameHessians[i];
    fh->camToWorld_evalPT.to12(fr[i].camToWorld_evalPT);
    std::memcpy(fr[i].state, fh->state, sizeof(double) * 10);
    std::memcpy(fr[i].state_zero, fh->state_zero, sizeof(double) * 10);
    std::memcpy(fr[i].prior, fh->efFrame->prior, sizeof(double) * 8);
    fr[i].ab_exposure = fh->ab_exposure;
    fr[i].pad = 0;
  }
  std::vector<float> th(n);
  for (int i = 0; i < n; i++) th[i] = frameHessians[i]->frameEnergyTH;
  const int rc = sos_ba_gn_resident_begin(ef->ba, fr.data(), HCalib.value, HCalib.value_zero, ef->cPrior[0], ef->HM.data(), ef->bM.data(), th.data());
  residentActive = rc == SOS_OK;
  residentSeq = residentQueued = 0;
  return rc;
}

bool FullSystem::residentConsume(int seq) {
  const int n = (int)frameHessians.size(), dim = SOS_CPARS + 8 * n;
  double hdr[16];
  std::vector<double> x(dim);
  const int rc = sos_ba_gn_resident_wait(ef->ba, seq, hdr, x.data());
  residentSeq = seq;   // consumed whatever it brought: residentFlush's loop over the queued iterations must advance past a failed one
  if (rc != SOS_OK) {  // a non-positive pivot of the unpivoted device solve (SOS_ERR_STATE) or a wait that timed out: no usable step --
    rcAcc(rc);         // the system is reported lost (the kernels behind the solve have stepped the device's copies with that x: the
    isLost = true;     // window is not continued from them)
    return true;
  }
  if (getenv("SOS_TIMING")) fprintf(stderr, "[k_gn_solve] assemble %.1f factorise %.1f substitute %.1f us\n", hdr[11], hdr[12], hdr[13]);
  ef->lastX = x;
  ef->resInA = (int)hdr[8];
  ef->resInL = (int)hdr[9];
  // the host's copies of the states follow x exactly as in the device-side-step loop (backupState + doStepFromBackup,
  // FS/FullSystemOptimize.cpp:185-269); the points stay on the device until residentFlush, their |idepth| sum comes with the slot
  std::memcpy(HCalib.value_backup, HCalib.value, sizeof(HCalib.value));
  for (int i = 0; i < 4; i++) HCalib.step[i] = -x[i];
  for (int h = 0; h < n; h++) {
    FrameHessian *fh = frameHessians[h];
    std::memcpy(fh->state_backup, fh->state, sizeof(fh->state));
    for (int i = 0; i < 8; i++) fh->step[i] = -x[SOS_CPARS + 8 * h + i];
    fh->step[8] = fh->step[9] = 0;
  }
  backupSumNID = (float)hdr[6];
  backupNumID = (float)hdr[7];
  const bool canbreak = doStepFromBackup(1, 1, 1, 1, 1, true, true);
  frameHessians.back()->frameEnergyTH = (float)hdr[10];
  return canbreak;
}

// Lazy point mirrors (sos_host.hpp): the flat copies are gathered once, in snapshot order, with the reference's summation order beside them
bool FullSystem::beginLazyPointMirrors() {
  const size_t P = ef->allPoints.size();
  if (P == 0) return false;
  flatIdepth.resize(P);
  flatOrder.clear();
  flatOrder.reserve(P);
  for (FrameHessian *fh : frameHessians)
    for (PointHessian *ph : fh->pointHessians) {
      if (ph->packIdx < 0 || (size_t)ph->packIdx >= P || ef->allPoints[(size_t)ph->packIdx]->data != ph) return false;  // not the snapshot's point set
      flatOrder.push_back(ph->packIdx);
      flatIdepth[(size_t)ph->packIdx] = ph->idepth;
    }
  if (flatOrder.size() != P) return false;
  pointMirrorsStale = true;
  flatStepped = false;
  return true;
}
void FullSystem::flushPointMirrors() {
  if (!pointMirrorsStale) return;
  pointMirrorsStale = false;
  if (!flatStepped) return;  // no step was taken on the flat copies (a failed solve): the objects are what they were
  const size_t P = std::min(flatIdepth.size(), ef->allPoints.size());
  const bool haveBackup = flatBackup.size() == flatIdepth.size(), haveStep = ef->pointStep.size() >= P;
  for (size_t k = 0; k < P; k++) {
    PointHessian *ph = ef->allPoints[k]->data;
    if (haveBackup) ph->idepth_backup = flatBackup[k];
    if (haveStep) ph->step = ef->pointStep[k];
    ph->setIdepth(flatIdepth[k]);
    ph->setIdepthZero(flatIdepth[k]);
    ef->allPoints[k]->deltaF = 0;
  }
}

int FullSystem::residentFlush() {
  flushPointMirrors();
  if (!residentActive) return SOS_OK;
  while (residentSeq < residentQueued) residentConsume(residentSeq + 1);   // iterations the caller has not asked about yet
  residentActive = false;
  const size_t P = ef->allPoints.size();
  ef->pointStep.resize(P);
  std::vector<float> idp(P);
  int cap = 0, cnt = 0;
  sos_ba_newest_capacity(ef->ba, &cap);
  newestE.resize((size_t)cap + 1);
  double E = 0;
  const int rc = sos_ba_gn_resident_end(ef->ba, ef->pointStep.data(), idp.data(), &E, newestE.data(), &cnt);
  if (rc != SOS_OK) return rcAcc(rc);
  newestE.resize(cnt);
  for (size_t k = 0; k < P; k++) {
    PointHessian *ph = ef->allPoints[k]->data;
    ph->step = ef->pointStep[k];
    ph->setIdepth(idp[k] * (1.0f / SOS_SCALE_IDEPTH));
    ph->setIdepthZero(idp[k] * (1.0f / SOS_SCALE_IDEPTH));
    ef->allPoints[k]->deltaF = 0;
  }
  // host-side per-step state the non-resident calls expect: precalc records / deltas of the final state, and the
  // threshold the last linearisation leaves behind (setNewFrameEnergyTH)
  ef->EFDeltaValid = false;
  setPrecalcValues(false);
  setNewFrameEnergyTH(newestE);
  return SOS_OK;
}

void FullSystem::loadSateBackup() {  // FS/FullSystemOptimize.cpp:271-287 (IMU off)
  ef->pointsOnDeviceCurrent = false;
  HCalib.setValue(HCalib.value_backup);
  for (FrameHessian *fh : frameHessians) {
    fh->setState(fh->state_backup);
    for (PointHessian *ph : fh->pointHessians) {
      ph->setIdepth(ph->idepth_backup);
      ph->setIdepthZero(ph->idepth_backup);
    }
  }
  ef->EFDeltaValid = false;
  setPrecalcValues();
}

// One loop body of FS/FullSystemOptimize.cpp:358-413 with setting_forceAceptStep OFF: the step is evaluated with the
// two-step device protocol (sos_ba_linearize fills PointFrameResidual::J, nothing is committed), accepted with applyRes
// when the total energy decreased, otherwise undone with loadSateBackup and the window is linearised again at the old state.
bool FullSystem::gnIterationChecked(int iteration, double &lastE, double &lastEL, double &lastEM) {
  lastLoopMode = 3;
  devStepActive = false;  // (this loop steps on the host and uploads the states)
  flushPointMirrors();
  backupState();
  if (rcAcc(ef->solveSystemF(iteration, 1e-1, &HCalib, false)) != SOS_OK) {
    isLost = true;
    return true;
  }
  const bool canbreak = doStepFromBackup(1, 1, 1, 1, 1, false);
  rcAcc(ef->pushState(&HCalib, false));
  const double newE = linearizeAll(false);
  const double newEL = ef->calcLEnergyF_MT(), newEM = ef->calcMEnergyF();
  if (newE + newEL + newEM < lastE + lastEL + lastEM) {
    applyRes();
    lastE = newE; lastEL = newEL; lastEM = newEM;
  } else {
    loadSateBackup();
    rcAcc(ef->pushState(&HCalib, false));
    lastE = linearizeAll(false);
    lastEL = ef->calcLEnergyF_MT();
    lastEM = ef->calcMEnergyF();
    stepsRejected++;
  }
  return canbreak;
}

float FullSystem::optimize(int mnumOptIts, int *iterations) {
  if (iterations) *iterations = 0;
  if (frameHessians.size() < 2) return 0;
  if (frameHessians.size() < 3) mnumOptIts = 20;
  if (frameHessians.size() < 4) mnumOptIts = 15;
  const bool tmg = getenv("SOS_TIMING") != nullptr;
  const double tp0 = now_s();
  if (prepare() != SOS_OK) return NAN;
  const double tp1 = now_s();
  int it = 0;
  stepsRejected = 0;
  double lastE = prepareEnergy, lastEL = prepareEnergyL, lastEM = prepareEnergyM;
  if (residentUsable() && residentBegin() == SOS_OK) {
    // the whole loop body runs on the device; the host learns x / canbreak of iteration k while its back-substitution and
    // linearisation are still running and enqueues iteration k + 1 behind them
    rcAcc(sos_ba_gn_resident_enqueue(ef->ba, &residentQueued));
    lastLoopMode = 2;
    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
      const bool canbreak = residentConsume(residentQueued);
      it++;
      if (isLost || (canbreak && iteration >= minOptIterations)) break;
      if (iteration + 1 < mnumOptIts) rcAcc(sos_ba_gn_resident_enqueue(ef->ba, &residentQueued));
    }
    residentFlush();
  } else {
    inOptimizeLoop = true;
    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
      const bool canbreak = forceAcceptStep ? gnIteration(iteration, iteration + 1 < mnumOptIts) : gnIterationChecked(iteration, lastE, lastEL, lastEM);
      it++;
      if (canbreak && iteration >= minOptIterations) break;
    }
    inOptimizeLoop = false;
    flushPointMirrors();
    if (devStepActive) {
      sos_ba_gn_devstep_end(ef->ba);
      devStepActive = false;
    }
  }
  if (iterations) *iterations = it;
  const double tp2 = now_s();
  // :415-425
  FrameHessian *last = frameHessians.back();
  double newStateZero[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  newStateZero[6] = last->state[6];
  newStateZero[7] = last->state[7];
  last->setEvalPT(last->PRE_camToWorld, newStateZero);
  ef->EFDeltaValid = ef->EFAdjointsValid = false;
  ef->setAdjointsF(&HCalib);
  setPrecalcValues();
  ef->pushState(&HCalib, true, false);  // the point values on the device are the host's after the loop (or were re-sent by its last pushState)
  const double tp3 = now_s();
  double lastEnergy = linearizeAll(true);
  if (ef->multiRank()) {  // sharded window: the returned rmse is the window's (energy of all shards over the global residual count)
    double v = lastEnergy;
    if (ef->allreduceF64(&v, 1) == SOS_OK) lastEnergy = v;
  }
  const double tp4 = now_s();
  if (tmg) fprintf(stderr, "[optimize] prepare %.0f us, %d iterations %.0f us, adjoints+precalc+push %.0f us, linearizeAll(true) %.0f us\n", (tp1 - tp0) * 1e6, it, (tp2 - tp1) * 1e6, (tp3 - tp2) * 1e6, (tp4 - tp3) * 1e6);
  if (!std::isfinite(lastEnergy)) isLost = true;
  // point results the viewer / tracker read: idepth_hessian
  if (!ef->allPoints.empty()) {
    std::vector<float> idh(ef->allPoints.size()), hdi(ef->allPoints.size()), bds(ef->allPoints.size());
    // the snapshot order is still the one of the last pack: graph edits above only dropped residuals
    sos_ba_get_point_hessian(ef->ba, idh.data(), hdi.data(), bds.data());
    for (FrameHessian *fh : frameHessians)
      for (PointHessian *ph : fh->pointHessians)
        if (ph->packIdx >= 0 && ph->packIdx < (int)idh.size()) {
          ph->idepth_hessian = idh[ph->packIdx];
          if (ph->efPoint) { ph->efPoint->HdiF = hdi[ph->packIdx]; ph->efPoint->bdSumF = bds[ph->packIdx]; }
        }
  }
  return sqrtf((float)(lastEnergy / (SOS_PATTERN_NUM * ef->resInA)));
}

void FullSystem::removeOutliers() {  // FS/FullSystemOptimize.cpp:507-526
  flushPointMirrors();  // the flat API's lazily stepped points: the objects are current before anything reads, edits or reorders them
  for (FrameHessian *fh : frameHessians)
    for (unsigned i = 0; i < fh->pointHessians.size(); i++) {
      PointHessian *ph = fh->pointHessians[i];
      if (ph->residuals.empty()) {
        fh->pointHessiansOut.push_back(ph);
        ph->efPoint->stateFlag = PS_DROP;
        fh->pointHessians[i] = fh->pointHessians.back();
        fh->pointHessians.pop_back();
        i--;
      }
    }
  ef->dropPointsF();
}

// flagPointsForRemoval for an explicit set (FS/FullSystem.cpp:566-601) followed by
// ef->marginalizePointsF (FS/FullSystem.cpp:912)
int FullSystem::marginalizePoints(const std::vector<PointHessian *> &pts, bool alreadyDetached) {
  residentFlush();
  if (pts.empty()) {  // nothing to linearise: only the PS_DROP points leave (ef->dropPointsF, FS/FullSystem.cpp:909)
    ef->dropPointsF();
    return SOS_OK;
  }
  // After optimize() the graph has only lost residuals (its final linearisation, removeOutliers): the snapshot of that optimize() is
  // kept and the lost residuals are marked dead on it -- no second pack + upload per keyframe.  Anything else repacks.
  int rc = SOS_OK;
  if (!ef->syncDropsToDevice()) {
    devStepActive = false;  // the upload below makes the host the source of the states again
    rc = ef->packWindow();
    if (rc) return rc;
    setPrecalcValues();
    rc = ef->pushState(&HCalib, true);
    if (rc) return rc;
  }
  // r->resetOOB(); r->linearize(); r->applyRes(true) for the residuals of the listed points.  The
  // device linearizes every active residual at the current state, which leaves all the others unchanged
  // in value (same state, same thresholds) -- only the listed ones are read back.
  // (after optimize() every surviving residual is IN and active, so re-evaluating the untouched ones at
  // the unchanged state is value-neutral; their host mirrors are not modified and the next pack rebuilds
  // the device copy from them.)
  const int n = (int)frameHessians.size();
  std::vector<float> th(n);
  for (int i = 0; i < n; i++) th[i] = frameHessians[i]->frameEnergyTH;
  const sos_resid_final *rec = nullptr;
  const float *pmax = nullptr;
  const int32_t *pcnt = nullptr;
  double E = 0;
  rc = sos_ba_linearize_final(ef->ba, th.data(), 1, &E, &rec, &pmax, &pcnt, nullptr, nullptr);
  if (rc) return rc;
  std::vector<int32_t> fixIdx;
  for (PointHessian *ph : pts) {
    int ngoodRes = 0;
    for (PointFrameResidual *r : ph->residuals) {
      const int k = r->packIdx;
      r->state_state = (ResState)rec[k].state_state;
      r->state_energy = rec[k].state_energy;
      r->efResidual->isLinearized = false;
      r->efResidual->isActiveAndIsGoodNEW = rec[k].active != 0;
      if (r->efResidual->isActive()) {
        fixIdx.push_back(k);
        r->efResidual->isLinearized = true;
        ngoodRes++;
      }
    }
    (void)ngoodRes;
    ph->efPoint->stateFlag = (ph->idepth_hessian > setting_minIdepthH_marg) ? PS_MARGINALIZE : PS_DROP;
    ph->wasMarginalized = ph->efPoint->stateFlag == PS_MARGINALIZE;
    FrameHessian *host = ph->host;
    if (ph->efPoint->stateFlag == PS_MARGINALIZE) host->pointHessiansMarginalized.push_back(ph);
    else host->pointHessiansOut.push_back(ph);
    for (size_t i = 0; !alreadyDetached && i < host->pointHessians.size(); i++)
      if (host->pointHessians[i] == ph) {
        host->pointHessians[i] = host->pointHessians.back();
        host->pointHessians.pop_back();
        break;
      }
  }
  sos_ba_fix_linearization(ef->ba, fixIdx.data(), (int)fixIdx.size());
  ef->dropPointsF();         // FS/FullSystem.cpp:909 (PS_DROP ones)
  // dropPointsF mutated the graph but the device snapshot still holds the to-be-marginalised points at
  // their packIdx: marginalizePointsF reads them before removing them.
  return ef->marginalizePointsF();  // :912
}

int FullSystem::dropPoints(const std::vector<PointHessian *> &pts) {
  flushPointMirrors();  // the flat API's lazily stepped points: the objects are current before anything reads, edits or reorders them
  for (PointHessian *ph : pts) {
    ph->efPoint->stateFlag = PS_DROP;
    FrameHessian *host = ph->host;
    host->pointHessiansOut.push_back(ph);
    for (size_t i = 0; i < host->pointHessians.size(); i++)
      if (host->pointHessians[i] == ph) {
        host->pointHessians[i] = host->pointHessians.back();
        host->pointHessians.pop_back();
        break;
      }
  }
  ef->dropPointsF();
  return SOS_OK;
}

int FullSystem::marginalizeFrame(FrameHessian *frame) {  // FS/FullSystemMarginalize.cpp:143-236 (backend part)
  flushPointMirrors();  // the flat API's lazily stepped points: the objects are current before anything reads, edits or reorders them
  if (!frame->pointHessians.empty()) return SOS_ERR_STATE;
  const bool tmg = getenv("SOS_TIMING") != nullptr;
  const double tm0 = now_s();
  {
    const int rc = ef->marginalizeFrame(frame->efFrame);
    if (rc != SOS_OK) { isLost = true; return lastError = rc; }
  }
  const double tm1 = now_s();
  // drop all observations of existing points in that frame (:148-176): the residuals registered with it whose point is still active (a
  // removed point's residuals have lost their EFResidual and stay with the point until it is deleted).  Every point has at most one
  // residual per target, so the order of the drops does not matter to any container.
  {
    const std::vector<PointFrameResidual *> obs(frame->targetedBy);
    for (PointFrameResidual *r : obs) {
      if (!r->efResidual) continue;
      PointHessian *ph = r->point;
      if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].first = nullptr;
      else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].first = nullptr;
      ef->dropResidual(r->efResidual);
      for (unsigned i = 0; i < ph->residuals.size(); i++)
        if (ph->residuals[i] == r) {
          ph->residuals[i] = ph->residuals.back();
          ph->residuals.pop_back();
          break;
        }
      delete r;
    }
  }
  const double tm2 = now_s();
  sos_frame_release(ctx, frame->slot);
  slotUsed[frame->slot] = false;
  for (size_t i = 0; i < frameHessians.size(); i++)
    if (frameHessians[i] == frame) {
      frameHessians.erase(frameHessians.begin() + i);
      break;
    }
  for (size_t i = 0; i < frameHessians.size(); i++) frameHessians[i]->idx = (int)i;
  // the frame owns its marginalised / dropped points: their flat-API indices must not resolve any more
  for (PointHessian *ph : frame->pointHessiansMarginalized)
    if (ph->userIdx >= 0 && ph->userIdx < (int)userPoints.size()) userPoints[ph->userIdx] = nullptr;
  for (PointHessian *ph : frame->pointHessiansOut)
    if (ph->userIdx >= 0 && ph->userIdx < (int)userPoints.size()) userPoints[ph->userIdx] = nullptr;
  delete frame;  // (the reference hands it to LoopHandler instead, src/LoopClosure/LoopHandler.cpp:249)
  const double tm3 = now_s();
  setPrecalcValues();
  ef->setAdjointsF(&HCalib);
  ef->setDeltaF(&HCalib);
  if (tmg) fprintf(stderr, "[marginalizeFrame] prior algebra %.0f us, residual drop walk %.0f us, release + delete frame %.0f us, precalc / adjoints / deltas %.0f us\n",
                   (tm1 - tm0) * 1e6, (tm2 - tm1) * 1e6, (tm3 - tm2) * 1e6, (now_s() - tm3) * 1e6);
  return SOS_OK;
}

// ------------------------------------------------------------------------------------------------
// keyframe-rate host logic of FullSystem::makeKeyFrame around the backend (FS/FullSystem.cpp:783-931)
// ------------------------------------------------------------------------------------------------
static const float setting_minPointsRemaining = 0.05f;    // util/settings.cpp:67-70
static const float setting_maxLogAffFacInWindow = 0.7f;
static const int setting_minFrames = 5, setting_maxFrames = 7, setting_minFrameAge = 1;  // :73-75
static const int setting_minGoodActiveResForMarg = 3, setting_minGoodResForMarg = 4;     // :95-96

// FS/FullSystemMarginalize.cpp:53-133; called BEFORE the new keyframe joins frameHessians (FS/FullSystem.cpp:798)
// The decision of FullSystem::flagFramesForMarginalization (FS/FullSystemMarginalize.cpp:54-141) on plain arrays: per keyframe (window
// order) frameID, the points it still hosts (active + immature) and the ones it lost (marginalised + dropped), refToFh[0] of
// AffLight::fromToVecExposure(newest -> keyframe), and distanceLL[h * n + t] = targetPrecalc[t].distanceLL of keyframe h.  Sets
// flagged[h] = 1 where the reference sets flaggedForMarginalization (entries already set stay set).
static void flag_frames_decision(int n, const int *frameID, const int *in, const int *out, const double *refToFh0, const float *distanceLL, uint8_t *flagged) {
  if (setting_minFrameAge > setting_maxFrames) {
    for (int i = setting_maxFrames; i < n; i++) flagged[i - setting_maxFrames] = 1;
    return;
  }
  int nflagged = 0;
  for (int h = 0; h < n; h++) {
    if ((in[h] < setting_minPointsRemaining * (in[h] + out[h]) || fabs(logf((float)refToFh0[h])) > setting_maxLogAffFacInWindow) &&
        n - nflagged > setting_minFrames) {
      flagged[h] = 1;
      nflagged++;
    }
  }
  if (n - nflagged >= setting_maxFrames) {  // marginalize one: the keyframe closest to the others
    double smallestScore = 1;
    int toMarginalize = -1;
    const int latestID = frameID[n - 1];
    for (int h = 0; h < n; h++) {
      if (frameID[h] > latestID - setting_minFrameAge || frameID[h] == 0) continue;
      double distScore = 0;
      for (int t = 0; t < n; t++) {
        if (frameID[t] > latestID - setting_minFrameAge + 1 || t == h) continue;
        distScore += 1 / (1e-5 + distanceLL[(size_t)h * n + t]);
      }
      distScore *= -sqrtf(distanceLL[(size_t)h * n + n - 1]);
      if (distScore < smallestScore) {
        smallestScore = distScore;
        toMarginalize = h;
      }
    }
    if (toMarginalize >= 0) flagged[toMarginalize] = 1;  // (the reference dereferences unconditionally)
  }
}

void FullSystem::flagFramesForMarginalization() {
  ensureHostPrecalc();  // distanceLL of the current states
  const int n = (int)frameHessians.size();
  std::vector<int> ids(n), in(n), out(n);
  std::vector<double> ref0(n);
  std::vector<float> dist((size_t)n * n, 0.f);
  std::vector<uint8_t> fl(n, 0);
  for (int h = 0; h < n; h++) {
    const FrameHessian *fh = frameHessians[h];
    ids[h] = fh->frameID;
    in[h] = (int)fh->pointHessians.size() + fh->numImmature;
    out[h] = (int)fh->pointHessiansMarginalized.size() + (int)fh->pointHessiansOut.size();
    double refToFh[2];
    AffLight::fromToVecExposure(frameHessians.back()->ab_exposure, fh->ab_exposure, frameHessians.back()->aff_g2l(), fh->aff_g2l(), refToFh);
    ref0[h] = refToFh[0];
    for (size_t t = 0; t < fh->targetPrecalc.size() && (int)t < n; t++) dist[(size_t)h * n + t] = fh->targetPrecalc[t].distanceLL;
    fl[h] = fh->flaggedForMarginalization ? 1 : 0;
  }
  flag_frames_decision(n, ids.data(), in.data(), out.data(), ref0.data(), dist.data(), fl.data());
  for (int h = 0; h < n; h++)
    if (fl[h]) frameHessians[h]->flaggedForMarginalization = true;
}

// the loop "add new residuals for old points" of makeKeyFrame, FS/FullSystem.cpp:818-832
int FullSystem::addResidualsToNewestFrame() {
  flushPointMirrors();  // the flat API's lazily stepped points: the objects are current before anything reads, edits or reorders them
  FrameHessian *fh = frameHessians.back();
  int added = 0;
  for (FrameHessian *fh1 : frameHessians) {
    if (fh1 == fh) continue;
    for (PointHessian *ph : fh1->pointHessians) {
      PointFrameResidual *r = new PointFrameResidual();
      r->point = ph; r->host = fh1; r->target = fh;
      r->registerTarget();
      r->state_state = IN;                      // r->setState(ResState::IN)
      ph->residuals.push_back(r);
      ef->insertResidual(r);
      ph->lastResiduals[1] = ph->lastResiduals[0];
      ph->lastResiduals[0] = std::make_pair(r, IN);
      added++;
    }
  }
  return added;
}

// the tail of FullSystem::optimizeImmaturePoint (FS/FullSystemOptPoint.cpp:151-185) and of activatePointsMT
// (FS/FullSystem.cpp:497-505) for one candidate the device activated: PointHessian(rawPoint), residuals towards the
// keyframes whose bit is set in inMask (frame idx order), lastResiduals as the reference leaves them
PointHessian *FullSystem::addActivatedPoint(const sos_point &p, uint32_t inMask) {
  flushPointMirrors();  // the flat API's lazily stepped points: the objects are current before anything reads, edits or reorders them
  if (p.host < 0 || p.host >= (int)frameHessians.size()) return nullptr;
  PointHessian *ph = new PointHessian();
  ph->host = frameHessians[p.host];
  ph->hasDepthPrior = false;
  ph->u = p.u; ph->v = p.v;
  std::memcpy(ph->color, p.color, sizeof(ph->color));
  std::memcpy(ph->weights, p.weights, sizeof(ph->weights));
  ph->lastResiduals[0] = std::make_pair((PointFrameResidual *)nullptr, OOB);
  ph->lastResiduals[1] = std::make_pair((PointFrameResidual *)nullptr, OOB);
  ph->setIdepthZero(p.idepth_scaled * (1.0f / SOS_SCALE_IDEPTH));
  ph->setIdepth(p.idepth_scaled * (1.0f / SOS_SCALE_IDEPTH));
  const int nf = (int)frameHessians.size();
  FrameHessian *newest = frameHessians.back(), *second = nf < 2 ? nullptr : frameHessians[nf - 2];
  for (int t = 0; t < nf; t++) {
    if (!((inMask >> t) & 1u) || frameHessians[t] == ph->host) continue;
    PointFrameResidual *r = new PointFrameResidual();
    r->point = ph; r->host = ph->host; r->target = frameHessians[t];
    r->registerTarget();
    r->state_NewEnergy = r->state_energy = 0;
    r->state_NewState = OUTLIER;
    r->state_state = IN;
    ph->residuals.push_back(r);
    if (r->target == newest) ph->lastResiduals[0] = std::make_pair(r, IN);
    else if (r->target == second) ph->lastResiduals[1] = std::make_pair(r, IN);
  }
  ph->host->pointHessians.push_back(ph);
  ph->userIdx = (int)userPoints.size();
  userPoints.push_back(ph);
  ef->insertPoint(ph);
  for (PointFrameResidual *r : ph->residuals) ef->insertResidual(r);
  return ph;
}

static bool pointIsOOB(const PointHessian *ph, const std::vector<FrameHessian *> &toMarg) {  // FS/HessianBlocks.h:619-643
  int visInToMarg = 0;
  for (const PointFrameResidual *r : ph->residuals) {
    if (r->state_state != IN) continue;
    for (const FrameHessian *k : toMarg)
      if (r->target == k) visInToMarg++;
  }
  if ((int)ph->residuals.size() >= setting_minGoodActiveResForMarg && ph->numGoodResiduals > setting_minGoodResForMarg + 10 &&
      (int)ph->residuals.size() - visInToMarg < setting_minGoodActiveResForMarg)
    return true;
  if (ph->lastResiduals[0].second == OOB) return true;
  if (ph->residuals.size() < 2) return false;
  if (ph->lastResiduals[0].second == OUTLIER && ph->lastResiduals[1].second == OUTLIER) return true;
  return false;
}

// FullSystem::flagPointsForRemoval (FS/FullSystem.cpp:535-614) followed by ef->dropPointsF and ef->marginalizePointsF
// (makeKeyFrame :908-912).  The decisions are the reference's; the per-residual work of the points that get
// marginalised (resetOOB, linearize, applyRes, fixLinearizationF) runs on the device for the whole set at once.
int FullSystem::flagPointsForRemoval(int *nMarg, int *nDrop) {
  flushPointMirrors();
  std::vector<FrameHessian *> fhsToMargPoints;
  for (FrameHessian *fh : frameHessians)
    if (fh->flaggedForMarginalization) fhsToMargPoints.push_back(fh);
  std::vector<PointHessian *> inliers;  // (isOOB || host flagged) && isInlierNew: linearised, then marginalised or dropped
  int dropped = 0;
  for (FrameHessian *host : frameHessians) {
    for (size_t i = 0; i < host->pointHessians.size(); i++) {
      PointHessian *ph = host->pointHessians[i];
      if (ph->idepth_scaled < 0 || ph->residuals.empty()) {
        host->pointHessiansOut.push_back(ph);
        ph->efPoint->stateFlag = PS_DROP;
        host->pointHessians[i] = nullptr;
        dropped++;
      } else if (pointIsOOB(ph, fhsToMargPoints) || host->flaggedForMarginalization) {
        const bool isInlierNew = (int)ph->residuals.size() >= setting_minGoodActiveResForMarg && ph->numGoodResiduals >= setting_minGoodResForMarg;
        if (isInlierNew) {
          inliers.push_back(ph);  // stays in host->pointHessians until marginalizePoints moves it
        } else {
          host->pointHessiansOut.push_back(ph);
          ph->efPoint->stateFlag = PS_DROP;
          host->pointHessians[i] = nullptr;
          dropped++;
        }
      }
    }
    // compaction of the reference: holes are filled from the back, in index order (:604-611); the points queued for
    // marginalisation keep their slot until marginalizePoints takes them out one by one the same way
  }
  // the reference removes dropped and to-be-marginalised points in ONE pass per host (hole filled from the back); to end
  // with the same pointHessians order, mark the inliers as holes too and compact once
  for (PointHessian *ph : inliers) {
    FrameHessian *host = ph->host;
    for (size_t i = 0; i < host->pointHessians.size(); i++)
      if (host->pointHessians[i] == ph) { host->pointHessians[i] = nullptr; break; }
  }
  for (FrameHessian *host : frameHessians)
    for (int i = 0; i < (int)host->pointHessians.size(); i++)
      if (host->pointHessians[i] == nullptr) {
        host->pointHessians[i] = host->pointHessians.back();
        host->pointHessians.pop_back();
        i--;
      }
  const int rc = marginalizePoints(inliers, /*alreadyDetached=*/true);
  int margd = 0;
  for (PointHessian *ph : inliers)
    if (ph->wasMarginalized) margd++;
  if (nMarg) *nMarg = margd;
  if (nDrop) *nDrop = dropped + ((int)inliers.size() - margd);
  return rc;
}

// the loop "Marginalize Frames" of makeKeyFrame, FS/FullSystem.cpp:926-931
int FullSystem::marginalizeFlaggedFrames(int cap, int32_t *frameIDs, double *camToWorld12, int *count) {
  int k = 0;
  for (unsigned i = 0; i < frameHessians.size(); i++)
    if (frameHessians[i]->flaggedForMarginalization) {
      FrameHessian *fh = frameHessians[i];
      if (k < cap) {
        if (frameIDs) frameIDs[k] = fh->frameID;
        if (camToWorld12) fh->PRE_camToWorld.to12(camToWorld12 + 12 * (size_t)k);
      }
      k++;
      const int rc = marginalizeFrame(fh);
      if (rc != SOS_OK) return rc;
      i = 0;  // (sic: the reference restarts at index 1 after the increment; flagged frame 0 would be caught first)
    }
  if (count) *count = k;
  return SOS_OK;
}

// ================================================================================================
// CoarseTracker / ScaleOptimizer host loops
// ================================================================================================
CoarseTracker::CoarseTracker(sos_ctx *c, const sos_params &p) : ctx(c), prm(p) {
  levels = sos_ctx_pyr_levels(c);
  for (int l = 0; l < SOS_PYR_LEVELS; l++) pc_n[l] = 0;
  if (sos_tracker_create(c, &prm, &trk) != SOS_OK) trk = nullptr;
}
CoarseTracker::~CoarseTracker() {
  if (trk) sos_tracker_destroy(trk);
}

void CoarseTracker::makeK(const CalibHessian *HCalib) {
  calib = HCalib->toCalib();
  fx[0] = HCalib->fxl(); fy[0] = HCalib->fyl(); cx[0] = HCalib->cxl(); cy[0] = HCalib->cyl();
  for (int l = 1; l < levels; l++) {
    fx[l] = fx[l - 1] * 0.5;
    fy[l] = fy[l - 1] * 0.5;
    cx[l] = (cx[0] + 0.5) / ((int)1 << l) - 0.5;
    cy[l] = (cy[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  for (int l = 0; l < levels; l++) {  // K^-1 of [fx 0 cx; 0 fy cy; 0 0 1]
    float *k = Ki[l];
    for (int i = 0; i < 9; i++) k[i] = 0;
    k[0] = 1.0f / fx[l]; k[2] = -cx[l] / fx[l];
    k[4] = 1.0f / fy[l]; k[5] = -cy[l] / fy[l];
    k[8] = 1;
  }
}

int CoarseTracker::setCoarseTrackingRefRaw(const FrameHessian *lastRef, int npts, const float *u, const float *v,
                                           const float *idepth, const float *hdi) {
  refFrameID = lastRef->frameID;
  ref_ab_exposure = lastRef->ab_exposure;
  lastRef_aff_g2l = lastRef->aff_g2l();
  firstCoarseRMSE = -1;
  return sos_tracker_set_ref(trk, &calib, lastRef->slot, npts, u, v, idepth, hdi, pc_n);
}

int CoarseTracker::setCoarseTrackingRef(const std::vector<FrameHessian *> &frameHessians) {
  const FrameHessian *lastRef = frameHessians.back();
  std::vector<float> u, v, id, hdi;
  for (FrameHessian *fh : frameHessians)
    for (PointHessian *ph : fh->pointHessians)
      if (ph->lastResiduals[0].first != nullptr && ph->lastResiduals[0].second == IN) {  // FS/CoarseTracker.cpp:64-66
        const PointFrameResidual *r = ph->lastResiduals[0].first;
        if (r->target != lastRef) continue;
        u.push_back(r->centerProjectedTo[0]);
        v.push_back(r->centerProjectedTo[1]);
        id.push_back(r->centerProjectedTo[2]);
        hdi.push_back(ph->efPoint ? ph->efPoint->HdiF : 0.f);
      }
  return setCoarseTrackingRefRaw(lastRef, (int)u.size(), u.data(), v.data(), id.data(), hdi.data());
}

void CoarseTracker::scaleCoarseDepthL0(float scale) { sos_tracker_scale_depth(trk, scale); }

static void rki_of(const SE3 &T, const float *Ki, float *RKi, float *t) {
  float Rf[9];
  for (int i = 0; i < 9; i++) Rf[i] = (float)T.R[i];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) RKi[3 * r + c] = Rf[3 * r] * Ki[c] + Rf[3 * r + 1] * Ki[3 + c] + Rf[3 * r + 2] * Ki[6 + c];
  for (int i = 0; i < 3; i++) t[i] = (float)T.t[i];
}

// what trackNewestCoarse leaves behind for a run of the device loop that finished `nvis` levels (all of them, or the
// levels up to the one where the :527 test fires): lastResiduals / lastInners / lastFlowIndicators, and for a complete run
// the pose, the affine parameters and their sanity tests, FS/CoarseTracker.cpp:535-551
bool CoarseTracker::finishTrack(const sos_track_hyp &h, int nvis, SE3 &lastToNew_out, AffLight &aff_g2l_out, double *lastResiduals) {
  for (int i = 0; i < 5; i++) lastResiduals[i] = NAN;
  for (int j = 0; j < nvis && j < 8; j++) lastResiduals[h.visit_lvl[j]] = h.visit_res[j];
  lastEvals = h.evals;
  const bool complete = !h.aborted && nvis == h.nvisits;
  if (!complete) return false;
  for (int i = 0; i < 5; i++) lastInners[i] = h.lastInners[i];
  for (int i = 0; i < 3; i++) lastFlowIndicators[i] = h.flow[i];
  lastToNew_out = SE3::from12(h.refToNew);
  aff_g2l_out = AffLight(h.aff[0], h.aff[1]);
  const float modeA = prm.affineOptModeA, modeB = prm.affineOptModeB;
  if ((modeA != 0 && (fabsf((float)aff_g2l_out.a) > 1.2)) || (modeB != 0 && (fabsf((float)aff_g2l_out.b) > 200))) return false;
  double rel[2];
  AffLight::fromToVecExposure(ref_ab_exposure, new_ab_exposure_last, lastRef_aff_g2l, aff_g2l_out, rel);
  const float relA = (float)rel[0], relB = (float)rel[1];
  if ((modeA == 0 && (fabsf(logf(relA)) > 1.5)) || (modeB == 0 && (fabsf(relB) > 200))) return false;
  if (modeA < 0) aff_g2l_out.a = 0;
  if (modeB < 0) aff_g2l_out.b = 0;
  return true;
}

void CoarseTracker::makeTrackTries(const SE3 &slast_2_sprelast, const SE3 &lastF_2_slast, const SE3 *lastF_2_fh_imu, bool posesValid,
                                   std::vector<SE3> &tries) {
  tries.clear();
  if (!posesValid) {  // :208-211
    tries.push_back(SE3());
    return;
  }
  const SE3 fh_2_slast = slast_2_sprelast;  // assumed to be the same as fh_2_slast
  if (lastF_2_fh_imu) tries.push_back(*lastF_2_fh_imu);
  const SE3 inv = fh_2_slast.inverse();
  tries.push_back(inv * lastF_2_slast);        // constant motion
  tries.push_back(inv * inv * lastF_2_slast);  // double motion (frame skipped)
  double lg[6];
  fh_2_slast.log(lg);
  for (int i = 0; i < 6; i++) lg[i] *= 0.5;
  tries.push_back(SE3::exp(lg).inverse() * lastF_2_slast);  // half motion
  tries.push_back(lastF_2_slast);                           // zero motion
  tries.push_back(SE3());                                   // zero motion from the keyframe
  const SE3 lastF_2_fh_const = inv * lastF_2_slast;
  static const float rot_signs[26][3] = {{1, 0, 0},   {0, 1, 0},   {0, 0, 1},   {-1, 0, 0},   {0, -1, 0},  {0, 0, -1},  {1, 1, 0},
                                         {0, 1, 1},   {1, 0, 1},   {-1, 1, 0},  {0, -1, 1},   {-1, 0, 1},  {1, -1, 0},  {0, 1, -1},
                                         {1, 0, -1},  {-1, -1, 0}, {0, -1, -1}, {-1, 0, -1},  {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1},
                                         {-1, 1, 1},  {1, -1, -1}, {1, -1, 1},  {1, 1, -1},   {1, 1, 1}};
  for (float rot_delta = 0.02; rot_delta < 0.05; rot_delta += 0.01) {  // the float loop variable of :199 (three passes)
    for (int k = 0; k < 26; k++) {
      // Sophus::SO3(Quaterniond(1, x, y, z)) normalises the quaternion; then the rotation matrix of the unit quaternion
      double q[4] = {1, (double)(rot_signs[k][0] * rot_delta), (double)(rot_signs[k][1] * rot_delta), (double)(rot_signs[k][2] * rot_delta)};
      const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int i = 0; i < 4; i++) q[i] /= nrm;
      const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
      const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
      const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy,
                   tzz = tz * qz;
      SE3 D;
      D.R[0] = 1 - (tyy + tzz); D.R[1] = txy - twz; D.R[2] = txz + twy;
      D.R[3] = txy + twz; D.R[4] = 1 - (txx + tzz); D.R[5] = tyz - twx;
      D.R[6] = txz - twy; D.R[7] = tyz + twx; D.R[8] = 1 - (txx + tyy);
      tries.push_back(lastF_2_fh_const * D);
    }
  }
}

int CoarseTracker::trackHypotheses(int newSlot, float new_ab_exposure, const std::vector<SE3> &tries, const AffLight &aff_last_2_l,
                                   int coarsestLvl, const double *lastCoarseRMSE, double reTrackThreshold, int batch, TrackResult &out) {
  out = TrackResult();
  out.flowVecs[0] = out.flowVecs[1] = out.flowVecs[2] = 100;
  out.lastF_2_fh = SE3();
  out.aff_g2l = AffLight(0, 0);
  double achievedRes[5] = {NAN, NAN, NAN, NAN, NAN};
  if (tries.empty()) return SOS_ERR_ARG;
  if (batch < 1) batch = 1;
  new_ab_exposure_last = new_ab_exposure;
  const double refAff[2] = {lastRef_aff_g2l.a, lastRef_aff_g2l.b};
  const size_t n = tries.size();
  std::vector<sos_track_hyp> hyp;
  bool stop = false;
  for (size_t i0 = 0; i0 < n && !stop;) {
    const size_t cnt = (i0 == 0) ? 1 : std::min<size_t>((size_t)batch, n - i0);
    hyp.assign(cnt, sos_track_hyp());
    for (size_t k = 0; k < cnt; k++) {
      memset(&hyp[k], 0, sizeof(sos_track_hyp));
      tries[i0 + k].to12(hyp[k].refToNew);
      hyp[k].aff[0] = aff_last_2_l.a;
      hyp[k].aff[1] = aff_last_2_l.b;
    }
    bool onDevice = deviceLM;
    if (onDevice) {
      const int rc = sos_tracker_track(trk, newSlot, &Ki[0][0], ref_ab_exposure, new_ab_exposure, refAff, coarsestLvl, achievedRes, (int)cnt, hyp.data());
      if (rc == SOS_ERR_TIMEOUT) {  // the launch could not get all its workgroups resident in time: this batch one by one, host loop
        onDevice = false;
        lmFallbacks++;
      } else if (rc != SOS_OK) return rc;
    }
    out.evaluated += (int)cnt;
    for (size_t k = 0; k < cnt && !stop; k++) {
      SE3 lastF_2_fh_this = tries[i0 + k];
      AffLight aff_g2l_this = aff_last_2_l;
      double currentRes[5];
      bool trackingIsGood;
      if (onDevice) {
        // where would the sequential loop have stopped this try?  thresholds only tighten, so never later than the device did
        const sos_track_hyp &h = hyp[k];
        int nvis = h.nvisits;
        for (int j = 0; j < h.nvisits && j < 8; j++)
          if (h.visit_res[j] > 1.5 * achievedRes[h.visit_lvl[j]]) { nvis = j + 1; break; }
        sos_track_hyp cut = h;
        if (nvis < h.nvisits) cut.aborted = 1;
        trackingIsGood = finishTrack(cut, nvis, lastF_2_fh_this, aff_g2l_this, currentRes);
      } else {
        const bool keep = deviceLM;
        deviceLM = false;
        trackingIsGood = trackNewestCoarse(newSlot, new_ab_exposure, lastF_2_fh_this, aff_g2l_this, coarsestLvl, achievedRes, currentRes);
        deviceLM = keep;
      }
      out.tryIterations++;
      if (trackingIsGood && std::isfinite((float)currentRes[0]) && !(currentRes[0] >= achievedRes[0])) {  // a new winner, :239-247
        for (int q = 0; q < 3; q++) out.flowVecs[q] = lastFlowIndicators[q];
        out.aff_g2l = aff_g2l_this;
        out.lastF_2_fh = lastF_2_fh_this;
        out.haveOneGood = true;
        out.chosen = (int)(i0 + k);
      }
      if (out.haveOneGood) {  // take over achieved res (always), :250-257
        for (int q = 0; q < 5; q++)
          if (!std::isfinite((float)achievedRes[q]) || achievedRes[q] > currentRes[q]) achievedRes[q] = currentRes[q];
      }
      if (out.haveOneGood && achievedRes[0] < lastCoarseRMSE[0] * reTrackThreshold) stop = true;
    }
    i0 += cnt;
  }
  if (!out.haveOneGood) {  // :276-283
    out.flowVecs[0] = out.flowVecs[1] = out.flowVecs[2] = 0;
    out.aff_g2l = aff_last_2_l;
    out.lastF_2_fh = tries[0];
  }
  for (int q = 0; q < 5; q++) out.achievedRes[q] = achievedRes[q];
  if (firstCoarseRMSE < 0) firstCoarseRMSE = achievedRes[0];
  return SOS_OK;
}

bool CoarseTracker::trackNewestCoarse(int newSlot, float new_ab_exposure, SE3 &lastToNew_out, AffLight &aff_g2l_out,
                                      int coarsestLvl, const double *minResForAbort, double *lastResiduals) {
  new_ab_exposure_last = new_ab_exposure;
  if (deviceLM) {
    sos_track_hyp hyp;
    memset(&hyp, 0, sizeof(hyp));
    lastToNew_out.to12(hyp.refToNew);
    hyp.aff[0] = aff_g2l_out.a;
    hyp.aff[1] = aff_g2l_out.b;
    for (int i = 0; i < 5; i++) lastResiduals[i] = NAN;
    lastFlowIndicators[0] = lastFlowIndicators[1] = lastFlowIndicators[2] = 1000;
    const double refAff[2] = {lastRef_aff_g2l.a, lastRef_aff_g2l.b};
    const int rc = sos_tracker_track(trk, newSlot, &Ki[0][0], ref_ab_exposure, new_ab_exposure, refAff, coarsestLvl, minResForAbort, 1, &hyp);
    if (rc == SOS_OK) return finishTrack(hyp, hyp.nvisits, lastToNew_out, aff_g2l_out, lastResiduals);
    if (rc != SOS_ERR_TIMEOUT) return false;
    lmFallbacks++;  // the one-launch loop gave up waiting for its workgroups: the same loop around the device passes, below
  }
  for (int i = 0; i < 5; i++) lastResiduals[i] = NAN;
  lastFlowIndicators[0] = lastFlowIndicators[1] = lastFlowIndicators[2] = 1000;
  const int maxIterations[] = {10, 20, 50, 50, 50};
  const float lambdaExtrapolationLimit = 0.001;
  SE3 refToNew_current = lastToNew_out;
  AffLight aff_g2l_current = aff_g2l_out;
  bool haveRepeated = false;
  const float modeA = prm.affineOptModeA, modeB = prm.affineOptModeB;
  sos_tracker_set_gs_hint(trk, 1, (float)lastRef_aff_g2l.b);  // calcGSSSE rides behind every calcRes (accepted steps: 1 round trip)
  bool devError = false;
  lastEvals = 0;
  auto calcRes = [&](int lvl, const SE3 &T, const AffLight &aff, float cutoff, double *rs, float *a_out) {
    float RKi[9], t[3], affLL[2];
    rki_of(T, Ki[lvl], RKi, t);
    double a2[2];
    AffLight::fromToVecExposure(ref_ab_exposure, new_ab_exposure, lastRef_aff_g2l, aff, a2);
    affLL[0] = (float)a2[0];
    affLL[1] = (float)a2[1];
    if (a_out) *a_out = affLL[0];
    lastEvals++;
    if (sos_tracker_calc_res(trk, lvl, newSlot, RKi, t, affLL, cutoff, rs) != SOS_OK) {
      devError = true;  // a failed device call must not leave uninitialised sums behind: the loop sees a lost track
      rs[0] = NAN; rs[1] = 0; rs[2] = rs[3] = rs[4] = NAN; rs[5] = 0;
    }
  };
  for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
    double H[64], b[8], resOld[6], resNew[6];
    float levelCutoffRepeat = 1, a_cur = 1;
    calcRes(lvl, refToNew_current, aff_g2l_current, prm.coarseCutoffTH * levelCutoffRepeat, resOld, &a_cur);
    while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
      levelCutoffRepeat *= 2;
      calcRes(lvl, refToNew_current, aff_g2l_current, prm.coarseCutoffTH * levelCutoffRepeat, resOld, &a_cur);
    }
    sos_tracker_calc_gs(trk, lvl, a_cur, (float)lastRef_aff_g2l.b, H, b);
    float lambda = 0.01;
    for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
      std::vector<double> Hl(H, H + 64), nb(8), inc;
      for (int i = 0; i < 8; i++) { Hl[9 * i] *= (1 + lambda); nb[i] = -b[i]; }
      ldlt_solve(Hl, nb, inc, 8);
      if (modeA < 0 && modeB < 0) {  // fix a, b
        std::vector<double> Hs(36), bs(6), xs;
        for (int r = 0; r < 6; r++) { for (int cc = 0; cc < 6; cc++) Hs[6 * r + cc] = Hl[8 * r + cc]; bs[r] = nb[r]; }
        ldlt_solve(Hs, bs, xs, 6);
        for (int r = 0; r < 6; r++) inc[r] = xs[r];
        inc[6] = inc[7] = 0;
      }
      if (!(modeA < 0) && modeB < 0) {  // fix b
        std::vector<double> Hs(49), bs(7), xs;
        for (int r = 0; r < 7; r++) { for (int cc = 0; cc < 7; cc++) Hs[7 * r + cc] = Hl[8 * r + cc]; bs[r] = nb[r]; }
        ldlt_solve(Hs, bs, xs, 7);
        for (int r = 0; r < 7; r++) inc[r] = xs[r];
        inc[7] = 0;
      }
      if (modeA < 0 && !(modeB < 0)) {  // fix a
        std::vector<double> HS(Hl), bS(b, b + 8), Hs(49), bs(7), xs;
        for (int r = 0; r < 8; r++) HS[8 * r + 6] = HS[8 * r + 7];
        for (int cc = 0; cc < 8; cc++) HS[8 * 6 + cc] = HS[8 * 7 + cc];
        bS[6] = bS[7];
        for (int r = 0; r < 7; r++) { for (int cc = 0; cc < 7; cc++) Hs[7 * r + cc] = HS[8 * r + cc]; bs[r] = -bS[r]; }
        ldlt_solve(Hs, bs, xs, 7);
        for (int r = 0; r < 8; r++) inc[r] = 0;
        for (int r = 0; r < 6; r++) inc[r] = xs[r];
        inc[7] = xs[6];
      }
      float extrapFac = 1;
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit / lambda));
      for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
      double incScaled[8];
      for (int i = 0; i < 8; i++) incScaled[i] = inc[i];
      for (int i = 0; i < 3; i++) incScaled[i] *= SOS_SCALE_XI_ROT;      // (sic) FS/CoarseTracker.cpp:455-459
      for (int i = 3; i < 6; i++) incScaled[i] *= SOS_SCALE_XI_TRANS;
      incScaled[6] *= SOS_SCALE_A;
      incScaled[7] *= SOS_SCALE_B;
      double sum = 0;
      for (int i = 0; i < 8; i++) sum += incScaled[i];
      if (!std::isfinite(sum)) for (int i = 0; i < 8; i++) incScaled[i] = 0;
      const SE3 refToNew_new = SE3::exp(incScaled) * refToNew_current;
      AffLight aff_g2l_new = aff_g2l_current;
      aff_g2l_new.a += incScaled[6];
      aff_g2l_new.b += incScaled[7];
      float a_new = 1;
      calcRes(lvl, refToNew_new, aff_g2l_new, prm.coarseCutoffTH * levelCutoffRepeat, resNew, &a_new);
      const bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      if (accept) {
        sos_tracker_calc_gs(trk, lvl, a_new, (float)lastRef_aff_g2l.b, H, b);
        for (int i = 0; i < 6; i++) resOld[i] = resNew[i];
        aff_g2l_current = aff_g2l_new;
        refToNew_current = refToNew_new;
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      double nrm = 0;
      for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
      if (!(std::sqrt(nrm) > 1e-3)) break;
    }
    if (devError) return false;
    lastResiduals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
    lastInners[lvl] = (int)resOld[1];
    lastFlowIndicators[0] = resOld[2];
    lastFlowIndicators[1] = resOld[3];
    lastFlowIndicators[2] = resOld[4];
    if (lastResiduals[lvl] > 1.5 * minResForAbort[lvl]) return false;
    if (levelCutoffRepeat > 1 && !haveRepeated) {
      lvl++;
      haveRepeated = true;
    }
  }
  lastToNew_out = refToNew_current;
  aff_g2l_out = aff_g2l_current;
  if ((modeA != 0 && (fabsf((float)aff_g2l_out.a) > 1.2)) || (modeB != 0 && (fabsf((float)aff_g2l_out.b) > 200))) return false;
  double rel[2];
  AffLight::fromToVecExposure(ref_ab_exposure, new_ab_exposure, lastRef_aff_g2l, aff_g2l_out, rel);
  const float relA = (float)rel[0], relB = (float)rel[1];
  if ((modeA == 0 && (fabsf(logf(relA)) > 1.5)) || (modeB == 0 && (fabsf(relB) > 200))) return false;
  if (modeA < 0) aff_g2l_out.a = 0;
  if (modeB < 0) aff_g2l_out.b = 0;
  return true;
}

int CoarseTracker::setPoints3d(const sos_calib &cam, float matched_ab_exposure, int n, const float *xyz, const float *colors) {
  // makeK(cur_frame->cam); pts = matched_frame->pts_dso; refAffGToL = AffLight(); refAbExposure = matched ab_exposure (:294-304)
  calib = cam;
  for (int l = 0; l < levels; l++)
    for (int q = 0; q < 9; q++) Ki[l][q] = (q % 4 == 0) ? 1.0f : 0.0f;  // the estimator projects with R alone
  lastRef_aff_g2l = AffLight(0, 0);
  ref_ab_exposure = matched_ab_exposure;
  loopPoints = n;
  return sos_tracker_set_points3d(trk, &cam, n, xyz, colors);
}

bool CoarseTracker::poseEstimate(int newSlot, float new_ab_exposure, SE3 &refToNew, int coarsestLvl, float loopDirectThres,
                                 int innerPercent, float *poseError, int *inlierPercent) {
  const double noAbort[5] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY};
  double lastResiduals[5];
  AffLight aff(0, 0);  // aff_g2l_current = AffLight(), :305
  const bool aff_good = trackNewestCoarse(newSlot, new_ab_exposure, refToNew, aff, coarsestLvl, noAbort, lastResiduals);
  const float pose_error = (float)lastResiduals[0];
  const bool low_res = pose_error < loopDirectThres;                                   // :470
  const int inlier_percent = (int)(100 * float(lastInners[0]) / (float)loopPoints);   // :473 (size_t -> float division)
  const bool enough_inlier = inlier_percent > innerPercent;
  if (poseError) *poseError = pose_error;
  if (inlierPercent) *inlierPercent = inlier_percent;
  return aff_good && low_res && enough_inlier;
}

int CoarseTracker::optimizeScaleHyp(int stereoSlot, const SE3 &tfmF0ToF1, const float *K1_0, int n, float *scales, float *errors, int coarsestLvl) {
  if (n < 1) return SOS_ERR_ARG;
  if (!deviceLM) {
    for (int k = 0; k < n; k++) errors[k] = optimizeScale(stereoSlot, tfmF0ToF1, K1_0, scales[k], coarsestLvl);
    return SOS_OK;
  }
  float RKiAll[SOS_PYR_LEVELS * 9], K1All[SOS_PYR_LEVELS * 4], tf[3];
  for (int l = 0; l < levels; l++) {  // FS/ScaleOptimizer.cpp:66-76
    rki_of(tfmF0ToF1, Ki[l], RKiAll + 9 * l, tf);
    K1All[4 * l] = l == 0 ? K1_0[0] : K1All[4 * (l - 1)] * 0.5f;
    K1All[4 * l + 1] = l == 0 ? K1_0[1] : K1All[4 * (l - 1) + 1] * 0.5f;
    K1All[4 * l + 2] = l == 0 ? K1_0[2] : (float)((K1_0[2] + 0.5) / ((int)1 << l) - 0.5);
    K1All[4 * l + 3] = l == 0 ? K1_0[3] : (float)((K1_0[3] + 0.5) / ((int)1 << l) - 0.5);
  }
  std::vector<double> lr((size_t)5 * n);
  const int rc = sos_tracker_optimize_scale(trk, stereoSlot, RKiAll, tf, K1All, coarsestLvl, n, scales, lr.data(), &lastEvals);
  if (rc == SOS_ERR_TIMEOUT) {  // (scales is untouched) one by one with the host loop
    lmFallbacks++;
    const bool keep = deviceLM;
    deviceLM = false;
    for (int k = 0; k < n; k++) errors[k] = optimizeScale(stereoSlot, tfmF0ToF1, K1_0, scales[k], coarsestLvl);
    deviceLM = keep;
    return SOS_OK;
  }
  if (rc != SOS_OK) return rc;
  for (int k = 0; k < n; k++) errors[k] = (float)lr[(size_t)5 * k];
  return SOS_OK;
}

float CoarseTracker::optimizeScaleKF(int stereoSlot, const SE3 &tfmF0ToF1, const float *K1_0, float trackingRefScale, int coarsestLvl, float thres,
                                     ScaleOptState &st, float *scale_error_out) {
  if (thres <= 0) return 1.0;
  float new_scale = 1.0, scale_error = -1;
  if (st.scaleTrapped) {
    new_scale = trackingRefScale;
    if (optimizeScaleHyp(stereoSlot, tfmF0ToF1, K1_0, 1, &new_scale, &scale_error, coarsestLvl) != SOS_OK) scale_error = -1;
  } else {
    float guess[7] = {0.1f, 0.2f, 0.5f, 1, 2, 5, 10}, err[7];
    if (optimizeScaleHyp(stereoSlot, tfmF0ToF1, K1_0, 7, guess, err, coarsestLvl) == SOS_OK)
      for (int k = 0; k < 7; k++)
        if (err[k] > 0 && (scale_error < 0 || scale_error > err[k])) {
          new_scale = guess[k];
          scale_error = err[k];
        }
  }
  const bool succeed = (0 < scale_error) && (scale_error < thres);
  st.fails = succeed ? 0 : st.fails + 1;  // when it fails continuously the scale is re-initialised
  if (st.fails > 5) st.scaleTrapped = 0;
  if (scale_error_out) *scale_error_out = scale_error;
  if (!succeed) return -1.0f;
  if (!st.scaleTrapped) st.scaleTrapped = 1;
  return new_scale;
}

float CoarseTracker::optimizeScale(int stereoSlot, const SE3 &tfmF0ToF1, const float *K1_0, float &scale, int coarsestLvl) {
  double last_residuals[5] = {NAN, NAN, NAN, NAN, NAN};
  const int maxIterations[] = {10, 20, 50, 50, 50};
  const float lambdaExtrapolationLimit = 0.001;
  float scale_current = scale;
  bool haveRepeated = false;
  float fx1[SOS_PYR_LEVELS], fy1[SOS_PYR_LEVELS], cx1[SOS_PYR_LEVELS], cy1[SOS_PYR_LEVELS];
  fx1[0] = K1_0[0]; fy1[0] = K1_0[1]; cx1[0] = K1_0[2]; cy1[0] = K1_0[3];  // FS/ScaleOptimizer.cpp:66-76
  for (int l = 1; l < levels; l++) {
    fx1[l] = fx1[l - 1] * 0.5;
    fy1[l] = fy1[l - 1] * 0.5;
    cx1[l] = (cx1[0] + 0.5) / ((int)1 << l) - 0.5;
    cy1[l] = (cy1[0] + 0.5) / ((int)1 << l) - 0.5;
  }
  if (deviceLM) {  // the whole loop as one launch
    float RKiAll[SOS_PYR_LEVELS * 9], K1All[SOS_PYR_LEVELS * 4], tf[3];
    for (int l = 0; l < levels; l++) {
      rki_of(tfmF0ToF1, Ki[l], RKiAll + 9 * l, tf);
      K1All[4 * l] = fx1[l]; K1All[4 * l + 1] = fy1[l]; K1All[4 * l + 2] = cx1[l]; K1All[4 * l + 3] = cy1[l];
    }
    float s = scale;
    const int rc = sos_tracker_optimize_scale(trk, stereoSlot, RKiAll, tf, K1All, coarsestLvl, 1, &s, last_residuals, &lastEvals);
    if (rc == SOS_OK) {
      scale = s;
      return (float)last_residuals[0];
    }
    if (rc != SOS_ERR_TIMEOUT) return NAN;
    lmFallbacks++;
  }
  lastEvals = 0;
  sos_tracker_set_gs_hint(trk, 1, 0.f);
  for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
    float H = 0, b = 0, levelCutoffRepeat = 1;
    double resOld[6], resNew[6];
    const float K1[4] = {fx1[lvl], fy1[lvl], cx1[lvl], cy1[lvl]};
    float RKi[9], tf[3];
    rki_of(tfmF0ToF1, Ki[lvl], RKi, tf);
    sos_tracker_calc_res_scale(trk, lvl, stereoSlot, RKi, tf, K1, scale_current, prm.coarseCutoffTH * levelCutoffRepeat, resOld);
    lastEvals++;
    while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
      levelCutoffRepeat *= 2;
      lastEvals++;
      sos_tracker_calc_res_scale(trk, lvl, stereoSlot, RKi, tf, K1, scale_current, prm.coarseCutoffTH * levelCutoffRepeat, resOld);
    }
    sos_tracker_calc_gs_scale(trk, lvl, tf, K1, scale_current, &H, &b);
    float lambda = 0.01;
    for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
      float Hl = H;
      Hl *= (1 + lambda);
      float inc = -b / Hl;
      float extrapFac = 1;
      if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit / lambda));
      inc *= extrapFac;
      if (!std::isfinite(inc) || fabs(inc) > scale_current) inc = 0.0;
      const float scale_new = scale_current + inc;
      sos_tracker_calc_res_scale(trk, lvl, stereoSlot, RKi, tf, K1, scale_new, prm.coarseCutoffTH * levelCutoffRepeat, resNew);
      lastEvals++;
      const bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
      if (accept) {
        sos_tracker_calc_gs_scale(trk, lvl, tf, K1, scale_new, &H, &b);
        for (int i = 0; i < 6; i++) resOld[i] = resNew[i];
        scale_current = scale_new;
        lambda *= 0.5;
      } else {
        lambda *= 4;
        if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
      }
      if (!(inc > 1e-3)) break;
    }
    last_residuals[lvl] = sqrtf((float)(resOld[0] / resOld[1]));
    if (levelCutoffRepeat > 1 && !haveRepeated) {
      lvl++;
      haveRepeated = true;
    }
  }
  scale = scale_current;
  return (float)last_residuals[0];
}

}  // namespace sos

// ================================================================================================
// flat C API
// ================================================================================================
using namespace sos;

struct sosf_system {
  FullSystem *fs;
};
sos::FullSystem *sosf_system_full(sosf_system *s) { return s ? s->fs : nullptr; }

extern "C" int sosf_create(const sos_params *params, int device, void *hip_stream, sosf_system **out) {
  if (!params || !out) return SOS_ERR_ARG;
  *out = nullptr;
  FullSystem *fs = new FullSystem(*params, device, hip_stream);
  if (!fs->ok()) {
    int e = fs->lastError ? fs->lastError : SOS_ERR_HIP;
    delete fs;
    return e;
  }
  sosf_system *s = new sosf_system();
  s->fs = fs;
  *out = s;
  return SOS_OK;
}
extern "C" int sosf_destroy(sosf_system *s) {
  if (!s) return SOS_OK;
  delete s->fs;
  delete s;
  return SOS_OK;
}
extern "C" int sosf_set_calib(sosf_system *s, const double *vs) {
  if (!s || !vs) return SOS_ERR_ARG;
  CalibHessian &C = s->fs->HCalib;
  for (int i = 0; i < 4; i++) C.value_zero[i] = 0;
  C.setValueScaled(vs);
  for (int i = 0; i < 4; i++) C.value_zero[i] = C.value[i];
  for (int i = 0; i < 4; i++) C.value_minus_value_zero[i] = 0;
  return SOS_OK;
}
extern "C" int sosf_add_frame(sosf_system *s, const sosf_frame_init *f, const float *image) {
  if (!s || !f || !image) return SOS_ERR_ARG;
  return s->fs->addFrame(f->camToWorld, f->state, f->ab_exposure, f->frameID, f->frameEnergyTH, image) ? SOS_OK : SOS_ERR_STATE;
}
extern "C" int sosf_add_frame_from_slot(sosf_system *s, const sosf_frame_init *f, int slot) {
  if (!s || !f) return SOS_ERR_ARG;
  return s->fs->addFrame(f->camToWorld, f->state, f->ab_exposure, f->frameID, f->frameEnergyTH, nullptr, slot) ? SOS_OK : SOS_ERR_STATE;
}
extern "C" int sosf_flag_frames_for_marginalization(sosf_system *s, const int32_t *numImmature, uint8_t *flagged) {
  if (!s) return SOS_ERR_ARG;
  FullSystem *fs = s->fs;
  for (size_t i = 0; i < fs->frameHessians.size(); i++) fs->frameHessians[i]->numImmature = numImmature ? numImmature[i] : 0;
  fs->flagFramesForMarginalization();
  if (flagged) for (size_t i = 0; i < fs->frameHessians.size(); i++) flagged[i] = fs->frameHessians[i]->flaggedForMarginalization ? 1 : 0;
  return SOS_OK;
}
extern "C" int sosf_add_new_frame_residuals(sosf_system *s, int *count) {
  if (!s || s->fs->frameHessians.empty()) return SOS_ERR_ARG;
  const int c = s->fs->addResidualsToNewestFrame();
  if (count) *count = c;
  return SOS_OK;
}
extern "C" int sosf_add_activated_points(sosf_system *s, int count, const sos_point *pts, const uint32_t *inMask) {
  if (!s || (count && (!pts || !inMask))) return SOS_ERR_ARG;
  for (int i = 0; i < count; i++)
    if (!s->fs->addActivatedPoint(pts[i], inMask[i])) return SOS_ERR_ARG;
  return SOS_OK;
}
extern "C" int sosf_remove_outliers(sosf_system *s, int *dropped) {
  if (!s) return SOS_ERR_ARG;
  const int before = s->fs->ef->nPoints;
  s->fs->removeOutliers();
  if (dropped) *dropped = before - s->fs->ef->nPoints;
  return SOS_OK;
}
extern "C" int sosf_flag_points_for_removal(sosf_system *s, int *nMarginalized, int *nDropped) {
  if (!s) return SOS_ERR_ARG;
  return s->fs->flagPointsForRemoval(nMarginalized, nDropped);
}
extern "C" int sosf_marginalize_flagged_frames(sosf_system *s, int cap, int32_t *frameIDs, double *camToWorld12, int *count) {
  if (!s) return SOS_ERR_ARG;
  return s->fs->marginalizeFlaggedFrames(cap, frameIDs, camToWorld12, count);
}
extern "C" int sosf_get_point_keys(sosf_system *s, int32_t *hostFrameID, float *u, float *v, int32_t *hostIdx) {
  if (!s) return SOS_ERR_ARG;
  size_t k = 0;
  for (FrameHessian *fh : s->fs->frameHessians)
    for (EFPoint *p : fh->efFrame->points) {
      if (hostFrameID) hostFrameID[k] = fh->frameID;
      if (u) u[k] = p->data->u;
      if (v) v[k] = p->data->v;
      if (hostIdx) hostIdx[k] = fh->idx;
      k++;
    }
  return SOS_OK;
}
extern "C" int sosf_get_frame_ids(sosf_system *s, int32_t *frameID, uint8_t *flagged, int32_t *nPoints, int32_t *nMarg, int32_t *nOut) {
  if (!s) return SOS_ERR_ARG;
  size_t k = 0;
  for (FrameHessian *fh : s->fs->frameHessians) {
    if (frameID) frameID[k] = fh->frameID;
    if (flagged) flagged[k] = fh->flaggedForMarginalization ? 1 : 0;
    if (nPoints) nPoints[k] = (int)fh->pointHessians.size();
    if (nMarg) nMarg[k] = (int)fh->pointHessiansMarginalized.size();
    if (nOut) nOut[k] = (int)fh->pointHessiansOut.size();
    k++;
  }
  return SOS_OK;
}
extern "C" int sosf_add_points(sosf_system *s, int count, const sos_point *pts) {
  if (!s || (count && !pts)) return SOS_ERR_ARG;
  for (int i = 0; i < count; i++)
    if (!s->fs->addPoint(pts[i])) return SOS_ERR_ARG;
  return SOS_OK;
}
extern "C" int sosf_add_residuals(sosf_system *s, int count, const sos_resid *res) {
  if (!s || (count && !res)) return SOS_ERR_ARG;
  FullSystem *fs = s->fs;
  for (int i = 0; i < count; i++) {
    const sos_resid &q = res[i];
    if (q.point < 0 || q.point >= (int)fs->userPoints.size() || q.target < 0 || q.target >= (int)fs->frameHessians.size()) return SOS_ERR_ARG;
    PointHessian *ph = fs->userPoints[q.point];
    if (!ph || !ph->efPoint) return SOS_ERR_ARG;  // marginalised / dropped, or its host frame left the window
    fs->addResidual(ph, fs->frameHessians[q.target], q);
  }
  return SOS_OK;
}
extern "C" int sosf_set_prior(sosf_system *s, const double *HM, const double *bM) {
  if (!s || !HM || !bM) return SOS_ERR_ARG;
  EnergyFunctional *ef = s->fs->ef;
  const int dim = SOS_CPARS + 8 * ef->nFrames;
  ef->HM.assign(HM, HM + (size_t)dim * dim);
  ef->bM.assign(bM, bM + dim);
  return SOS_OK;
}
extern "C" int sosf_get_prior(sosf_system *s, double *HM, double *bM) {
  if (!s) return SOS_ERR_ARG;
  EnergyFunctional *ef = s->fs->ef;
  if (HM) std::memcpy(HM, ef->HM.data(), sizeof(double) * ef->HM.size());
  if (bM) std::memcpy(bM, ef->bM.data(), sizeof(double) * ef->bM.size());
  return SOS_OK;
}
extern "C" int sosf_optimize(sosf_system *s, int mnumOptIts, float *rmse, int *iterations) {
  if (!s) return SOS_ERR_ARG;
  const float r = s->fs->optimize(mnumOptIts, iterations);
  if (rmse) *rmse = r;
  return s->fs->lastError;
}
extern "C" int sosf_set_device_step(sosf_system *s, int on) {
  if (!s) return SOS_ERR_ARG;
  s->fs->flushPointMirrors();
  s->fs->devStepAllowed = on != 0;
  if (!on && s->fs->devStepActive) {
    sos_ba_gn_devstep_end(s->fs->ef->ba);
    s->fs->devStepActive = false;
  }
  return SOS_OK;
}
extern "C" int sosf_prepare(sosf_system *s) { return s ? s->fs->prepare() : SOS_ERR_ARG; }
extern "C" int sosf_gn_iteration(sosf_system *s, int iteration, int *canbreak) {
  if (!s) return SOS_ERR_ARG;
  const bool cb = s->fs->gnIteration(iteration, false);
  if (canbreak) *canbreak = cb ? 1 : 0;
  return s->fs->lastError;
}
extern "C" int sosf_set_force_accept_step(sosf_system *s, int on) {
  if (!s) return SOS_ERR_ARG;
  s->fs->forceAcceptStep = on != 0;
  return SOS_OK;
}
extern "C" int sosf_get_rejected_steps(sosf_system *s, int *count) {
  if (!s || !count) return SOS_ERR_ARG;
  *count = s->fs->stepsRejected;
  return SOS_OK;
}
extern "C" int sosf_get_loop_mode(sosf_system *s, int *mode) {
  if (!s || !mode) return SOS_ERR_ARG;
  *mode = s->fs->lastLoopMode;
  return SOS_OK;
}
extern "C" int sosf_set_resident(sosf_system *s, int on) {
  if (!s) return SOS_ERR_ARG;
  s->fs->residentFlush();
  s->fs->residentAllowed = on != 0;
  return SOS_OK;
}
extern "C" int sosf_set_host_threads(int n) {
  if (n < 1 || n > 16) return SOS_ERR_ARG;
  sos::WalkPool::get().setThreads(n);
  return SOS_OK;
}
extern "C" int sosf_get_host_threads(void) { return sos::WalkPool::get().threads(); }
extern "C" int sosf_set_min_opt_iterations(sosf_system *s, int its) {
  if (!s || its < 0) return SOS_ERR_ARG;
  s->fs->minOptIterations = its;
  return SOS_OK;
}

extern "C" int sosf_invalidate_pack(sosf_system *s) {
  if (!s) return SOS_ERR_ARG;
  s->fs->ef->packDirty = s->fs->ef->structDirty = true;
  return SOS_OK;
}

extern "C" int sosf_set_pipeline(sosf_system *s, int on) {
  if (!s) return SOS_ERR_ARG;
  if (!on) s->fs->residentFlush();
  s->fs->pipelineAlways = on != 0;
  if (!on) sos_ba_set_prefetch(s->fs->ef->ba, 0);
  return SOS_OK;
}
extern "C" int sosf_set_comm(sosf_system *s, sos_comm *comm) {
  if (!s) return SOS_ERR_ARG;
  s->fs->comm = comm;
  const int rc = sos_ba_set_comm(s->fs->ef->ba, comm);
  s->fs->ef->commAttached = (rc == SOS_OK && comm != nullptr);
  return rc;
}
extern "C" int sosf_counts(sosf_system *s, int *nF, int *nP, int *nR) {
  if (!s) return SOS_ERR_ARG;
  if (nF) *nF = s->fs->ef->nFrames;
  if (nP) *nP = s->fs->ef->nPoints;
  if (nR) *nR = s->fs->ef->nResiduals;
  return SOS_OK;
}
extern "C" int sosf_get_frame(sosf_system *s, int idx, double *c2w, double *state, double *state_zero, float *th) {
  if (!s || idx < 0 || idx >= (int)s->fs->frameHessians.size()) return SOS_ERR_ARG;
  s->fs->residentFlush();
  const FrameHessian *f = s->fs->frameHessians[idx];
  if (c2w) f->PRE_camToWorld.to12(c2w);
  if (state) std::memcpy(state, f->state, sizeof(double) * 10);
  if (state_zero) std::memcpy(state_zero, f->state_zero, sizeof(double) * 10);
  if (th) *th = f->frameEnergyTH;
  return SOS_OK;
}
extern "C" int sosf_get_calib(sosf_system *s, double *vs) {
  if (!s || !vs) return SOS_ERR_ARG;
  s->fs->residentFlush();
  std::memcpy(vs, s->fs->HCalib.value_scaled, sizeof(double) * 4);
  return SOS_OK;
}
extern "C" int sosf_get_points(sosf_system *s, float *idepth, float *idh, float *mrb, int32_t *ngr) {
  if (!s) return SOS_ERR_ARG;
  s->fs->residentFlush();
  size_t k = 0;
  for (FrameHessian *fh : s->fs->frameHessians)
    for (EFPoint *p : fh->efFrame->points) {
      const PointHessian *ph = p->data;
      if (idepth) idepth[k] = ph->idepth;
      if (idh) idh[k] = ph->idepth_hessian;
      if (mrb) mrb[k] = ph->maxRelBaseline;
      if (ngr) ngr[k] = ph->numGoodResiduals;
      k++;
    }
  return SOS_OK;
}
extern "C" int sosf_get_residuals(sosf_system *s, int32_t *state_state, int32_t *isActive, int32_t *removed) {
  // reported per *user-order* residual is not tracked; this reports the current graph in packing order
  if (!s) return SOS_ERR_ARG;
  size_t k = 0;
  for (FrameHessian *fh : s->fs->frameHessians)
    for (EFPoint *p : fh->efFrame->points)
      for (EFResidual *r : p->residualsAll) {
        if (state_state) state_state[k] = (int)r->data->state_state;
        if (isActive) isActive[k] = r->isActive() ? 1 : 0;
        if (removed) removed[k] = 0;
        k++;
      }
  return (int)k >= 0 ? SOS_OK : SOS_ERR_STATE;
}
extern "C" int sosf_get_residual_ids(sosf_system *s, int32_t *pointAddIdx, int32_t *targetFrameID) {
  if (!s) return SOS_ERR_ARG;
  size_t k = 0;
  for (FrameHessian *fh : s->fs->frameHessians)
    for (EFPoint *p : fh->efFrame->points)
      for (EFResidual *r : p->residualsAll) {
        if (pointAddIdx) pointAddIdx[k] = p->data->userIdx;
        if (targetFrameID) targetFrameID[k] = r->target->frameID;
        k++;
      }
  return SOS_OK;
}
extern "C" int sosf_get_lastX(sosf_system *s, double *x) {
  if (!s || !x) return SOS_ERR_ARG;
  std::memcpy(x, s->fs->ef->lastX.data(), sizeof(double) * s->fs->ef->lastX.size());
  return SOS_OK;
}
extern "C" int sosf_keep_last_system(sosf_system *s, int on) {
  if (!s) return SOS_ERR_ARG;
  s->fs->ef->keepSystem = on != 0;
  return SOS_OK;
}
extern "C" int sosf_get_last_system(sosf_system *s, double *H_top, double *b_top, double *H_sc, double *b_sc) {
  if (!s) return SOS_ERR_ARG;
  EnergyFunctional *ef = s->fs->ef;
  const size_t dim = (size_t)(SOS_CPARS + 8 * ef->nFrames);
  if (ef->keptH.size() != dim * dim) return SOS_ERR_STATE;
  if (H_top) std::memcpy(H_top, ef->keptH.data(), sizeof(double) * dim * dim);
  if (b_top) std::memcpy(b_top, ef->keptb.data(), sizeof(double) * dim);
  if (H_sc) std::memcpy(H_sc, ef->keptHsc.data(), sizeof(double) * dim * dim);
  if (b_sc) std::memcpy(b_sc, ef->keptbsc.data(), sizeof(double) * dim);
  return SOS_OK;
}
extern "C" int sosf_get_stats(sosf_system *s, int *a, int *l, int *m) {
  if (!s) return SOS_ERR_ARG;
  if (a) *a = s->fs->ef->resInA;
  if (l) *l = s->fs->ef->resInL;
  if (m) *m = s->fs->ef->resInM;
  return SOS_OK;
}
static int collect_points(FullSystem *fs, const int32_t *idx, int count, std::vector<PointHessian *> &out) {
  // idx = running index in which the point was added (stable across removals)
  std::unordered_map<int, PointHessian *> byUser;
  for (FrameHessian *fh : fs->frameHessians)
    for (PointHessian *ph : fh->pointHessians) byUser[ph->userIdx] = ph;
  for (int i = 0; i < count; i++) {
    auto it = byUser.find(idx[i]);
    if (it == byUser.end()) return SOS_ERR_ARG;
    out.push_back(it->second);
  }
  return SOS_OK;
}
extern "C" int sosf_get_point_ids(sosf_system *s, int32_t *userIdx) {
  if (!s || !userIdx) return SOS_ERR_ARG;
  size_t k = 0;
  for (FrameHessian *fh : s->fs->frameHessians)
    for (EFPoint *p : fh->efFrame->points) userIdx[k++] = p->data->userIdx;
  return SOS_OK;
}
extern "C" int sosf_marginalize_points(sosf_system *s, const int32_t *pointIdx, int count) {
  if (!s || (count && !pointIdx)) return SOS_ERR_ARG;
  std::vector<PointHessian *> pts;
  int rc = collect_points(s->fs, pointIdx, count, pts);
  if (rc) return rc;
  return s->fs->marginalizePoints(pts);
}
extern "C" int sosf_drop_points(sosf_system *s, const int32_t *pointIdx, int count) {
  if (!s || (count && !pointIdx)) return SOS_ERR_ARG;
  std::vector<PointHessian *> pts;
  int rc = collect_points(s->fs, pointIdx, count, pts);
  if (rc) return rc;
  return s->fs->dropPoints(pts);
}
extern "C" int sosf_marginalize_frame(sosf_system *s, int frameIdx) {
  if (!s || frameIdx < 0 || frameIdx >= (int)s->fs->frameHessians.size()) return SOS_ERR_ARG;
  return s->fs->marginalizeFrame(s->fs->frameHessians[frameIdx]);
}
extern "C" int sosf_ldlt_solve(const double *A, const double *b, double *x, int n, int which) {
  if (!A || !b || !x || n <= 0) return SOS_ERR_ARG;
  std::vector<double> Av(A, A + (size_t)n * n), bv(b, b + n), xv;
  if (which == 0) ldlt_solve(Av, bv, xv, n);
  else ldlt_solve_ref(Av, bv, xv, n);
  std::memcpy(x, xv.data(), sizeof(double) * n);
  return SOS_OK;
}
// The per-keyframe host math that feeds the device, on frames built for the occasion (no system, no device): setEvalPT + setState
// (FS/HessianBlocks.h:217-260), FrameFramePrecalc::set of every pair (FS/HessianBlocks.cpp:90-120), setAdjointsF and setDeltaF's
// adHTdeltaF (OB/EnergyFunctional.cpp:42-103, 163-176).  Pair index h + n t.  Exposed for the CPU test-suite.
extern "C" int sosf_host_frame_math(int n, const double *evalPT12, const double *state_zero10, const double *state10, const float *ab_exposure,
                                    const double *calib_value4, const double *calib_value_zero4, double *camToWorld12, sos_precalc *precalc,
                                    double *adHost, double *adTarget, float *adHTdeltaF) {
  if (n < 1 || n > SOS_MAX_FRAMES || !evalPT12 || !state_zero10 || !state10 || !ab_exposure || !calib_value4 || !calib_value_zero4) return SOS_ERR_ARG;
  CalibHessian HC;
  for (int i = 0; i < 4; i++) HC.value_zero[i] = calib_value_zero4[i];
  HC.setValue(calib_value4);
  std::vector<std::unique_ptr<FrameHessian>> F;
  for (int f = 0; f < n; f++) {
    F.emplace_back(new FrameHessian());
    F[f]->idx = f;
    F[f]->ab_exposure = ab_exposure[f];
    F[f]->setEvalPT(SE3::from12(evalPT12 + 12 * f), state_zero10 + 10 * f);
    F[f]->setState(state10 + 10 * f);
    if (camToWorld12) F[f]->PRE_camToWorld.to12(camToWorld12 + 12 * f);
  }
  std::vector<float> ahf(64), atf(64);
  for (int h = 0; h < n; h++)
    for (int t = 0; t < n; t++) {
      const size_t k = (size_t)(h + n * t);
      if (precalc) {
        FrameFramePrecalc pc;
        pc.set(F[h].get(), F[t].get(), &HC);
        precalc[k] = pc.dev;
      }
      double AH[64], AT[64];
      adjoint_pair(F[h].get(), F[t].get(), AH, AT);
      if (adHost) std::memcpy(adHost + 64 * k, AH, sizeof(AH));
      if (adTarget) std::memcpy(adTarget + 64 * k, AT, sizeof(AT));
      if (adHTdeltaF) {
        for (int i = 0; i < 64; i++) { ahf[i] = (float)AH[i]; atf[i] = (float)AT[i]; }
        ad_ht_delta_pair(F[h].get(), F[t].get(), ahf.data(), atf.data(), adHTdeltaF + 8 * k);
      }
    }
  return SOS_OK;
}
extern "C" int sosf_flag_frames(int n, const int32_t *frameID, const int32_t *pointsIn, const int32_t *pointsOut, const double *refToFh0,
                                const float *distanceLL, uint8_t *flagged) {
  if (n < 1 || !frameID || !pointsIn || !pointsOut || !refToFh0 || !distanceLL || !flagged) return SOS_ERR_ARG;
  std::vector<int> ids(frameID, frameID + n), in(pointsIn, pointsIn + n), out(pointsOut, pointsOut + n);
  flag_frames_decision(n, ids.data(), in.data(), out.data(), refToFh0, distanceLL, flagged);
  return SOS_OK;
}
extern "C" int sosf_new_frame_energy_th(const float *energies, int count, float frameEnergyTHN, float facMedian, float constWeight, float overall,
                                       float *th) {
  if (count < 0 || (count && !energies) || !th) return SOS_ERR_ARG;
  std::vector<float> v(energies, energies + count);
  *th = energy_threshold(v, frameEnergyTHN, facMedian, constWeight, overall);
  return SOS_OK;
}
extern "C" int sosf_solve_system(int n, const double *H_top, const double *b_top, const double *H_sc, const double *b_sc, const double *HM,
                                 const double *bM, const double *delta, double lambda, double *x) {
  if (n < 1 || !H_top || !b_top || !H_sc || !b_sc || !HM || !bM || !delta || !x) return SOS_ERR_ARG;
  const int dim = SOS_CPARS + 8 * n;
  const size_t dd = (size_t)dim * dim;
  MatXX H(H_top, H_top + dd), Hs(H_sc, H_sc + dd), M(HM, HM + dd);
  VecX b(b_top, b_top + dim), bs(b_sc, b_sc + dim), bm(bM, bM + dim), dl(delta, delta + dim), xs;
  double keep[2] = {g_phase[1], g_phase[2]};
  solve_visual_system(H, b, Hs, bs, M, bm, dl, lambda, xs, now_s());
  g_phase[1] = keep[0]; g_phase[2] = keep[1];  // (a test hook does not count as an iteration's phase)
  std::memcpy(x, xs.data(), sizeof(double) * dim);
  return SOS_OK;
}
extern "C" int sosf_marginalize_frame_prior(int n, int idx, const double *HM, const double *bM, const double *prior8, const double *delta_prior8,
                                            double *HM_out, double *bM_out) {
  if (n < 2 || idx < 0 || idx >= n || !HM || !bM || !prior8 || !delta_prior8 || !HM_out || !bM_out) return SOS_ERR_ARG;
  const int od = SOS_CPARS + 8 * n, nd = od - 8;
  MatXX H(HM, HM + (size_t)od * od), Ho;
  VecX b(bM, bM + od), bo;
  if (!marginalize_frame_prior(H, b, n, idx, prior8, delta_prior8, Ho, bo)) return SOS_ERR_STATE;
  std::memcpy(HM_out, Ho.data(), sizeof(double) * (size_t)nd * nd);
  std::memcpy(bM_out, bo.data(), sizeof(double) * nd);
  return SOS_OK;
}
extern "C" int sosf_ldlt_partial_solve(const double *A, const double *b, double *x, int n, int m) {
  if (!A || !b || !x || n <= 0 || m < 0 || m > n) return SOS_ERR_ARG;
  sos::LdltPartial F;
  F.n = n;
  F.m = m;
  F.U.assign(A, A + (size_t)n * n);
  sos::ldlt_partial_factor(F);
  std::vector<double> y(b, b + n);
  sos::ldlt_partial_forward(F, y.data());
  const int nb = n - m;
  if (nb > 0) {  // the trailing block: Schur complement as the factorisation left it (strict upper part in U, diagonal apart)
    std::vector<double> B((size_t)nb * nb, 0.0), rb(y.begin() + m, y.end()), xb;
    for (int j = 0; j < nb; j++) {
      B[(size_t)j * nb + j] = F.diag[m + j];
      for (int c = j + 1; c < nb; c++) B[(size_t)j * nb + c] = B[(size_t)c * nb + j] = F.U[(size_t)(m + j) * n + m + c];
    }
    ldlt_solve(B, rb, xb, nb);
    for (int j = 0; j < nb; j++) y[m + j] = xb[j];
  }
  sos::ldlt_partial_backward(F, y.data());
  std::memcpy(x, y.data(), sizeof(double) * n);
  return SOS_OK;
}
extern "C" int sosf_get_timing(double *phases8, int reset) {
  if (phases8) std::memcpy(phases8, sos::g_phase, sizeof(double) * 8);
  if (reset) std::memset(sos::g_phase, 0, sizeof(double) * 8);
  return SOS_OK;
}
extern "C" int sosf_set_hooks(sosf_system *s, sosf_allreduce_fn ar, sosf_nth_fn nth, void *user) {
  if (!s) return SOS_ERR_ARG;
  s->fs->ef->allreduceHook = ar;
  s->fs->ef->nthHook = nth;
  s->fs->ef->hookUser = user;
  if (!ar) s->fs->ef->allreduceF64Hook = nullptr;
  return SOS_OK;
}
extern "C" int sosf_set_allreduce_f64_hook(sosf_system *s, sosf_allreduce_f64_fn ar64) {
  if (!s) return SOS_ERR_ARG;
  s->fs->ef->allreduceF64Hook = ar64;
  return SOS_OK;
}
struct sosf_tracker {
  CoarseTracker *ct;
  FullSystem *fs;
};
extern "C" int sosf_upload_image(sosf_system *s, const float *image, int *slot_out) {
  if (!s || !image || !slot_out) return SOS_ERR_ARG;
  FullSystem *fs = s->fs;
  for (int k = 0; k < SOS_MAX_SLOTS; k++)
    if (!fs->slotUsed[k]) {
      int rc = sos_make_pyramid(fs->ctx, k, image, nullptr);
      if (rc) return rc;
      fs->slotUsed[k] = true;
      *slot_out = k;
      return SOS_OK;
    }
  return SOS_ERR_STATE;
}
extern "C" int sosf_alloc_slot(sosf_system *s, int *slot_out) {
  if (!s || !slot_out) return SOS_ERR_ARG;
  for (int k = 0; k < SOS_MAX_SLOTS; k++)
    if (!s->fs->slotUsed[k]) {
      s->fs->slotUsed[k] = true;
      *slot_out = k;
      return SOS_OK;
    }
  return SOS_ERR_STATE;
}
extern "C" int sosf_release_image(sosf_system *s, int slot) {
  if (!s || slot < 0 || slot >= SOS_MAX_SLOTS) return SOS_ERR_ARG;
  s->fs->slotUsed[slot] = false;
  return sos_frame_release(s->fs->ctx, slot);
}
extern "C" int sosf_tracker_create(sosf_system *s, sosf_tracker **out) {
  if (!s || !out) return SOS_ERR_ARG;
  CoarseTracker *ct = new CoarseTracker(s->fs->ctx, s->fs->prm);
  if (!ct->ok()) { delete ct; return SOS_ERR_HIP; }
  sosf_tracker *t = new sosf_tracker();
  t->ct = ct;
  t->fs = s->fs;
  *out = t;
  return SOS_OK;
}
extern "C" int sosf_tracker_destroy(sosf_tracker *t) {
  if (!t) return SOS_OK;
  delete t->ct;
  delete t;
  return SOS_OK;
}
extern "C" int sosf_tracker_set_ref(sosf_tracker *t, int32_t *pc_n_out) {
  if (!t) return SOS_ERR_ARG;
  t->ct->makeK(&t->fs->HCalib);
  int rc = t->ct->setCoarseTrackingRef(t->fs->frameHessians);
  if (pc_n_out) for (int l = 0; l < t->ct->levels; l++) pc_n_out[l] = t->ct->pc_n[l];
  return rc;
}
extern "C" int sosf_tracker_set_ref_raw(sosf_tracker *t, int npts, const float *u, const float *v, const float *idepth,
                                        const float *hdi, int32_t *pc_n_out) {
  if (!t) return SOS_ERR_ARG;
  t->ct->makeK(&t->fs->HCalib);
  int rc = t->ct->setCoarseTrackingRefRaw(t->fs->frameHessians.back(), npts, u, v, idepth, hdi);
  if (pc_n_out) for (int l = 0; l < t->ct->levels; l++) pc_n_out[l] = t->ct->pc_n[l];
  return rc;
}
extern "C" sos_tracker *sosf_tracker_handle(sosf_tracker *t) { return t ? t->ct->trk : nullptr; }
extern "C" int sosf_tracker_track(sosf_tracker *t, int newSlot, float new_ab_exposure, double *lastToNew12, double *aff2,
                                  int coarsestLvl, const double *minResForAbort5, double *lastResiduals5, double *flow3,
                                  int *ok) {
  if (!t || !lastToNew12 || !aff2 || !minResForAbort5 || !lastResiduals5) return SOS_ERR_ARG;
  SE3 T = SE3::from12(lastToNew12);
  AffLight aff(aff2[0], aff2[1]);
  const bool good = t->ct->trackNewestCoarse(newSlot, new_ab_exposure, T, aff, coarsestLvl, minResForAbort5, lastResiduals5);
  T.to12(lastToNew12);
  aff2[0] = aff.a;
  aff2[1] = aff.b;
  if (flow3) for (int i = 0; i < 3; i++) flow3[i] = t->ct->lastFlowIndicators[i];
  if (ok) *ok = good ? 1 : 0;
  return SOS_OK;
}
extern "C" int sosf_tracker_optimize_scale_kf(sosf_tracker *t, int stereoSlot, const double *tfmF0ToF1_12, const float *K1_level0,
                                              float trackingRefScale, int coarsestLvl, float thres, int32_t *state2, float *new_scale,
                                              float *scale_error) {
  if (!t || !tfmF0ToF1_12 || !K1_level0 || !state2 || !new_scale) return SOS_ERR_ARG;
  CoarseTracker::ScaleOptState st;
  st.scaleTrapped = state2[0];
  st.fails = state2[1];
  *new_scale = t->ct->optimizeScaleKF(stereoSlot, SE3::from12(tfmF0ToF1_12), K1_level0, trackingRefScale, coarsestLvl, thres, st, scale_error);
  state2[0] = st.scaleTrapped;
  state2[1] = st.fails;
  return SOS_OK;
}
extern "C" int sosf_tracker_set_device_lm(sosf_tracker *t, int on) {
  if (!t) return SOS_ERR_ARG;
  t->ct->deviceLM = on != 0;
  return SOS_OK;
}
extern "C" int sosf_tracker_last_evals(sosf_tracker *t, int *evals) {
  if (!t || !evals) return SOS_ERR_ARG;
  *evals = t->ct->lastEvals;
  return SOS_OK;
}
extern "C" int sosf_tracker_lm_profile(sosf_tracker *t, int hyp, double *us7) {
  if (!t || !us7) return SOS_ERR_ARG;
  return sos_tracker_lm_profile(t->ct->trk, hyp, us7);
}
extern "C" int sosf_tracker_set_lm_spin_limit(sosf_tracker *t, unsigned rounds) {
  if (!t) return SOS_ERR_ARG;
  return sos_tracker_set_lm_spin_limit(t->ct->trk, rounds);
}
extern "C" int sosf_tracker_lm_fallbacks(sosf_tracker *t, int *count) {
  if (!t || !count) return SOS_ERR_ARG;
  *count = t->ct->lmFallbacks;
  return SOS_OK;
}
extern "C" int sosf_tracker_make_tries(const double *slast_2_sprelast12, const double *lastF_2_slast12, const double *lastF_2_fh_imu12,
                                       int posesValid, int cap, double *tries12, int *n_out) {
  if (!slast_2_sprelast12 || !lastF_2_slast12 || !tries12 || !n_out) return SOS_ERR_ARG;
  std::vector<SE3> tries;
  SE3 imu;
  if (lastF_2_fh_imu12) imu = SE3::from12(lastF_2_fh_imu12);
  CoarseTracker::makeTrackTries(SE3::from12(slast_2_sprelast12), SE3::from12(lastF_2_slast12), lastF_2_fh_imu12 ? &imu : nullptr,
                                posesValid != 0, tries);
  *n_out = (int)tries.size();
  if ((int)tries.size() > cap) return SOS_ERR_ARG;
  for (size_t i = 0; i < tries.size(); i++) tries[i].to12(tries12 + 12 * i);
  return SOS_OK;
}
extern "C" int sosf_tracker_track_hypotheses(sosf_tracker *t, int newSlot, float new_ab_exposure, int nTries, const double *tries12,
                                             const double *aff_last2, int coarsestLvl, const double *lastCoarseRMSE5,
                                             double reTrackThreshold, int batch, double *lastF_2_fh12, double *aff2,
                                             double *achievedRes5, double *flow3, int *info4) {
  if (!t || !tries12 || nTries < 1 || !aff_last2 || !lastCoarseRMSE5 || !lastF_2_fh12 || !aff2 || !achievedRes5) return SOS_ERR_ARG;
  std::vector<SE3> tries(nTries);
  for (int i = 0; i < nTries; i++) tries[i] = SE3::from12(tries12 + 12 * i);
  CoarseTracker::TrackResult r;
  const int rc = t->ct->trackHypotheses(newSlot, new_ab_exposure, tries, AffLight(aff_last2[0], aff_last2[1]), coarsestLvl, lastCoarseRMSE5,
                                        reTrackThreshold, batch, r);
  if (rc != SOS_OK) return rc;
  r.lastF_2_fh.to12(lastF_2_fh12);
  aff2[0] = r.aff_g2l.a;
  aff2[1] = r.aff_g2l.b;
  for (int i = 0; i < 5; i++) achievedRes5[i] = r.achievedRes[i];
  if (flow3) for (int i = 0; i < 3; i++) flow3[i] = r.flowVecs[i];
  if (info4) { info4[0] = r.tryIterations; info4[1] = r.chosen; info4[2] = r.evaluated; info4[3] = r.haveOneGood ? 1 : 0; }
  return SOS_OK;
}
extern "C" int sosf_set_imu(sosf_system *sy, const sosf_imu_settings *S, sosf_imu_calib *C, sosf_imu_frame *frames, const double *HM,
                            const double *bM) {
  if (!sy) return SOS_ERR_ARG;
  EnergyFunctional *ef = sy->fs->ef;
  if (S && (!C || !frames || (HM == nullptr) != (bM == nullptr))) return SOS_ERR_ARG;
  ef->imuSettings = S; ef->imuCalib = C; ef->imuFrames = frames; ef->imuHM = HM; ef->imuBM = bM;
  ef->imuCallerPriorName++;  // a caller's prior is named by the call that handed it over (see the header: rewritten in place -> call again)
  if (!S) ef->imuMergedSamples.clear();
  if (S && !HM) {
    if (!ef->imuOwnPrior) ef->imuAdoptPrior();  // first call: the visual prior, expanded; later calls only renew the records
  } else {
    ef->imuOwnPrior = false;
  }
  return SOS_OK;
}
extern "C" int sosf_get_imu_prior(sosf_system *sy, double *HM, double *bM, int *dim) {
  if (!sy) return SOS_ERR_ARG;
  EnergyFunctional *ef = sy->fs->ef;
  if (!ef->imuOwnPrior) return SOS_ERR_STATE;
  const int nd = SOSF_IMU_DIM(ef->nFrames);
  if (dim) *dim = nd;
  if (HM) std::memcpy(HM, ef->HMi.data(), sizeof(double) * (size_t)nd * nd);
  if (bM) std::memcpy(bM, ef->bMi.data(), sizeof(double) * nd);
  return SOS_OK;
}
extern "C" int sosf_get_imu_step(sosf_system *sy, double *scale_step, double *step_imu) {
  if (!sy) return SOS_ERR_ARG;
  EnergyFunctional *ef = sy->fs->ef;
  if (scale_step) *scale_step = ef->imuScaleStep;
  if (step_imu && !ef->imuStep.empty()) std::memcpy(step_imu, ef->imuStep.data(), sizeof(double) * ef->imuStep.size());
  return SOS_OK;
}
extern "C" int sosf_write_poses(const char *path, int n, const int32_t *incoming_id, const double *t_wc) {
  if (!path || n < 0 || (n && (!incoming_id || !t_wc))) return SOS_ERR_ARG;
  FILE *f = fopen(path, "w");
  if (!f) return SOS_ERR_STATE;
  for (int i = 0; i < n; i++) fprintf(f, "%d %.6g %.6g %.6g\n", incoming_id[i], t_wc[3 * i], t_wc[3 * i + 1], t_wc[3 * i + 2]);
  fclose(f);
  return SOS_OK;
}
extern "C" int sosf_tracker_set_points3d(sosf_tracker *t, const sos_calib *cam, float matched_ab_exposure, int n, const float *xyz,
                                         const float *colors) {
  if (!t || !cam) return SOS_ERR_ARG;
  return t->ct->setPoints3d(*cam, matched_ab_exposure, n, xyz, colors);
}
extern "C" int sosf_tracker_pose_estimate(sosf_tracker *t, int newSlot, float new_ab_exposure, double *refToNew12, int coarsestLvl,
                                          float loopDirectThres, int innerPercent, float *poseError, int *inlierPercent, int *ok) {
  if (!t || !refToNew12) return SOS_ERR_ARG;
  SE3 T = SE3::from12(refToNew12);
  const bool good = t->ct->poseEstimate(newSlot, new_ab_exposure, T, coarsestLvl, loopDirectThres, innerPercent, poseError, inlierPercent);
  T.to12(refToNew12);
  if (ok) *ok = good ? 1 : 0;
  return SOS_OK;
}
extern "C" int sosf_tracker_optimize_scale(sosf_tracker *t, int stereoSlot, const double *tfmF0ToF1_12, const float *K1,
                                           float *scale_inout, int coarsestLvl, float *rmse) {
  if (!t || !tfmF0ToF1_12 || !K1 || !scale_inout) return SOS_ERR_ARG;
  const float r = t->ct->optimizeScale(stereoSlot, SE3::from12(tfmF0ToF1_12), K1, *scale_inout, coarsestLvl);
  if (rmse) *rmse = r;
  return SOS_OK;
}
extern "C" sos_ctx *sosf_ctx(sosf_system *s) { return s ? s->fs->ctx : nullptr; }
extern "C" sos_ba *sosf_ba(sosf_system *s) { return s ? s->fs->ef->ba : nullptr; }
extern "C" int sosf_frame_slot(sosf_system *s, int frameIdx) {
  if (!s || frameIdx < 0 || frameIdx >= (int)s->fs->frameHessians.size()) return -1;
  return s->fs->frameHessians[frameIdx]->slot;
}


// ================================================================================================
// Candidate selection of FullSystem::activatePointsMT (FS/FullSystem.cpp:375-470) around the device-side
// optimizeImmaturePoint; CoarseDistanceMap (FS/CoarseTracker.cpp:766-954) as a breadth-first distance transform
// ================================================================================================
namespace {
struct DistanceMap {
  int w1 = 0, h1 = 0, W = 0;
  std::vector<float> dist;
  std::vector<int> bfs1, bfs2;  // packed (x | y << 16)
  std::vector<uint64_t> visited, front, S, H, nxt;  // one bit per cell, W words per row (the seeding pass)
  void reset(int w, int h, int nSeedsMax) {  // buffers kept between calls (thread-local instance): only dist is re-initialised
    w1 = w; h1 = h; W = (w + 63) / 64;
    const size_t n = (size_t)w * h, q = std::max(n, (size_t)nSeedsMax + 1);  // (several active points may share a cell: the seed list can be longer than the image)
    dist.assign(n, 1000.f);
    if (bfs1.size() < q) { bfs1.resize(q); bfs2.resize(q); }
  }
  inline void relax(int idx, int x, int y, int k, int &num) {
    if (dist[idx] > k) {
      dist[idx] = (float)k;
      bfs1[num++] = x | (y << 16);
    }
  }
  void grow(int bfsNum) {  // growDistBFS, FS/CoarseTracker.cpp:828-917: 4-neighbours on even rounds, 8 on odd ones
    for (int k = 1; k < 40 && bfsNum > 0; k++) {  // (an empty front stays empty: the remaining rounds do nothing)
      const int bfsNum2 = bfsNum;
      std::swap(bfs1, bfs2);
      bfsNum = 0;
      for (int i = 0; i < bfsNum2; i++) {
        const int x = bfs2[i] & 0xffff, y = bfs2[i] >> 16;
        if (x == 0 || y == 0 || x == w1 - 1 || y == h1 - 1) continue;
        const int idx = x + y * w1;
        relax(idx + 1, x + 1, y, k, bfsNum);
        relax(idx - 1, x - 1, y, k, bfsNum);
        relax(idx + w1, x, y + 1, k, bfsNum);
        relax(idx - w1, x, y - 1, k, bfsNum);
        if (k & 1) {
          relax(idx + 1 + w1, x + 1, y + 1, k, bfsNum);
          relax(idx - 1 + w1, x - 1, y + 1, k, bfsNum);
          relax(idx - 1 - w1, x - 1, y - 1, k, bfsNum);
          relax(idx + 1 - w1, x + 1, y - 1, k, bfsNum);
        }
      }
    }
  }
  // The same rounds for the seeding pass (all active points at once, every other cell at 1000), on bitmaps.  With every cell either
  // unreached or assigned in an earlier round, round k of growDistBFS assigns k to exactly the unreached cells next to the previous
  // round's cells (4-neighbourhood for even k, 8 for odd k; cells on the image border are reached but do not spread): a dilation of
  // the front, 64 cells per operation, instead of one queue entry per cell and eight tests per entry.  dist[] comes out as grow() leaves it.
  void grow_seeds(int nSeeds) {
    const size_t nw = (size_t)W * h1;
    visited.assign(nw, 0);
    front.assign(nw, 0);
    S.assign(nw, 0);
    H.assign(nw, 0);
    nxt.assign(nw, 0);
    for (int i = 0; i < nSeeds; i++) {
      const int x = bfs1[i] & 0xffff, y = bfs1[i] >> 16;
      front[(size_t)y * W + (x >> 6)] |= 1ull << (x & 63);
    }
    visited = front;
    // column masks: cells that may spread (1 <= x <= w1 - 2) and cells that exist (x < w1)
    std::vector<uint64_t> inner(W, ~0ull), valid(W, ~0ull);
    if (w1 & 63) valid[W - 1] = (1ull << (w1 & 63)) - 1;
    for (int i = 0; i < W; i++) inner[i] = valid[i];
    inner[0] &= ~1ull;
    inner[(w1 - 1) >> 6] &= ~(1ull << ((w1 - 1) & 63));
    for (int k = 1; k < 40; k++) {
      for (int y = 1; y < h1 - 1; y++) {  // rows 0 and h1 - 1 do not spread: their S / H stay zero
        const uint64_t *f = &front[(size_t)y * W];
        uint64_t *s = &S[(size_t)y * W], *h = &H[(size_t)y * W];
        for (int i = 0; i < W; i++) s[i] = f[i] & inner[i];
        for (int i = 0; i < W; i++)
          h[i] = (s[i] << 1) | (i ? s[i - 1] >> 63 : 0) | (s[i] >> 1) | (i + 1 < W ? s[i + 1] << 63 : 0);
      }
      bool any = false;
      const bool eight = (k & 1) != 0;
      for (int y = 0; y < h1; y++) {
        const uint64_t *hy = &H[(size_t)y * W], *su = y > 0 ? &S[(size_t)(y - 1) * W] : nullptr, *sd = y + 1 < h1 ? &S[(size_t)(y + 1) * W] : nullptr;
        const uint64_t *hu = y > 0 ? &H[(size_t)(y - 1) * W] : nullptr, *hd = y + 1 < h1 ? &H[(size_t)(y + 1) * W] : nullptr;
        uint64_t *v = &visited[(size_t)y * W], *o = &nxt[(size_t)y * W];
        float *drow = &dist[(size_t)y * w1];
        for (int i = 0; i < W; i++) {
          uint64_t nb = hy[i];
          if (su) nb |= su[i] | (eight ? hu[i] : 0);
          if (sd) nb |= sd[i] | (eight ? hd[i] : 0);
          uint64_t fresh = nb & ~v[i] & valid[i];
          o[i] = fresh;
          v[i] |= fresh;
          any |= fresh != 0;
          while (fresh) {
            drow[(i << 6) + __builtin_ctzll(fresh)] = (float)k;
            fresh &= fresh - 1;
          }
        }
      }
      if (!any) break;
      front.swap(nxt);
    }
  }
  void add(int u, int v) {  // addIntoDistFinal, :919-925
    bfs1[0] = u | (v << 16);
    dist[u + w1 * v] = 0;
    grow(1);
  }
};
}  // namespace

extern "C" float sosf_next_min_act_dist(float d, int nPoints, float desired) {  // FS/FullSystem.cpp:377-399
  if (nPoints < desired * 0.66) d -= 0.8;
  if (nPoints < desired * 0.8) d -= 0.5;
  else if (nPoints < desired * 0.9) d -= 0.2;
  else if (nPoints < desired) d -= 0.1;
  if (nPoints > desired * 1.5) d += 0.8;
  if (nPoints > desired * 1.3) d += 0.5;
  if (nPoints > desired * 1.15) d += 0.2;
  if (nPoints > desired) d += 0.1;
  if (d < 0) d = 0;
  if (d > 4) d = 4;
  return d;
}

extern "C" int sosf_activate_select(int w1, int h1, int nFrames, int newest, const float *KRKi, const float *Kt, int nActive,
                                    const float *act_u, const float *act_v, const float *act_id, const int32_t *act_host,
                                    float currentMinActDist, float minTraceQuality, int nCand, const sos_immature *cand,
                                    const int32_t *cand_host, const float *cand_type, const uint8_t *hostFlagged,
                                    int8_t *decision, float *distFinal) {
  if (w1 < 3 || h1 < 3 || w1 > 65535 || h1 > 32767 || nFrames < 1 || newest < 0 || newest >= nFrames || !KRKi || !Kt ||
      nActive < 0 || nCand < 0 || (nActive && (!act_u || !act_v || !act_id || !act_host)) ||
      (nCand && (!cand || !cand_host || !cand_type || !hostFlagged || !decision)))
    return SOS_ERR_ARG;
  static thread_local DistanceMap dm;
  static const bool tmg = getenv("SOS_TIMING") != nullptr;
  const double tq0 = tmg ? now_s() : 0;
  dm.reset(w1, h1, nActive);
  // makeDistanceMap, FS/CoarseTracker.cpp:793-826
  int numItems = 0;
  for (int i = 0; i < nActive; i++) {
    const int f = act_host[i];
    if (f < 0 || f >= nFrames) return SOS_ERR_ARG;
    if (f == newest) continue;
    const float *K = KRKi + 9 * f, *T = Kt + 3 * f;
    const float p0 = K[0] * act_u[i] + K[1] * act_v[i] + K[2] + T[0] * act_id[i];
    const float p1 = K[3] * act_u[i] + K[4] * act_v[i] + K[5] + T[1] * act_id[i];
    const float p2 = K[6] * act_u[i] + K[7] * act_v[i] + K[8] + T[2] * act_id[i];
    const int u = (int)(p0 / p2 + 0.5f), v = (int)(p1 / p2 + 0.5f);
    if (!(u > 0 && v > 0 && u < w1 && v < h1)) continue;
    dm.dist[u + w1 * v] = 0;
    dm.bfs1[numItems++] = u | (v << 16);
  }
  // (bitmap dilations for the seeding pass; the per-cell queue of the single-seed passes, kept as a knob until round 6, takes 3x as long
  // on 4 k seeds with the same map: 0.67 against 2.34 ms at 376 x 240)
  dm.grow_seeds(numItems);
  const double tq1 = tmg ? now_s() : 0;
  // the candidate loop, FS/FullSystem.cpp:417-470
  for (int i = 0; i < nCand; i++) {
    const sos_immature &ph = cand[i];
    const int f = cand_host[i];
    if (f < 0 || f >= nFrames || f == newest) return SOS_ERR_ARG;
    if (!std::isfinite(ph.idepth_max) || ph.lastTraceStatus == SOS_IPS_OUTLIER) {  // :423-430
      decision[i] = SOSF_SEL_DELETE;
      continue;
    }
    const bool canActivate = (ph.lastTraceStatus == SOS_IPS_GOOD || ph.lastTraceStatus == SOS_IPS_SKIPPED ||
                              ph.lastTraceStatus == SOS_IPS_BADCONDITION || ph.lastTraceStatus == SOS_IPS_OOB) &&
                             ph.lastTracePixelInterval < 8 && ph.quality > minTraceQuality && (ph.idepth_max + ph.idepth_min) > 0;
    if (!canActivate) {  // :441-451
      decision[i] = (hostFlagged[f] || ph.lastTraceStatus == SOS_IPS_OOB) ? SOSF_SEL_DELETE : SOSF_SEL_KEEP;
      continue;
    }
    const float *K = KRKi + 9 * f, *T = Kt + 3 * f;
    const float idm = 0.5f * (ph.idepth_max + ph.idepth_min);
    const float p0 = K[0] * ph.u + K[1] * ph.v + K[2] + T[0] * idm;
    const float p1 = K[3] * ph.u + K[4] * ph.v + K[5] + T[1] * idm;
    const float p2 = K[6] * ph.u + K[7] * ph.v + K[8] + T[2] * idm;
    const int u = (int)(p0 / p2 + 0.5f), v = (int)(p1 / p2 + 0.5f);
    if (u > 0 && v > 0 && u < w1 && v < h1) {
      const float dist = dm.dist[u + w1 * v] + (p0 - floorf(p0));  // :461-462 (the fractional part of ptp[0], as written)
      if (dist >= currentMinActDist * cand_type[i]) {
        dm.add(u, v);
        decision[i] = SOSF_SEL_OPTIMIZE;
      } else {
        decision[i] = SOSF_SEL_KEEP;
      }
    } else {
      decision[i] = SOSF_SEL_DELETE;  // :468-471
    }
  }
  if (tmg) fprintf(stderr, "[activate_select] distance map of %d seeds %.0f us, %d candidates %.0f us\n", numItems, (tq1 - tq0) * 1e6, nCand, (now_s() - tq1) * 1e6);
  if (distFinal) memcpy(distFinal, dm.dist.data(), sizeof(float) * (size_t)w1 * h1);
  return SOS_OK;
}
