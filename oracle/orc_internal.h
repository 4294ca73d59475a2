/* oracle/orc_internal.h -- TEST INFRASTRUCTURE (CPU oracle internals, see oracle.h). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "oracle.h"
#include "orc_math.h"
#include "orc_pool.h"

#define ORC_RF_REMOVED 0x100u /* residual dropped by linearizeAll(true) (ef->dropResidual) */

typedef struct orc_hframe { /* host mirror of FrameHessian + EFFrame (FS/HessianBlocks.h:136-424) */
  orc_se3 camToWorld_evalPT;
  double state_zero[10], state[10], state_scaled[10], step[10], state_backup[10];
  orc_se3 PRE_camToWorld, PRE_worldToCam;
  float ab_exposure;
  int frameID;
  double prior[8], delta[8], delta_prior[8];
} orc_hframe;

struct orc_window {
  sos_params prm;
  sos_calib calib;
  int n, P, R;
  sos_point *pts;
  sos_resid *res;
  const float *img[SOS_MAX_FRAMES];
  int *pt_begin;
  sos_rawjac *J, *Jn;
  float *res_toZeroF, *JpJdF;
  int32_t *newState;
  float *newEnergy, *newEnergyWO, *center;
  double *retEnergy;
  float *Hdd_accAF, *bd_accAF, *Hcd_accAF, *Hdd_accLF, *bd_accLF, *Hcd_accLF;
  float *HdiF, *bdSumF, *idepth_hessian, *step, *maxRelBaseline, *idepth_backup;
  int32_t *numGoodResiduals;
  sos_precalc *precalc;
  float *adHTdeltaF;
  float cDeltaF[4];
  double *adHost, *adTarget;
  float *adHostF, *adTargetF;
  float frameEnergyTH[SOS_MAX_FRAMES];
  /* host level */
  orc_hframe hf[SOS_MAX_FRAMES];
  double c_value[4], c_value_zero[4], c_value_scaled[4], c_step[4], c_value_backup[4],
      c_value_minus_value_zero[4];
  double *HM, *bM, *lastX;
  /* IMU branch of solveSystemF (orc_host_set_imu): caller-owned records, the prior in the expanded dimension */
  const void *imuS; void *imuC; void *imuF; const double *imuHM, *imuBM;
  double imuScaleStep; double *imuStep;
  int resInA, resInL;
  int truth_mode; /* 1: the GN loop accumulates H/b in fp64 (not the reference's behaviour; error yardstick) */
};

void orc_apply_res_one(orc_window *W, int r);
double orc_linearize_one(orc_window *W, int r, const float *frameEnergyTH);
void orc_reset_oob_one(orc_window *W, int r);

#endif
