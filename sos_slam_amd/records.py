"""ctypes mirrors of the plain-data records of include/sos_slam.h (no native code is loaded here)."""
from __future__ import annotations

import ctypes as C

import numpy as np


class Params(C.Structure):
    """sos_params: the globals the path reads (util/settings.cpp:47-119)."""
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("huberTH", C.c_float),
                ("outlierTHSumComponent", C.c_float), ("affineOptModeA", C.c_float),
                ("affineOptModeB", C.c_float), ("idepthFixPrior", C.c_float),
                ("idepthFixPriorMargFac", C.c_float), ("margWeightFac", C.c_float),
                ("initialCalibHessian", C.c_float), ("coarseCutoffTH", C.c_float),
                ("frameEnergyTHN", C.c_float), ("frameEnergyTHFacMedian", C.c_float),
                ("frameEnergyTHConstWeight", C.c_float), ("overallEnergyTHWeight", C.c_float),
                ("reserved", C.c_float * 3)]

    @classmethod
    def from_dict(cls, d):
        p = cls()
        for k, v in d.items():
            setattr(p, k, v)
        return p


class Calib(C.Structure):
    """sos_calib: CalibHessian::value_scaledf / value_scaledi (FS/HessianBlocks.h:476-514)."""
    _fields_ = [(k, C.c_float) for k in ("fxl", "fyl", "cxl", "cyl", "fxli", "fyli", "cxli", "cyli")]

    @classmethod
    def from_K(cls, K):
        c = cls()
        f = np.asarray(K, dtype=np.float64).astype(np.float32)
        c.fxl, c.fyl, c.cxl, c.cyl = [float(x) for x in f]
        c.fxli = float(np.float32(1.0) / f[0])
        c.fyli = float(np.float32(1.0) / f[1])
        c.cxli = float(-f[2] / f[0])
        c.cyli = float(-f[3] / f[1])
        return c


class TraceParams(C.Structure):
    """sos_trace_params: the globals ImmaturePoint's constructor / traceOn read (util/settings.cpp:82-83,118,128-143)."""
    _fields_ = [(k, C.c_float) for k in ("maxPixSearch", "stepsize", "GNThreshold", "extraSlackOnTH", "slackInterval",
                                           "minImprovementFactor", "huberTH", "outlierTHSumComponent", "outlierTH",
                                           "overallEnergyTHWeight")] + [("GNIterations", C.c_int32),
                                                                        ("minTraceTestRadius", C.c_int32)]

    @classmethod
    def default(cls, **over):
        p = cls(0.027, 1.0, 0.1, 1.2, 1.5, 2.0, 9.0, 50.0 * 50.0, 12.0 * 12.0, 1.0, 3, 2)
        for k, v in over.items():
            setattr(p, k, v)
        return p


# numpy mirror of sos_immature (128 bytes)
IMMATURE_DTYPE = np.dtype([("u", "f4"), ("v", "f4"), ("idepth_min", "f4"), ("idepth_max", "f4"), ("color", "f4", (8,)),
                           ("weights", "f4", (8,)), ("gradH", "f4", (4,)), ("energyTH", "f4"), ("quality", "f4"),
                           ("lastTraceUV", "f4", (2,)), ("lastTracePixelInterval", "f4"), ("lastTraceStatus", "i4"),
                           ("pad", "i4", (2,))])
assert IMMATURE_DTYPE.itemsize == 128


class ActivateParams(C.Structure):
    """sos_activate_params: globals of FullSystem::optimizeImmaturePoint (util/settings.cpp:61,118,133)."""
    _fields_ = [("huberTH", C.c_float), ("minIdepthH_act", C.c_float), ("GNIts", C.c_int32), ("minObs", C.c_int32)]

    @classmethod
    def default(cls, **over):
        p = cls(9.0, 100.0, 3, 1)
        for k, v in over.items():
            setattr(p, k, v)
        return p


# numpy mirrors of sos_pair_tfm (64 bytes) and sos_activation (32 bytes)
PAIR_TFM_DTYPE = np.dtype([("R", "f4", (9,)), ("t", "f4", (3,)), ("aff", "f4", (2,)), ("pad", "f4", (2,))])
ACTIVATION_DTYPE = np.dtype([("status", "i4"), ("idepth", "f4"), ("inMask", "u4"), ("energy", "f4"), ("Hdd", "f4"),
                             ("bd", "f4"), ("iterations", "i4"), ("pad", "i4")])
assert PAIR_TFM_DTYPE.itemsize == 64 and ACTIVATION_DTYPE.itemsize == 32
ACT_SKIP, ACT_DELETE, ACT_ACTIVATED = 0, -1, 1


class PixselParams(C.Structure):
    """sos_pixsel_params: globals of PixelSelector (util/settings.cpp:122-125)."""
    _fields_ = [("minGradHistCut", C.c_float), ("minGradHistAdd", C.c_float), ("gradDownweightPerLevel", C.c_float),
                ("selectDirectionDistribution", C.c_int32)]

    @classmethod
    def default(cls, **over):
        p = cls(0.5, 7.0, 0.75, 1)
        for k, v in over.items():
            setattr(p, k, v)
        return p


def random_pattern(npx):
    """PixelSelector's randomPattern: glibc rand() & 0xFF after srand(3141592) (FS/PixelSelector2.cpp:37-40)."""
    libc = C.CDLL("libc.so.6")
    libc.srand(3141592)
    return np.fromiter((libc.rand() & 0xFF for _ in range(npx)), dtype=np.uint8, count=npx)


class CameraModel(C.Structure):
    """sos_camera_model: a parsed DSO camera file (U/Undistort.cpp:240-351, 679-800)."""
    _fields_ = [("model", C.c_int32), ("rect", C.c_int32), ("pars", C.c_double * 8), ("wOrg", C.c_int32), ("hOrg", C.c_int32),
                ("w", C.c_int32), ("h", C.c_int32), ("outCal", C.c_float * 5), ("pad", C.c_int32)]


CAM_RADTAN, CAM_PINHOLE, CAM_EQUIDISTANT, CAM_KB, CAM_FOV = range(5)
RECT_CROP, RECT_NONE, RECT_GIVEN = -1, -3, 0
assert C.sizeof(CameraModel) == 112


class ImuSettings(C.Structure):
    """sosf_imu_settings (util/settings.cpp:184-196, src/main.cpp:122-150)."""
    _fields_ = [("weight_imu", C.c_double * 36), ("weight_imu_bias", C.c_double * 36), ("gravity", C.c_double * 3),
                ("rot_imu_cam", C.c_double * 9), ("maxImuInterval", C.c_double), ("enable_scale_opt", C.c_int32), ("pad", C.c_int32)]


class ImuCalib(C.Structure):
    _fields_ = [("scale", C.c_double), ("scale_zero", C.c_double), ("scale_trapped", C.c_int32), ("imu_initialized", C.c_int32)]


class ImuFrame(C.Structure):
    """sosf_imu_frame; `imu` points at n_imu x 7 doubles kept alive by the caller."""
    _fields_ = [("timestamp", C.c_double), ("camToWorld", C.c_double * 12), ("evalPT_R", C.c_double * 9),
                ("state_imu", C.c_double * 21), ("state_imu_zero", C.c_double * 21), ("trackingRefIsPrev", C.c_int32),
                ("n_imu", C.c_int32), ("imu", C.c_void_p)]


class ImuShell(C.Structure):
    """sosf_imu_shell: the FrameShell fields the VIO front-end touches."""
    _fields_ = [("timestamp", C.c_double), ("camToWorld", C.c_double * 12), ("velInWorld", C.c_double * 3)]


def imu_dim(n):
    return 4 + 1 + 29 * n


# sos_resid_final (include/sos_slam.h): what linearizeAll(true) leaves in a PointFrameResidual, one record per residual
RESID_FINAL_DTYPE = np.dtype([("state_NewEnergy", "f4"), ("state_NewEnergyWithOutlier", "f4"), ("state_energy", "f4"),
                              ("centerProjectedTo", "f4", (3,)), ("state_NewState", "u1"), ("state_state", "u1"), ("active", "u1"),
                              ("pad", "u1")], align=True)


class SequenceParams(C.Structure):
    """sosf_sequence_params (include/sos_slam_host.h)"""
    _fields_ = [("trace", TraceParams), ("activate", ActivateParams), ("pixsel", PixselParams),
                ("desiredPointDensity", C.c_float), ("immatureDensity", C.c_float), ("minTraceQuality", C.c_float),
                ("kfEvery", C.c_int32), ("maxOptIterations", C.c_int32), ("patternPadding", C.c_int32),
                ("kfGlobalWeight", C.c_float), ("maxShiftWeightT", C.c_float), ("maxShiftWeightR", C.c_float),
                ("maxShiftWeightRT", C.c_float), ("maxAffineWeight", C.c_float)]

    @classmethod
    def default(cls, desired_points=2000.0, immature_density=1500.0, kf_every=0):
        p = cls()
        p.trace, p.activate, p.pixsel = TraceParams.default(), ActivateParams.default(), PixselParams.default()
        p.desiredPointDensity, p.immatureDensity, p.minTraceQuality = desired_points, immature_density, 3.0
        p.kfEvery, p.maxOptIterations, p.patternPadding = kf_every, 6, 2
        # util/settings.cpp:36-42
        p.kfGlobalWeight, p.maxShiftWeightT, p.maxShiftWeightR, p.maxShiftWeightRT, p.maxAffineWeight = 1.0, 0.04 * (640 + 480), 0.0, 0.02 * (640 + 480), 2.0
        return p


class FrameResult(C.Structure):
    """sosf_frame_result"""
    _fields_ = [("trackingOk", C.c_int32), ("isKeyframe", C.c_int32), ("refToNew", C.c_double * 12), ("camToWorld", C.c_double * 12),
                ("aff", C.c_double * 2), ("trackResiduals", C.c_double * 5), ("flow", C.c_double * 3),
                ("nActivated", C.c_int32), ("nDeletedImmature", C.c_int32), ("nPointsBeforeOpt", C.c_int32), ("iterations", C.c_int32),
                ("rmse", C.c_float), ("nOutliersRemoved", C.c_int32), ("nMargPoints", C.c_int32), ("nDroppedPoints", C.c_int32),
                ("nNewImmature", C.c_int32), ("nMargFrames", C.c_int32), ("margFrameIDs", C.c_int32 * 8), ("margCamToWorld", C.c_double * 96),
                ("newScale", C.c_float), ("scaleError", C.c_float)]


class FrameExtra(C.Structure):
    """sosf_frame_extra"""
    _fields_ = [("timestamp", C.c_double), ("n_imu", C.c_int32), ("stereoSlot", C.c_int32), ("imu", C.c_void_p)]
