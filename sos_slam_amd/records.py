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
