"""Shared helpers of the parity tests: build the oracle and the HIP backend on the same synthetic window."""
import numpy as np

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import Calib


def oracle_window(win):
    return orc.window_from_synth(win)


def gpu_backend(win, ow):
    """HIP backend holding `win` with the per-step state (precalc, adjoints, deltas) of the oracle host."""
    from sos_slam_amd import lib
    ctx = lib.Context(win.w, win.h)
    for i in range(win.n):
        ctx.make_pyramid(i, win.images[i])
    ba = lib.Backend(ctx, win.params)
    ba.set_window(np.arange(win.n), win.points, win.resid)
    push_state(ba, ow)
    return ctx, ba


def push_state(ba, ow):
    calib = Calib.from_K(ow.calib_value_scaled())
    pts = ow.pts()
    ba.set_state(calib=calib, precalc=ow.precalc().copy(), adHTdeltaF=ow.adHTdeltaF().copy(),
                 cDeltaF=np.zeros(4, np.float32), adHost=ow.adHost().copy(), adTarget=ow.adTarget().copy(),
                 idepth=pts["idepth_scaled"].copy(), idepth_zero=pts["idepth_zero_scaled"].copy(),
                 deltaF=pts["deltaF"].copy())


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def jac_equal(a, b):
    return all(np.array_equal(a[f], b[f]) for f in synth.RAWJAC_DTYPE.names)
