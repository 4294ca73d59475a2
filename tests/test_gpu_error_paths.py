"""Error behaviour of the C-ABI entry points added for the device-resident tracker loops: every misuse comes back as a negative
status (SOS_ERR_ARG = -1, SOS_ERR_STATE = -3), nothing throws, nothing is launched, and the object stays usable."""
import ctypes as C

import numpy as np
import pytest

from sos_slam_amd import synth

pytestmark = pytest.mark.gpu

ERR_ARG, ERR_STATE = -1, -3


class Hyp(C.Structure):
    _fields_ = [("refToNew", C.c_double * 12), ("aff", C.c_double * 2), ("lastResiduals", C.c_double * 5), ("flow", C.c_double * 3),
                ("visit_res", C.c_double * 8), ("lastInners", C.c_int32 * 5), ("visit_lvl", C.c_int32 * 8), ("aborted", C.c_int32),
                ("nvisits", C.c_int32), ("evals", C.c_int32)]


def test_tracker_loop_entry_points_reject_misuse():
    from sos_slam_amd import host, lib
    L = lib.load()
    vp = C.c_void_p
    L.sos_tracker_track.argtypes = [vp, C.c_int, vp, C.c_float, C.c_float, vp, C.c_int, vp, C.c_int, vp]
    L.sos_tracker_optimize_scale.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    win = synth.make_window("T6", extra_frames=1)
    sysm = host.System.from_window(win)
    sysm.optimize(2)
    ht = host.HostTracker(sysm)
    trk = C.c_void_p(sysm.L.sosf_tracker_handle(ht.h_))
    slot = sysm.upload_image(win.extra_images[0])
    levels = sysm.context().levels
    Ki = np.tile(np.eye(3, dtype=np.float32).reshape(-1), levels)
    aff, mr = np.zeros(2), np.full(5, np.nan)
    h = Hyp()
    h.refToNew[:] = list(np.concatenate([np.eye(3).reshape(-1), np.zeros(3)]))
    p = lambda a: a.ctypes.data_as(vp)
    # no reference set yet
    assert L.sos_tracker_track(trk, slot, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), 1, C.byref(h)) == ERR_STATE
    ht.set_ref()
    assert L.sos_tracker_track(None, slot, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), 1, C.byref(h)) == ERR_ARG
    assert L.sos_tracker_track(trk, slot, None, 1.0, 1.0, p(aff), levels - 1, p(mr), 1, C.byref(h)) == ERR_ARG
    assert L.sos_tracker_track(trk, slot, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), 0, C.byref(h)) == ERR_ARG
    assert L.sos_tracker_track(trk, slot, p(Ki), 1.0, 1.0, p(aff), levels, p(mr), 1, C.byref(h)) == ERR_ARG          # level out of range
    assert L.sos_tracker_track(trk, 63, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), 1, C.byref(h)) == ERR_STATE       # empty slot
    assert L.sos_tracker_track(trk, -1, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), 1, C.byref(h)) == ERR_STATE
    sc = np.ones(1, np.float32)
    t3, K1 = np.zeros(3, np.float32), np.tile(np.array([100, 100, 50, 50], np.float32), levels)
    RK = np.tile(np.eye(3, dtype=np.float32).reshape(-1), levels)
    assert L.sos_tracker_optimize_scale(trk, slot, p(RK), p(t3), p(K1), levels - 1, 0, p(sc), None, None) == ERR_ARG
    assert L.sos_tracker_optimize_scale(trk, slot, p(RK), p(t3), None, levels - 1, 1, p(sc), None, None) == ERR_ARG
    assert L.sos_tracker_optimize_scale(trk, 63, p(RK), p(t3), p(K1), levels - 1, 1, p(sc), None, None) == ERR_STATE
    # and the object still works
    T0 = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
    ok, T, a, lr, fl = ht.track(slot, 1.0, T0, np.zeros(2), levels - 1)
    assert np.isfinite(T).all() and np.isfinite(lr[:levels]).all()
    # facade: the IMU prior getter without IMU mode, the hypothesis loop without tries
    assert sysm.L.sosf_get_imu_prior(sysm.h_, None, None, None) == ERR_STATE
    sysm.L.sosf_tracker_track_hypotheses.restype = C.c_int
    out = np.zeros(12), np.zeros(2), np.zeros(5), np.zeros(3)
    rc = sysm.L.sosf_tracker_track_hypotheses(ht.h_, slot, 1.0, 0, p(T0), p(aff), levels - 1, p(mr), 1.5, 16, p(out[0]), p(out[1]), p(out[2]),
                                             p(out[3]), None)
    assert rc == ERR_ARG
    ht.close()
    sysm.close()
