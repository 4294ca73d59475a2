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


def test_device_loop_timeout_falls_back_to_host_loop():
    """A one-launch loop whose workgroups do not all become resident in time returns SOS_ERR_TIMEOUT (bounded waits, no hang); the
    facade then runs the same loop around the device passes.  Driven here with a spin limit of 0 (the first unmet wait gives up)."""
    from sos_slam_amd import host, lib
    from sos_slam_amd.synth import se3_exp12 as se3_exp, se3_mul12 as se3_mul
    L = lib.load()
    vp = C.c_void_p
    L.sos_tracker_track.argtypes = [vp, C.c_int, vp, C.c_float, C.c_float, vp, C.c_int, vp, C.c_int, vp]
    win = synth.make_window("W7", extra_frames=2)
    sysm = host.System.from_window(win)
    sysm.optimize(3)
    ht = host.HostTracker(sysm)
    ht.set_ref()
    slot = sysm.upload_image(win.extra_images[0])
    st_slot = sysm.upload_image(win.extra_images[1])
    levels = sysm.context().levels
    ref, new = win.frames[win.n - 1]["camToWorld"], win.extra_poses[0]
    Rr, tr, Rn, tn = ref[:9].reshape(3, 3), ref[9:], new[:9].reshape(3, 3), new[9:]
    T0 = se3_mul(se3_exp(np.array([0.004, -0.003, 0.002, 0.002, -0.002, 0.001])), np.concatenate([(Rn.T @ Rr).reshape(-1), Rn.T @ (tr - tn)]))
    # references: the host loop and the device loop with the default bound
    ht.set_device_lm(False)
    ok_h, Th, ah, lh, fh = ht.track(slot, 1.0, T0, np.zeros(2), levels - 1)
    ev_h = ht.last_evals()
    ht.set_device_lm(True)
    ok_d, Td, ad, ld, fd = ht.track(slot, 1.0, T0, np.zeros(2), levels - 1)
    assert ok_h and ok_d and ht.lm_fallbacks() == 0
    # the C-ABI itself: a multi-workgroup launch with limit 0 reports the timeout, and works again with the bound restored
    trk = C.c_void_p(sysm.L.sosf_tracker_handle(ht.h_))
    Ki = np.zeros((levels, 9), np.float32)
    K = sysm.calib_value_scaled()
    for l in range(levels):
        fx, fy = K[0] / 2 ** l, K[1] / 2 ** l
        cx, cy = (K[2] + 0.5) / 2 ** l - 0.5, (K[3] + 0.5) / 2 ** l - 0.5
        Ki[l] = np.array([1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1], np.float32)
    h = Hyp()
    h.refToNew[:] = list(T0)
    aff, mr = np.zeros(2), np.full(5, np.nan)
    p = lambda a: a.ctypes.data_as(vp)
    ht.set_lm_spin_limit(0)
    assert L.sos_tracker_track(trk, slot, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), 1, C.byref(h)) == -5     # SOS_ERR_TIMEOUT
    assert list(h.refToNew) == list(T0)                                                                       # nothing was written
    # the facade: same call, redone by the host loop -- the host loop's result exactly
    ok_f, Tf, af, lf, ff = ht.track(slot, 1.0, T0, np.zeros(2), levels - 1)
    assert ok_f and ht.lm_fallbacks() == 1 and ht.last_evals() == ev_h
    assert np.array_equal(Tf, Th) and np.array_equal(af, ah) and np.array_equal(lf[:levels], lh[:levels])
    # the scale loop falls back the same way
    tfm = np.concatenate([np.eye(3).reshape(-1), np.array([-0.11, 0.0, 0.0])])
    K1 = np.asarray(K, np.float32)
    ht.set_device_lm(False)
    s_h, e_h = ht.optimize_scale(st_slot, tfm, K1, 1.3, levels - 1)
    ht.set_device_lm(True)
    s_f, e_f = ht.optimize_scale(st_slot, tfm, K1, 1.3, levels - 1)
    assert ht.lm_fallbacks() == 2 and s_f == s_h and e_f == e_h
    ht.set_lm_spin_limit(1 << 18)
    ok_r, Tr, ar, lr_, fr = ht.track(slot, 1.0, T0, np.zeros(2), levels - 1)
    assert ok_r and ht.lm_fallbacks() == 2 and np.array_equal(Tr, Td)
    ht.close()
    sysm.close()


def test_more_hypotheses_than_one_launch_holds():
    """sos_tracker_track with 37 hypotheses in ONE call (three launches of at most 16 inside the library) and sos_tracker_optimize_scale with
    19 scales: every hypothesis comes back exactly as when it is run alone."""
    from sos_slam_amd import host, lib
    from sos_slam_amd.synth import se3_exp12 as se3_exp, se3_mul12 as se3_mul
    L = lib.load()
    vp = C.c_void_p
    L.sos_tracker_track.argtypes = [vp, C.c_int, vp, C.c_float, C.c_float, vp, C.c_int, vp, C.c_int, vp]
    L.sos_tracker_optimize_scale.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    win = synth.make_window("T6", extra_frames=2)
    sysm = host.System.from_window(win)
    sysm.optimize(3)
    ht = host.HostTracker(sysm)
    ht.set_ref()
    slot, st_slot = sysm.upload_image(win.extra_images[0]), sysm.upload_image(win.extra_images[1])
    levels = sysm.context().levels
    trk = C.c_void_p(sysm.L.sosf_tracker_handle(ht.h_))
    K = sysm.calib_value_scaled()
    Ki = np.zeros((levels, 9), np.float32)
    for l in range(levels):
        fx, fy = K[0] / 2 ** l, K[1] / 2 ** l
        cx, cy = (K[2] + 0.5) / 2 ** l - 0.5, (K[3] + 0.5) / 2 ** l - 0.5
        Ki[l] = np.array([1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1], np.float32)
    ref, new = win.frames[win.n - 1]["camToWorld"], win.extra_poses[0]
    Rr, tr, Rn, tn = ref[:9].reshape(3, 3), ref[9:], new[:9].reshape(3, 3), new[9:]
    T0 = np.concatenate([(Rn.T @ Rr).reshape(-1), Rn.T @ (tr - tn)])
    rng = np.random.default_rng(5)
    N = 37
    starts = [se3_mul(se3_exp(rng.normal(0, 0.004, 6)), T0) for _ in range(N)]
    aff, mr = np.zeros(2), np.full(5, np.nan)
    p = lambda a: a.ctypes.data_as(vp)
    hyps = (Hyp * N)()
    for k in range(N):
        hyps[k].refToNew[:] = list(starts[k])
    assert L.sos_tracker_track(trk, slot, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), N, hyps) == 0
    for k in (0, 7, 15, 16, 17, 31, 32, 36):
        one = Hyp()
        one.refToNew[:] = list(starts[k])
        assert L.sos_tracker_track(trk, slot, p(Ki), 1.0, 1.0, p(aff), levels - 1, p(mr), 1, C.byref(one)) == 0
        assert bytes(one) == bytes(hyps[k]), k
    assert len({bytes(hyps[k]) for k in range(N)}) > 1
    # the scale loop
    t3 = np.array([-0.11, 0.0, 0.0], np.float32)
    RK = np.ascontiguousarray(Ki.copy())           # rot(tfmF0ToF1) = I
    K1 = np.zeros((levels, 4), np.float32)
    for l in range(levels):
        K1[l] = (K[0] / 2 ** l, K[1] / 2 ** l, (K[2] + 0.5) / 2 ** l - 0.5, (K[3] + 0.5) / 2 ** l - 0.5)
    M = 19
    scales = np.linspace(0.5, 2.5, M).astype(np.float32)
    s_all, lr_all, ev = scales.copy(), np.zeros(5 * M), C.c_int(0)
    assert L.sos_tracker_optimize_scale(trk, st_slot, p(RK), p(t3), p(K1), levels - 1, M, p(s_all), p(lr_all), C.byref(ev)) == 0
    tot = 0
    for k in range(M):
        s1, lr1, e1 = scales[k:k + 1].copy(), np.zeros(5), C.c_int(0)
        assert L.sos_tracker_optimize_scale(trk, st_slot, p(RK), p(t3), p(K1), levels - 1, 1, p(s1), p(lr1), C.byref(e1)) == 0
        assert s1[0] == s_all[k] and np.array_equal(lr1, lr_all[5 * k:5 * k + 5], equal_nan=True), k
        tot += e1.value
    assert tot == ev.value
    ht.close()
    sysm.close()
