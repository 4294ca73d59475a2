"""ctypes binding of csrc/libsos_host.so: the C++ host facade (FullSystem / EnergyFunctional surface)
driving the HIP backend.  No CPU fallback: creation fails when the HIP library or the GPU is missing."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from . import lib as _lib
from .records import Params
from .synth import FRAME_INIT_DTYPE

_HOST = None


def load():
    global _HOST
    if _HOST is not None:
        return _HOST
    _lib.load()  # libsos_host.so links against libsos_slam_hip.so
    path = _build.HOST_LIB
    if not os.path.exists(path):
        raise _lib.SosError(f"{path} not built (run __graft_entry__.build())")
    L = C.CDLL(path)
    vp, ci = C.c_void_p, C.c_int
    L.sosf_create.argtypes = [C.POINTER(Params), ci, vp, C.POINTER(vp)]
    L.sosf_destroy.argtypes = [vp]
    L.sosf_set_calib.argtypes = [vp, vp]
    L.sosf_add_frame.argtypes = [vp, vp, vp]
    L.sosf_add_points.argtypes = [vp, ci, vp]
    L.sosf_add_residuals.argtypes = [vp, ci, vp]
    L.sosf_set_prior.argtypes = [vp, vp, vp]
    L.sosf_get_prior.argtypes = [vp, vp, vp]
    L.sosf_optimize.argtypes = [vp, ci, C.POINTER(C.c_float), C.POINTER(ci)]
    L.sosf_prepare.argtypes = [vp]
    L.sosf_gn_iteration.argtypes = [vp, ci, C.POINTER(ci)]
    L.sosf_set_pipeline.argtypes = [vp, ci]
    L.sosf_set_resident.argtypes = [vp, ci]
    L.sosf_set_comm.argtypes = [vp, vp]
    L.sosf_counts.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.sosf_get_frame.argtypes = [vp, ci, vp, vp, vp, C.POINTER(C.c_float)]
    L.sosf_get_calib.argtypes = [vp, vp]
    L.sosf_get_points.argtypes = [vp, vp, vp, vp, vp]
    L.sosf_get_point_ids.argtypes = [vp, vp]
    L.sosf_get_residuals.argtypes = [vp, vp, vp, vp]
    L.sosf_get_lastX.argtypes = [vp, vp]
    L.sosf_get_residual_ids.argtypes = [vp, vp, vp]
    L.sosf_add_frame_from_slot.argtypes = [vp, vp, ci]
    L.sosf_alloc_slot.argtypes = [vp, C.POINTER(ci)]
    L.sosf_release_image.argtypes = [vp, ci]
    L.sosf_flag_frames_for_marginalization.argtypes = [vp, vp, vp]
    L.sosf_add_new_frame_residuals.argtypes = [vp, C.POINTER(ci)]
    L.sosf_add_activated_points.argtypes = [vp, ci, vp, vp]
    L.sosf_remove_outliers.argtypes = [vp, C.POINTER(ci)]
    L.sosf_flag_points_for_removal.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    L.sosf_marginalize_flagged_frames.argtypes = [vp, ci, vp, vp, C.POINTER(ci)]
    L.sosf_get_point_keys.argtypes = [vp, vp, vp, vp, vp]
    L.sosf_get_frame_ids.argtypes = [vp, vp, vp, vp, vp, vp]
    L.sosf_keep_last_system.argtypes = [vp, ci]
    L.sosf_get_last_system.argtypes = [vp, vp, vp, vp, vp]
    L.sosf_set_force_accept_step.argtypes = [vp, ci]
    L.sosf_get_rejected_steps.argtypes = [vp, C.POINTER(ci)]
    L.sosf_get_stats.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.sosf_marginalize_points.argtypes = [vp, vp, ci]
    L.sosf_drop_points.argtypes = [vp, vp, ci]
    L.sosf_marginalize_frame.argtypes = [vp, ci]
    L.sosf_upload_image.argtypes = [vp, vp, C.POINTER(ci)]
    L.sosf_release_image.argtypes = [vp, ci]
    L.sosf_tracker_create.argtypes = [vp, C.POINTER(vp)]
    L.sosf_tracker_destroy.argtypes = [vp]
    L.sosf_tracker_set_ref.argtypes = [vp, vp]
    L.sosf_tracker_set_ref_raw.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    L.sosf_tracker_handle.restype = vp
    L.sosf_tracker_handle.argtypes = [vp]
    L.sosf_tracker_track.argtypes = [vp, ci, C.c_float, vp, vp, ci, vp, vp, vp, C.POINTER(ci)]
    L.sosf_write_poses.argtypes = [C.c_char_p, ci, vp, vp]
    L.sosf_tracker_set_device_lm.argtypes = [vp, ci]
    L.sosf_tracker_optimize_scale_kf.argtypes = [vp, ci, vp, vp, C.c_float, ci, C.c_float, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.sosf_tracker_last_evals.argtypes = [vp, C.POINTER(ci)]
    L.sosf_tracker_make_tries.argtypes = [vp, vp, vp, ci, ci, vp, C.POINTER(ci)]
    L.sosf_tracker_track_hypotheses.argtypes = [vp, ci, C.c_float, ci, vp, vp, ci, vp, C.c_double, ci, vp, vp, vp, vp, vp]
    L.sosf_set_imu.argtypes = [vp, vp, vp, vp, vp, vp]
    L.sosf_get_imu_prior.argtypes = [vp, vp, vp, C.POINTER(ci)]
    L.sosf_get_imu_step.argtypes = [vp, vp, vp]
    L.sosf_tracker_set_points3d.argtypes = [vp, vp, C.c_float, ci, vp, vp]
    L.sosf_tracker_pose_estimate.argtypes = [vp, ci, C.c_float, vp, ci, C.c_float, ci, vp, vp, vp]
    L.sosf_tracker_optimize_scale.argtypes = [vp, ci, vp, vp, C.POINTER(C.c_float), ci, C.POINTER(C.c_float)]
    L.sosf_get_timing.argtypes = [vp, ci]
    L.sosf_ldlt_solve.argtypes = [vp, vp, vp, ci, ci]
    L.sosf_activate_select.argtypes = [ci] * 4 + [vp, vp, ci, vp, vp, vp, vp, C.c_float, C.c_float, ci, vp, vp, vp, vp, vp, vp]
    L.sosf_next_min_act_dist.argtypes = [C.c_float, ci, C.c_float]
    L.sosf_next_min_act_dist.restype = C.c_float
    L.sosf_ctx.restype = vp
    L.sosf_ctx.argtypes = [vp]
    L.sosf_ba.restype = vp
    L.sosf_ba.argtypes = [vp]
    L.sosf_frame_slot.argtypes = [vp, ci]
    _HOST = L
    return L


_p = _lib._p
_chk = _lib._chk


def ldlt_solve(A, b, which=0):
    """The facade's pivoted LDL^T solve (which=0 blocked production variant, 1 unblocked reference)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(len(b))
    _chk(load().sosf_ldlt_solve(_p(A), _p(b), _p(x), len(b), which), "sosf_ldlt_solve")
    return x


def ldlt_partial_solve(A, b, m):
    """The same solve split as the cached visual-inertial solve splits it: leading m unknowns eliminated, trailing block on its own."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(len(b))
    _chk(load().sosf_ldlt_partial_solve(_p(A), _p(b), _p(x), len(b), m), "sosf_ldlt_partial_solve")
    return x


def solve_system(H_top, b_top, H_sc, b_sc, HM, bM, delta, lam=1e-5):
    """The facade's visual solve from its pieces (sosf_solve_system): upper triangles of H_top / H_sc / HM, all of HM for bM + HM delta."""
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (H_top, b_top, H_sc, b_sc, HM, bM, delta)]
    n = (len(a[1]) - 4) // 8
    x = np.zeros(len(a[1]))
    _chk(load().sosf_solve_system(n, *[_p(v) for v in a], C.c_double(float(lam)), _p(x)), "sosf_solve_system")
    return x


def marginalize_frame_prior(HM, bM, idx, prior8, delta_prior8):
    """The facade's prior algebra of marginalizeFrame (visual form) on a prior of dimension 4 + 8 n; returns (HM', bM') of 4 + 8 (n - 1)."""
    HM = np.ascontiguousarray(HM, dtype=np.float64)
    bM = np.ascontiguousarray(bM, dtype=np.float64)
    n = (len(bM) - 4) // 8
    p8, d8 = np.ascontiguousarray(prior8, dtype=np.float64), np.ascontiguousarray(delta_prior8, dtype=np.float64)
    nd = len(bM) - 8
    Ho, bo = np.zeros((nd, nd)), np.zeros(nd)
    _chk(load().sosf_marginalize_frame_prior(n, idx, _p(HM), _p(bM), _p(p8), _p(d8), _p(Ho), _p(bo)), "sosf_marginalize_frame_prior")
    return Ho, bo


def activate_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, host_flagged):
    """Candidate loop of FullSystem::activatePointsMT (sosf_activate_select); `act` = structured array with u, v,
    idepth_scaled, host of the active points.  Returns (decision int8[nCand], fwdWarpedIDDistFinal (h1, w1))."""
    KRKi = np.ascontiguousarray(KRKi, dtype=np.float32).reshape(-1, 9)
    Kt = np.ascontiguousarray(Kt, dtype=np.float32).reshape(-1, 3)
    au, av, aid = [np.ascontiguousarray(act[k], dtype=np.float32) for k in ("u", "v", "idepth_scaled")]
    ah = np.ascontiguousarray(act["host"], dtype=np.int32)
    cand = np.ascontiguousarray(cand)
    ch = np.ascontiguousarray(cand_host, dtype=np.int32)
    ct = np.ascontiguousarray(cand_type, dtype=np.float32)
    hf = np.ascontiguousarray(host_flagged, dtype=np.uint8)
    dec = np.zeros(len(cand), dtype=np.int8)
    dist = np.zeros((h1, w1), dtype=np.float32)
    _chk(load().sosf_activate_select(w1, h1, len(KRKi), newest, _p(KRKi), _p(Kt), len(au), _p(au), _p(av), _p(aid), _p(ah),
                                     float(min_dist), float(min_quality), len(cand), _p(cand), _p(ch), _p(ct), _p(hf), _p(dec),
                                     _p(dist)), "sosf_activate_select")
    return dec, dist


def next_min_act_dist(cur, n_points, desired):
    return float(load().sosf_next_min_act_dist(cur, n_points, desired))


def write_poses(path, incoming_id, t_wc):
    """LoopHandler::savePose: "incoming_id tx ty tz" per keyframe, 6 significant digits."""
    ids = np.ascontiguousarray(incoming_id, dtype=np.int32)
    t = np.ascontiguousarray(t_wc, dtype=np.float64).reshape(-1, 3)
    _chk(load().sosf_write_poses(str(path).encode(), len(ids), _p(ids), _p(t)), "sosf_write_poses")


def timing(reset=False):
    """Accumulated seconds per host phase of sosf_gn_iteration (see include/sos_slam_host.h)."""
    ph = np.zeros(8)
    load().sosf_get_timing(_p(ph), int(reset))
    return ph


class System:
    """FullSystem (backend-facing subset): frames, points, residuals, optimize(), marginalisation."""

    def __init__(self, params: dict, device: int = 0, stream: int | None = None):
        self.L = load()
        self.params = Params.from_dict(params)
        self.h_ = C.c_void_p()
        _chk(self.L.sosf_create(C.byref(self.params), device, C.c_void_p(stream) if stream else None,
                                C.byref(self.h_)), "sosf_create (is an MI355X visible?)")

    @classmethod
    def from_window(cls, win, device: int = 0):
        s = cls(win.params, device)
        s.set_calib(win.K)
        for i in range(win.n):
            s.add_frame(win.frames[i], win.images[i])
        s.add_points(win.points)
        s.add_residuals(win.resid)
        s.set_prior(win.HM, win.bM)
        return s

    def close(self):
        for t in list(getattr(self, "_trackers", [])):  # trackers borrow the system's context: they go first
            t.close()
        if getattr(self, "_ctx", None) is not None:   # ... and so do the front-end objects created on context()
            self._ctx.close()
            self._ctx = None
        if getattr(self, "h_", None):
            self.L.sosf_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_calib(self, K):
        k = np.ascontiguousarray(K, dtype=np.float64)
        _chk(self.L.sosf_set_calib(self.h_, _p(k)), "sosf_set_calib")

    def add_frame(self, frame_init, image):
        f = np.ascontiguousarray(np.asarray(frame_init, dtype=FRAME_INIT_DTYPE).reshape(1))
        img = np.ascontiguousarray(image, dtype=np.float32)
        _chk(self.L.sosf_add_frame(self.h_, _p(f), _p(img)), "sosf_add_frame")

    def add_points(self, pts):
        pts = np.ascontiguousarray(pts)
        _chk(self.L.sosf_add_points(self.h_, len(pts), _p(pts)), "sosf_add_points")

    def add_residuals(self, res):
        res = np.ascontiguousarray(res)
        _chk(self.L.sosf_add_residuals(self.h_, len(res), _p(res)), "sosf_add_residuals")

    def set_prior(self, HM, bM):
        _chk(self.L.sosf_set_prior(self.h_, _p(np.ascontiguousarray(HM, dtype=np.float64)),
                                   _p(np.ascontiguousarray(bM, dtype=np.float64))), "sosf_set_prior")

    def counts(self):
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        _chk(self.L.sosf_counts(self.h_, C.byref(a), C.byref(b), C.byref(c)), "sosf_counts")
        return a.value, b.value, c.value

    def get_prior(self):
        n = self.counts()[0]
        dim = 4 + 8 * n
        HM, bM = np.zeros((dim, dim)), np.zeros(dim)
        _chk(self.L.sosf_get_prior(self.h_, _p(HM), _p(bM)), "sosf_get_prior")
        return HM, bM

    def optimize(self, iters=6):
        r, it = C.c_float(0), C.c_int(0)
        _chk(self.L.sosf_optimize(self.h_, iters, C.byref(r), C.byref(it)), "sosf_optimize")
        return r.value, it.value

    def prepare(self):
        _chk(self.L.sosf_prepare(self.h_), "sosf_prepare")

    def set_imu(self, S=None, calib=None, frames=None, HM=None, bM=None):
        """IMU branch of solveSystemF on / off; the records stay owned (and kept alive) by this object."""
        if S is None:
            self._imu = None
            _chk(self.L.sosf_set_imu(self.h_, None, None, None, None, None), "sosf_set_imu")
            return
        from .records import ImuFrame
        arr = (ImuFrame * len(frames))(*frames)
        if HM is None:   # the facade keeps the expanded prior (setting_enable_imu semantics)
            self._imu = (S, calib, arr, frames, None, None)
            _chk(self.L.sosf_set_imu(self.h_, C.byref(S), C.byref(calib), arr, None, None), "sosf_set_imu")
            return
        HM = np.ascontiguousarray(HM, dtype=np.float64)
        bM = np.ascontiguousarray(bM, dtype=np.float64)
        self._imu = (S, calib, arr, frames, HM, bM)
        _chk(self.L.sosf_set_imu(self.h_, C.byref(S), C.byref(calib), arr, _p(HM), _p(bM)), "sosf_set_imu")

    def imu_prior(self):
        """the expanded prior kept by the facade (sosf_get_imu_prior): (HM (d, d), bM (d))"""
        d = C.c_int(0)
        _chk(self.L.sosf_get_imu_prior(self.h_, None, None, C.byref(d)), "sosf_get_imu_prior")
        HM, bM = np.zeros((d.value, d.value)), np.zeros(d.value)
        _chk(self.L.sosf_get_imu_prior(self.h_, _p(HM), _p(bM), None), "sosf_get_imu_prior")
        return HM, bM

    def imu_state(self):
        """(scale_step, step_imu (n, 21), state_imu (n, 21), scale) after the last solve"""
        S, calib, arr, frames, HM, bM = self._imu
        n = len(arr)
        ss, st = C.c_double(0), np.zeros((n, 21))
        _chk(self.L.sosf_get_imu_step(self.h_, C.byref(ss), _p(st)), "sosf_get_imu_step")
        return ss.value, st, np.array([list(arr[i].state_imu) for i in range(n)]), calib.scale

    def set_device_step(self, on=True):
        """device-side step of the GN loop (poses / precalc / deltas formed on the device from x) on / off"""
        self.L.sosf_set_device_step.argtypes = [C.c_void_p, C.c_int]
        _chk(self.L.sosf_set_device_step(self.h_, 1 if on else 0), "sosf_set_device_step")

    def loop_mode(self) -> int:
        """0 host step, 1 device-side step, 2 device-resident loop, 3 energy-checked step, -1 before the first iteration"""
        m = C.c_int(-1)
        _chk(self.L.sosf_get_loop_mode(self.h_, C.byref(m)), "sosf_get_loop_mode")
        return int(m.value)

    def set_resident(self, on=True):
        """device-resident Gauss-Newton loop on / off (off: the host solves, as in round 1)"""
        _chk(self.L.sosf_set_resident(self.h_, int(on)), "sosf_set_resident")

    def invalidate_pack(self):
        """the next prepare() / optimize() packs and uploads the window again (per-keyframe cost measurements)"""
        _chk(self.L.sosf_invalidate_pack(self.h_), "sosf_invalidate_pack")

    def set_host_threads(self, n: int):
        """threads of the facade's per-keyframe graph walks (process-wide; results do not depend on it)"""
        self.L.sosf_set_host_threads.argtypes = [C.c_int]
        _chk(self.L.sosf_set_host_threads(int(n)), "sosf_set_host_threads")

    def host_threads(self) -> int:
        return int(self.L.sosf_get_host_threads())

    def set_min_opt_iterations(self, its):
        _chk(self.L.sosf_set_min_opt_iterations(self.h_, int(its)), "sosf_set_min_opt_iterations")

    def set_pipeline(self, on=True):
        _chk(self.L.sosf_set_pipeline(self.h_, int(on)), "sosf_set_pipeline")

    def gn_iteration(self, iteration=0):
        cb = C.c_int(0)
        _chk(self.L.sosf_gn_iteration(self.h_, iteration, C.byref(cb)), "sosf_gn_iteration")
        return bool(cb.value)

    def frame(self, idx):
        c2w, st, sz = np.zeros(12), np.zeros(10), np.zeros(10)
        th = C.c_float(0)
        _chk(self.L.sosf_get_frame(self.h_, idx, _p(c2w), _p(st), _p(sz), C.byref(th)), "sosf_get_frame")
        return dict(camToWorld=c2w, state=st, state_zero=sz, frameEnergyTH=th.value)

    def calib_value_scaled(self):
        v = np.zeros(4)
        _chk(self.L.sosf_get_calib(self.h_, _p(v)), "sosf_get_calib")
        return v

    def points(self):
        P = self.counts()[1]
        idp, idh, mrb = [np.zeros(P, dtype=np.float32) for _ in range(3)]
        ngr = np.zeros(P, dtype=np.int32)
        _chk(self.L.sosf_get_points(self.h_, _p(idp), _p(idh), _p(mrb), _p(ngr)), "sosf_get_points")
        return dict(idepth=idp, idepth_hessian=idh, maxRelBaseline=mrb, numGoodResiduals=ngr)

    def residuals(self):
        R = self.counts()[2]
        st, act, rem = [np.zeros(R, dtype=np.int32) for _ in range(3)]
        _chk(self.L.sosf_get_residuals(self.h_, _p(st), _p(act), _p(rem)), "sosf_get_residuals")
        return dict(state_state=st, isActive=act)

    # ---- keyframe-rate host logic of makeKeyFrame (FS/FullSystem.cpp:783-931)
    def add_frame_from_slot(self, frame_init, slot):
        f = np.ascontiguousarray(np.asarray(frame_init, dtype=FRAME_INIT_DTYPE).reshape(1))
        _chk(self.L.sosf_add_frame_from_slot(self.h_, _p(f), int(slot)), "sosf_add_frame_from_slot")

    def flag_frames_for_marginalization(self, num_immature):
        n = self.counts()[0]
        ni = np.ascontiguousarray(num_immature, dtype=np.int32)
        assert len(ni) == n
        fl = np.zeros(n, dtype=np.uint8)
        _chk(self.L.sosf_flag_frames_for_marginalization(self.h_, _p(ni), _p(fl)), "sosf_flag_frames_for_marginalization")
        return fl.astype(bool)

    def add_new_frame_residuals(self):
        c = C.c_int(0)
        _chk(self.L.sosf_add_new_frame_residuals(self.h_, C.byref(c)), "sosf_add_new_frame_residuals")
        return c.value

    def add_activated_points(self, pts, in_mask):
        pts = np.ascontiguousarray(pts)
        m = np.ascontiguousarray(in_mask, dtype=np.uint32)
        _chk(self.L.sosf_add_activated_points(self.h_, len(pts), _p(pts), _p(m)), "sosf_add_activated_points")

    def remove_outliers(self):
        c = C.c_int(0)
        _chk(self.L.sosf_remove_outliers(self.h_, C.byref(c)), "sosf_remove_outliers")
        return c.value

    def flag_points_for_removal(self):
        a, b = C.c_int(0), C.c_int(0)
        _chk(self.L.sosf_flag_points_for_removal(self.h_, C.byref(a), C.byref(b)), "sosf_flag_points_for_removal")
        return a.value, b.value

    def marginalize_flagged_frames(self, cap=8):
        ids, poses, c = np.zeros(cap, np.int32), np.zeros((cap, 12)), C.c_int(0)
        _chk(self.L.sosf_marginalize_flagged_frames(self.h_, cap, _p(ids), _p(poses), C.byref(c)), "sosf_marginalize_flagged_frames")
        return ids[:c.value].copy(), poses[:c.value].copy()

    def point_keys(self):
        """(host frameID, u, v, host idx) per point in allPoints order"""
        P = self.counts()[1]
        hf, hi = np.zeros(P, np.int32), np.zeros(P, np.int32)
        u, v = np.zeros(P, np.float32), np.zeros(P, np.float32)
        _chk(self.L.sosf_get_point_keys(self.h_, _p(hf), _p(u), _p(v), _p(hi)), "sosf_get_point_keys")
        return hf, u, v, hi

    def frame_ids(self):
        n = self.counts()[0]
        fid, npt, nm, no = [np.zeros(n, np.int32) for _ in range(4)]
        fl = np.zeros(n, np.uint8)
        _chk(self.L.sosf_get_frame_ids(self.h_, _p(fid), _p(fl), _p(npt), _p(nm), _p(no)), "sosf_get_frame_ids")
        return dict(frameID=fid, flagged=fl.astype(bool), nPoints=npt, nMarg=nm, nOut=no)

    def keep_last_system(self, on=True):
        _chk(self.L.sosf_keep_last_system(self.h_, int(on)), "sosf_keep_last_system")

    def last_system(self):
        """(H_top, b_top, H_sc, b_sc) the last solveSystemF assembled (priors of the L stitch included)"""
        dim = 4 + 8 * self.counts()[0]
        H, Hsc, b, bsc = np.zeros((dim, dim)), np.zeros((dim, dim)), np.zeros(dim), np.zeros(dim)
        _chk(self.L.sosf_get_last_system(self.h_, _p(H), _p(b), _p(Hsc), _p(bsc)), "sosf_get_last_system")
        return H, b, Hsc, bsc

    def residual_ids(self):
        """(point add-index, target frameID) per residual of the current graph, order of residuals()"""
        R = self.counts()[2]
        pi, tf = np.zeros(R, dtype=np.int32), np.zeros(R, dtype=np.int32)
        _chk(self.L.sosf_get_residual_ids(self.h_, _p(pi), _p(tf)), "sosf_get_residual_ids")
        return pi, tf

    def set_force_accept_step(self, on=True):
        _chk(self.L.sosf_set_force_accept_step(self.h_, int(on)), "sosf_set_force_accept_step")

    def rejected_steps(self):
        c = C.c_int(0)
        _chk(self.L.sosf_get_rejected_steps(self.h_, C.byref(c)), "sosf_get_rejected_steps")
        return c.value

    def lastX(self):
        n = self.counts()[0]
        x = np.zeros(4 + 8 * n)
        _chk(self.L.sosf_get_lastX(self.h_, _p(x)), "sosf_get_lastX")
        return x

    def stats(self):
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        _chk(self.L.sosf_get_stats(self.h_, C.byref(a), C.byref(b), C.byref(c)), "sosf_get_stats")
        return dict(resInA=a.value, resInL=b.value, resInM=c.value)

    def point_ids(self):
        _, P, _ = self.counts()
        ids = np.zeros(P, dtype=np.int32)
        _chk(self.L.sosf_get_point_ids(self.h_, _p(ids)), "sosf_get_point_ids")
        return ids

    def marginalize_points(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        _chk(self.L.sosf_marginalize_points(self.h_, _p(idx), len(idx)), "sosf_marginalize_points")

    def drop_points(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        _chk(self.L.sosf_drop_points(self.h_, _p(idx), len(idx)), "sosf_drop_points")

    def marginalize_frame(self, frame_idx):
        _chk(self.L.sosf_marginalize_frame(self.h_, frame_idx), "sosf_marginalize_frame")


    def upload_image(self, image) -> int:
        img = np.ascontiguousarray(image, dtype=np.float32)
        slot = C.c_int(-1)
        _chk(self.L.sosf_upload_image(self.h_, _p(img), C.byref(slot)), "sosf_upload_image")
        return slot.value

    def alloc_slot(self) -> int:
        slot = C.c_int(-1)
        _chk(self.L.sosf_alloc_slot(self.h_, C.byref(slot)), "sosf_alloc_slot")
        return slot.value

    def release_image(self, slot):
        _chk(self.L.sosf_release_image(self.h_, int(slot)), "sosf_release_image")

    def context(self):
        """The system's own sos_ctx as a non-owning lib.Context (front-end objects -- undistorter, pixel selector, immature
        kernels -- work on the same frame store)."""
        if getattr(self, "_ctx", None) is None:
            self._ctx = _lib.BorrowedContext(self.L.sosf_ctx(self.h_), self.params.w, self.params.h)
        return self._ctx

    def frame_slot(self, idx) -> int:
        return self.L.sosf_frame_slot(self.h_, idx)


class HostTracker:
    """CoarseTracker / ScaleOptimizer of the C++ facade (host LM loops + device primitives)."""

    def __init__(self, sysm: System):
        self.L = sysm.L
        self.sys = sysm
        self.h_ = C.c_void_p()
        _chk(self.L.sosf_tracker_create(sysm.h_, C.byref(self.h_)), "sosf_tracker_create")
        self.pc_n = np.zeros(6, dtype=np.int32)
        if not hasattr(sysm, "_trackers"):
            sysm._trackers = []
        sysm._trackers.append(self)

    def close(self):
        if getattr(self, "h_", None):
            if getattr(self.sys, "h_", None):  # after the system is gone its context is gone too
                self.L.sosf_tracker_destroy(self.h_)
            self.h_ = None
        if self in getattr(self.sys, "_trackers", []):
            self.sys._trackers.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_ref(self):
        _chk(self.L.sosf_tracker_set_ref(self.h_, _p(self.pc_n)), "sosf_tracker_set_ref")
        return self.pc_n.copy()

    def set_ref_raw(self, u, v, idepth, hdi):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (u, v, idepth, hdi)]
        _chk(self.L.sosf_tracker_set_ref_raw(self.h_, len(a[0]), *[_p(x) for x in a], _p(self.pc_n)),
             "sosf_tracker_set_ref_raw")
        return self.pc_n.copy()

    def device(self):
        """The underlying sos_tracker as a lib.Tracker-like object (primitives calc_res / calc_gs ...)."""
        t = _lib.Tracker.__new__(_lib.Tracker)
        t.L = _lib.load()
        t.ctx = None
        t.h_ = None  # not owned
        t._borrowed = C.c_void_p(self.L.sosf_tracker_handle(self.h_))
        t.pc_n = self.pc_n
        return t

    def track(self, newSlot, new_ab_exposure, lastToNew12, aff2, coarsest, minRes=None):
        T = np.ascontiguousarray(lastToNew12, dtype=np.float64).copy()
        aff = np.ascontiguousarray(aff2, dtype=np.float64).copy()
        mr = np.full(5, np.nan) if minRes is None else np.ascontiguousarray(minRes, dtype=np.float64)
        lr, fl = np.zeros(5), np.zeros(3)
        ok = C.c_int(0)
        _chk(self.L.sosf_tracker_track(self.h_, newSlot, new_ab_exposure, _p(T), _p(aff), coarsest, _p(mr), _p(lr), _p(fl),
                                       C.byref(ok)), "sosf_tracker_track")
        return bool(ok.value), T, aff, lr, fl

    def optimize_scale_kf(self, stereoSlot, tfm12, K1, tracking_ref_scale, coarsest, thres, state):
        """FullSystem::optimizeScale (FS/FullSystem.cpp:1117-1177); state = [scaleTrapped, scale_opt_fails], updated in place.
        Returns (new_scale or -1, scale_error)."""
        tf = np.ascontiguousarray(tfm12, dtype=np.float64)
        k1 = np.ascontiguousarray(K1, dtype=np.float32)
        st = np.ascontiguousarray(state, dtype=np.int32)
        ns, er = C.c_float(0), C.c_float(0)
        _chk(self.L.sosf_tracker_optimize_scale_kf(self.h_, stereoSlot, _p(tf), _p(k1), float(tracking_ref_scale), coarsest, float(thres), _p(st),
                                                   C.byref(ns), C.byref(er)), "sosf_tracker_optimize_scale_kf")
        state[0], state[1] = int(st[0]), int(st[1])
        return ns.value, er.value

    def set_device_lm(self, on: bool):
        """LM loop of track / pose_estimate as one device launch (default) or on the host around device residual passes."""
        _chk(self.L.sosf_tracker_set_device_lm(self.h_, 1 if on else 0), "sosf_tracker_set_device_lm")

    def set_lm_spin_limit(self, rounds: int):
        _chk(self.L.sosf_tracker_set_lm_spin_limit(self.h_, C.c_uint(int(rounds))), "sosf_tracker_set_lm_spin_limit")

    def lm_fallbacks(self) -> int:
        n = C.c_int(0)
        _chk(self.L.sosf_tracker_lm_fallbacks(self.h_, C.byref(n)), "sosf_tracker_lm_fallbacks")
        return int(n.value)

    def lm_profile(self, hyp: int = 0) -> dict:
        """Phase times of the last one-launch LM loop from the kernel's own stamps (sosf_tracker_lm_profile), per launch and per evaluation."""
        us = np.zeros(7)
        self.L.sosf_tracker_lm_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _chk(self.L.sosf_tracker_lm_profile(self.h_, hyp, _p(us)), "sosf_tracker_lm_profile")
        ev = max(1.0, us[6])
        names = ("kernel", "residual_pass", "gather_chunk_sums", "bookkeeping", "solve_8x8", "se3_exp_and_request")
        out = {"evaluations": int(us[6]), "us_per_launch": {k: round(float(v), 2) for k, v in zip(names, us[:6])},
               "us_per_evaluation": {k: round(float(v / ev), 3) for k, v in zip(names, us[:6])}}
        return out

    def last_evals(self) -> int:
        n = C.c_int(0)
        _chk(self.L.sosf_tracker_last_evals(self.h_, C.byref(n)), "sosf_tracker_last_evals")
        return n.value

    def make_tries(self, slast_2_sprelast12, lastF_2_slast12, imu12=None, poses_valid=True):
        """The pose hypotheses of FullSystem::trackNewCoarse (FS/FullSystem.cpp:150-213) as an (n, 12) array."""
        a = np.ascontiguousarray(slast_2_sprelast12, dtype=np.float64)
        b = np.ascontiguousarray(lastF_2_slast12, dtype=np.float64)
        im = None if imu12 is None else np.ascontiguousarray(imu12, dtype=np.float64)
        out = np.zeros((96, 12))
        n = C.c_int(0)
        _chk(self.L.sosf_tracker_make_tries(_p(a), _p(b), None if im is None else _p(im), 1 if poses_valid else 0, 96, _p(out),
                                            C.byref(n)), "sosf_tracker_make_tries")
        return out[:n.value].copy()

    def track_hypotheses(self, newSlot, new_ab_exposure, tries12, aff_last2, coarsest, last_coarse_rmse5, retrack_threshold=1.5,
                         batch=16):
        """FullSystem::trackNewCoarse's loop over the tries (FS/FullSystem.cpp:219-283)."""
        tr = np.ascontiguousarray(tries12, dtype=np.float64).reshape(-1, 12)
        af = np.ascontiguousarray(aff_last2, dtype=np.float64)
        lc = np.ascontiguousarray(last_coarse_rmse5, dtype=np.float64)
        T, aff, ach, fl = np.zeros(12), np.zeros(2), np.zeros(5), np.zeros(3)
        info = np.zeros(4, dtype=np.int32)
        _chk(self.L.sosf_tracker_track_hypotheses(self.h_, newSlot, new_ab_exposure, len(tr), _p(tr), _p(af), coarsest, _p(lc),
                                                  float(retrack_threshold), batch, _p(T), _p(aff), _p(ach), _p(fl), _p(info)),
             "sosf_tracker_track_hypotheses")
        return dict(lastF_2_fh=T, aff=aff, achievedRes=ach, flow=fl, tryIterations=int(info[0]), chosen=int(info[1]),
                    evaluated=int(info[2]), haveOneGood=bool(info[3]))

    def set_points3d(self, calib, matched_ab_exposure, xyz, colors):
        """PoseEstimator template (loop-closure aligner): xyz (n, 3), colors (levels, n)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        col = np.ascontiguousarray(colors, dtype=np.float32)
        assert col.ndim == 2 and col.shape[1] == len(xyz)
        _chk(self.L.sosf_tracker_set_points3d(self.h_, C.byref(calib), matched_ab_exposure, len(xyz), _p(xyz), _p(col)),
             "sosf_tracker_set_points3d")

    def pose_estimate(self, newSlot, new_ab_exposure, refToNew12, coarsest, loop_direct_thres, inner_percent=90):
        T = np.ascontiguousarray(refToNew12, dtype=np.float64).copy()
        err, pct, ok = C.c_float(0), C.c_int(0), C.c_int(0)
        _chk(self.L.sosf_tracker_pose_estimate(self.h_, newSlot, new_ab_exposure, _p(T), coarsest, loop_direct_thres, inner_percent,
                                               C.byref(err), C.byref(pct), C.byref(ok)), "sosf_tracker_pose_estimate")
        return bool(ok.value), T, err.value, pct.value

    def optimize_scale(self, stereoSlot, tfm12, K1, scale, coarsest):
        s = C.c_float(scale)
        r = C.c_float(0)
        tf = np.ascontiguousarray(tfm12, dtype=np.float64)
        k1 = np.ascontiguousarray(K1, dtype=np.float32)
        _chk(self.L.sosf_tracker_optimize_scale(self.h_, stereoSlot, _p(tf), _p(k1), C.byref(s), coarsest, C.byref(r)),
             "sosf_tracker_optimize_scale")
        return r.value, s.value

class _ImuApi:
    """Shared ctypes plumbing of the IMU / spline assembly (facade: sosf_imu_*, oracle: orc_imu_*)."""

    def __init__(self, L, prefix):
        self.L, self.p = L, prefix
        vp = C.c_void_p
        getattr(L, prefix + "get_Hi").argtypes = [vp, vp, vp, C.c_double, vp, vp, vp, vp, vp]
        getattr(L, prefix + "hessian").argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
        getattr(L, prefix + "expand").argtypes = [C.c_int, vp, vp, vp, vp]
        getattr(L, prefix + "marginalize_frame").argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_double, vp, vp, vp, vp]
        getattr(L, prefix + "solve").argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp, vp, vp]
        if prefix == "sosf_imu_":   # the facade's two-call form and its kept factor (no counterpart in the oracle)
            L.sosf_imu_solve_prepare.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, C.c_double, C.c_uint64]
            L.sosf_imu_solve_finish.argtypes = [vp, vp, vp, vp, vp, vp, vp]
            L.sosf_imu_solve_mode.argtypes = [C.c_int]
            L.sosf_imu_solve_stats.argtypes = [vp, vp, vp, C.c_int]

    @staticmethod
    def _frames(frames):
        from sos_slam_amd.records import ImuFrame
        arr = (ImuFrame * len(frames))(*frames)
        return arr

    def get_Hi(self, S, Cal, frame, tt):
        JsTW, JfTW, Hss, Hff, Hfs = np.zeros(6), np.zeros((29, 6)), C.c_double(0), np.zeros((29, 29)), np.zeros(29)
        getattr(self.L, self.p + "get_Hi")(C.byref(S), C.byref(Cal), C.byref(frame), tt, _p(JsTW), _p(JfTW), C.byref(Hss), _p(Hff), _p(Hfs))
        return JsTW, JfTW, Hss.value, Hff, Hfs

    def hessian(self, S, Cal, frames):
        from sos_slam_amd.records import imu_dim
        n = len(frames)
        dim = imu_dim(n)
        H, b = np.zeros((dim, dim)), np.zeros(dim)
        J, r = np.zeros((6 * n, dim)), np.zeros(6 * n)
        nc = C.c_int32(0)
        sv = np.zeros(n, np.int32)
        arr = self._frames(frames)
        getattr(self.L, self.p + "hessian")(C.byref(S), C.byref(Cal), n, arr, _p(H), _p(b), _p(J), _p(r), C.byref(nc), _p(sv))
        return H, b, J[:nc.value], r[:nc.value], sv

    def expand(self, n, H, b):
        from sos_slam_amd.records import imu_dim
        dim = imu_dim(n)
        He, be = np.zeros((dim, dim)), np.zeros(dim)
        H = np.ascontiguousarray(H, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        getattr(self.L, self.p + "expand")(n, _p(H), _p(b), _p(He), _p(be))
        return He, be

    def marginalize_frame(self, S, Cal, frames, idx, delta, prior8, delta_prior8, HM, bM, marg_weight=0.25):
        from sos_slam_amd.records import imu_dim
        n = len(frames)
        nd = imu_dim(n - 1)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (delta, prior8, delta_prior8, HM, bM)]
        Ho, bo = np.zeros((nd, nd)), np.zeros(nd)
        arr = self._frames(frames)
        getattr(self.L, self.p + "marginalize_frame")(C.byref(S), C.byref(Cal), n, arr, idx, _p(a[0]), _p(a[1]), _p(a[2]), marg_weight,
                                                      _p(a[3]), _p(a[4]), _p(Ho), _p(bo))
        return Ho, bo

    def solve(self, S, Cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta, lam=1e-5):
        n = len(frames)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (H_top, b_top, H_sc, b_sc, HM, bM, delta)]
        x, ss, si = np.zeros(4 + 8 * n), C.c_double(0), np.zeros((n, 21))
        arr = self._frames(frames)
        getattr(self.L, self.p + "solve")(C.byref(S), C.byref(Cal), n, arr, *[_p(v) for v in a], lam, _p(x), C.byref(ss), _p(si))
        return x, ss.value, si


    def solve_mode(self, mode=-1):
        """0: the literal form for every solve, 1: the kept factor with first-estimate Jacobians (default); returns the previous mode"""
        return self.L.sosf_imu_solve_mode(mode)

    def solve_stats(self, reset=False):
        """(solves on a kept factor, factor rebuilds, literal-form solves) of this thread"""
        k, r, l = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.L.sosf_imu_solve_stats(C.byref(k), C.byref(r), C.byref(l), int(reset))
        return k.value, r.value, l.value

    def solve_two_calls(self, S, Cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta, lam=1e-5, prior_id=0, between=None):
        """sosf_imu_solve_prepare, then `between()` (what the facade does there: wait for the device), then sosf_imu_solve_finish"""
        n = len(frames)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (H_top, b_top, H_sc, b_sc, HM, bM, delta)]
        x, ss, si = np.zeros(4 + 8 * n), C.c_double(0), np.zeros((n, 21))
        arr = self._frames(frames)
        _chk(self.L.sosf_imu_solve_prepare(C.byref(S), C.byref(Cal), n, arr, _p(a[4]), _p(a[5]), _p(a[6]), lam, prior_id), "sosf_imu_solve_prepare")
        if between is not None:
            between()
        _chk(self.L.sosf_imu_solve_finish(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(x), C.byref(ss), _p(si)), "sosf_imu_solve_finish")
        return x, ss.value, si


class ImuFrontEnd:
    """The VIO front-end functions of the facade (sosf_imu_propagate_state / update_vel / initialize / try_trap_scale)."""

    def __init__(self):
        self.L = load()
        vp = C.c_void_p
        self.L.sosf_imu_propagate_state.argtypes = [vp, vp, vp, vp, vp, vp]
        self.L.sosf_imu_update_vel.argtypes = [vp, vp, vp]
        self.L.sosf_imu_initialize.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int)]
        self.L.sosf_imu_try_trap_scale.argtypes = [vp, vp, C.POINTER(C.c_int32), C.c_double]

    def propagate_state(self, S, Cal, frame, shell, last_shell, last_bias6):
        lb = np.ascontiguousarray(last_bias6, dtype=np.float64)
        _chk(self.L.sosf_imu_propagate_state(C.byref(S), C.byref(Cal), C.byref(frame), C.byref(shell), C.byref(last_shell), _p(lb)),
             "sosf_imu_propagate_state")

    def update_vel(self, frame, shell, last_shell):
        _chk(self.L.sosf_imu_update_vel(C.byref(frame), C.byref(shell), C.byref(last_shell)), "sosf_imu_update_vel")

    def initialize(self, S, Cal, frames5, shells5):
        from sos_slam_amd.records import ImuFrame, ImuShell
        fa, sa = (ImuFrame * 5)(*frames5), (ImuShell * 5)(*shells5)
        ok = C.c_int(0)
        _chk(self.L.sosf_imu_initialize(C.byref(S), C.byref(Cal), fa, sa, C.byref(ok)), "sosf_imu_initialize")
        return bool(ok.value), list(fa), list(sa)

    def try_trap_scale(self, Cal, queue10, qi, thres):
        q = np.ascontiguousarray(queue10, dtype=np.float64).copy()
        i = C.c_int32(qi)
        _chk(self.L.sosf_imu_try_trap_scale(C.byref(Cal), _p(q), C.byref(i), thres), "sosf_imu_try_trap_scale")
        return q, i.value


def imu():
    """The facade's IMU / spline factor assembly (sosf_imu_*)."""
    return _ImuApi(load(), "sosf_imu_")


class Sequence:
    """sosf_sequence: the frame-rate loop of FullSystem in C++ (addActiveFrame -> track -> traceNewCoarse -> keyframe decision ->
    makeKeyFrame), csrc/host/sos_sequence.cpp.  `sysm` holds the bootstrap window."""

    def __init__(self, sysm: System, params, random_pattern):
        from .records import SequenceParams
        self.L, self.sysm = load(), sysm
        assert isinstance(params, SequenceParams)
        self.params = params
        pat = np.ascontiguousarray(random_pattern, dtype=np.uint8)
        self.h_ = C.c_void_p()
        self.L.sosf_sequence_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_create(sysm.h_, C.byref(self.params), _p(pat), C.byref(self.h_)), "sosf_sequence_create")
        sysm._trackers = getattr(sysm, "_trackers", [])
        sysm._trackers.append(self)     # closed before the system (it borrows the context)

    def close(self):
        if getattr(self, "h_", None):
            if getattr(self.sysm, "h_", None):  # after the system is gone its context is gone too
                self.L.sosf_sequence_destroy.argtypes = [C.c_void_p]
                self.L.sosf_sequence_destroy(self.h_)
            self.h_ = None
        if self in getattr(self.sysm, "_trackers", []):
            self.sysm._trackers.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enable_imu(self, S, timestamps, imu_per_keyframe):
        """visual-inertial branch: timestamps / IMU samples (n x 7) of the bootstrap keyframes, in window order"""
        n = len(timestamps)
        ts = np.ascontiguousarray(timestamps, dtype=np.float64)
        arrs = [np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 7) for a in imu_per_keyframe]
        cnt = np.array([len(a) for a in arrs], np.int32)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data if len(a) else None for a in arrs])
        self._S = S
        self.L.sosf_sequence_enable_imu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_enable_imu(self.h_, C.byref(S), n, _p(ts), _p(cnt), ptrs), "sosf_sequence_enable_imu")

    def enable_stereo(self, tfm12, scale_opt_thres):
        t = np.ascontiguousarray(tfm12, dtype=np.float64)
        self.L.sosf_sequence_enable_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        _chk(self.L.sosf_sequence_enable_stereo(self.h_, _p(t), float(scale_opt_thres)), "sosf_sequence_enable_stereo")

    def bootstrap(self, stereo_slot=-1):
        r, it = C.c_float(0), C.c_int(0)
        self.L.sosf_sequence_bootstrap_ex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_bootstrap_ex(self.h_, int(stereo_slot), C.byref(r), C.byref(it)), "sosf_sequence_bootstrap_ex")
        return r.value, it.value

    def add_active_frame(self, slot, frame_id, T_init=None, ab_exposure=1.0, timestamp=None, imu=None, stereo_slot=-1):
        from .records import FrameExtra, FrameResult
        out = FrameResult()
        t = None if T_init is None else np.ascontiguousarray(T_init, dtype=np.float64)
        ex = None
        if timestamp is not None or imu is not None or stereo_slot >= 0:
            a = np.ascontiguousarray(imu if imu is not None else np.zeros((0, 7)), dtype=np.float64).reshape(-1, 7)
            ex = FrameExtra(float(timestamp or 0.0), len(a), int(stereo_slot), a.ctypes.data if len(a) else None)
        self.L.sosf_add_active_frame_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_add_active_frame_ex(self.h_, int(slot), int(frame_id), float(ab_exposure), _p(t), C.byref(ex) if ex is not None else None,
                                             C.byref(out)), "sosf_add_active_frame_ex")
        return out

    def imu(self, frame_id):
        """(state_imu (21, unscaled), state_imu_zero, velInWorld) of a keyframe in the window"""
        st, ze, ve = np.zeros(21), np.zeros(21), np.zeros(3)
        self.L.sosf_sequence_get_imu.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_get_imu(self.h_, int(frame_id), _p(st), _p(ze), _p(ve)), "sosf_sequence_get_imu")
        return st, ze, ve

    def set_snapshots(self, on=True):
        self.L.sosf_sequence_set_snapshots.argtypes = [C.c_void_p, C.c_int]
        _chk(self.L.sosf_sequence_set_snapshots(self.h_, int(on)), "sosf_sequence_set_snapshots")

    def snapshot(self, which):
        """graph state between the stages of the last makeKeyFrame: 0 flagged frameIDs, 1 activated (host frameID, u, v), 2 residuals
        after optimize (host frameID, u, v, target frameID), 3 points after flagPointsForRemoval (host frameID, u, v); rows as float64"""
        width = (1, 3, 4, 3)[which]
        c = C.c_int(0)
        self.L.sosf_sequence_get_snapshot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_get_snapshot(self.h_, which, 0, None, C.byref(c)), "sosf_sequence_get_snapshot")
        out = np.zeros(c.value)
        _chk(self.L.sosf_sequence_get_snapshot(self.h_, which, c.value, _p(out), C.byref(c)), "sosf_sequence_get_snapshot")
        return out.reshape(-1, width)

    def scale_state(self):
        st = np.zeros(2, np.int32)
        self.L.sosf_sequence_get_scale_state.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_get_scale_state(self.h_, _p(st)), "sosf_sequence_get_scale_state")
        return [int(st[0]), int(st[1])]

    def imu_calib(self):
        from .records import ImuCalib
        c = ImuCalib()
        self.L.sosf_sequence_get_imu_calib.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_get_imu_calib(self.h_, C.byref(c)), "sosf_sequence_get_imu_calib")
        return c

    def immature(self, frame_id):
        from .records import IMMATURE_DTYPE
        c = C.c_int(0)
        self.L.sosf_sequence_immature_count.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _chk(self.L.sosf_sequence_immature_count(self.h_, int(frame_id), C.byref(c)), "sosf_sequence_immature_count")
        rec, ty = np.zeros(c.value, dtype=IMMATURE_DTYPE), np.zeros(c.value, np.float32)
        self.L.sosf_sequence_get_immature.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _chk(self.L.sosf_sequence_get_immature(self.h_, int(frame_id), c.value, _p(rec), _p(ty)), "sosf_sequence_get_immature")
        return rec, ty


def host_frame_math(evalPT, state_zero, state, ab_exposure, calib_value, calib_value_zero):
    """The facade's per-keyframe host math on frames built for the occasion (sosf_host_frame_math): returns PRE_camToWorld (n, 12), the
    n x n precalc records (index host + n target), adHost / adTarget (n * n, 8, 8) and adHTdeltaF (n * n, 8)."""
    from . import synth
    ev = np.ascontiguousarray(evalPT, dtype=np.float64).reshape(-1, 12)
    n = len(ev)
    sz = np.ascontiguousarray(state_zero, dtype=np.float64).reshape(n, 10)
    st = np.ascontiguousarray(state, dtype=np.float64).reshape(n, 10)
    ab = np.ascontiguousarray(ab_exposure, dtype=np.float32).reshape(n)
    v, vz = np.ascontiguousarray(calib_value, dtype=np.float64), np.ascontiguousarray(calib_value_zero, dtype=np.float64)
    c2w = np.zeros((n, 12))
    pc = np.zeros(n * n, dtype=synth.PRECALC_DTYPE)
    adH, adT = np.zeros((n * n, 8, 8)), np.zeros((n * n, 8, 8))
    adHT = np.zeros((n * n, 8), np.float32)
    _chk(load().sosf_host_frame_math(n, _p(ev), _p(sz), _p(st), _p(ab), _p(v), _p(vz), _p(c2w), _p(pc), _p(adH), _p(adT), _p(adHT)),
         "sosf_host_frame_math")
    return c2w, pc, adH, adT, adHT


def new_frame_energy_th(energies, thn=0.7, fac_median=1.5, const_weight=0.5, overall=1.0):
    """The facade's setNewFrameEnergyTH on a list of energies (sosf_new_frame_energy_th)."""
    e = np.ascontiguousarray(energies, dtype=np.float32)
    th = C.c_float(0)
    _chk(load().sosf_new_frame_energy_th(_p(e), len(e), C.c_float(thn), C.c_float(fac_median), C.c_float(const_weight), C.c_float(overall), C.byref(th)),
         "sosf_new_frame_energy_th")
    return th.value


def flag_frames(frameID, points_in, points_out, ref_to_fh0, distanceLL):
    """The facade's decision of flagFramesForMarginalization on plain arrays (sosf_flag_frames); returns the flags (uint8, window order)."""
    ids = np.ascontiguousarray(frameID, dtype=np.int32)
    n = len(ids)
    pi, po = np.ascontiguousarray(points_in, dtype=np.int32), np.ascontiguousarray(points_out, dtype=np.int32)
    r0 = np.ascontiguousarray(ref_to_fh0, dtype=np.float64)
    d = np.ascontiguousarray(distanceLL, dtype=np.float32).reshape(n, n)
    fl = np.zeros(n, np.uint8)
    _chk(load().sosf_flag_frames(n, _p(ids), _p(pi), _p(po), _p(r0), _p(d), _p(fl)), "sosf_flag_frames")
    return fl
