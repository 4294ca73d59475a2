"""ctypes front-end of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (sos_slam_amd) never does.  The oracle is "parity unpinned" (see oracle/oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_FAST = None

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int32)


from sos_slam_amd.records import Calib, Params  # noqa: E402  (record mirrors only, no native code)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))] + [os.path.join(_HERE, "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "sos_slam.h"))
    stale = lambda t: not os.path.exists(t) or any(os.path.getmtime(s) > os.path.getmtime(t) for s in srcs)  # noqa: E731
    if force or stale(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib_fast():
    """The -O3 -march=native timing build of the same sources (bench.py's cpu_baseline leg); never the parity oracle.
    Built on THIS machine (it is bound to the host CPU), into a per-CPU scratch directory."""
    global _LIB_FAST
    if _LIB_FAST is None:
        import hashlib
        import tempfile
        try:
            flags = [ln for ln in open("/proc/cpuinfo") if ln.startswith(("flags", "model name"))][:2]
        except OSError:
            flags = []
        srcs = sorted(f for f in os.listdir(_HERE) if f.endswith((".c", ".h")))
        h = hashlib.sha256(("".join(flags) + "".join(open(os.path.join(_HERE, f)).read() for f in srcs)).encode()).hexdigest()[:16]
        out = os.path.join(tempfile.gettempdir(), "sos_oracle_fast_" + h)
        os.makedirs(out, exist_ok=True)
        so = os.path.join(out, "liboracle_fast.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s", "fast", "FASTOUT=" + out])
        _LIB_FAST = _bind(C.CDLL(so))
    return _LIB_FAST


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


def _bind(L):
    if True:
        vp = C.c_void_p
        L.orc_window_create.restype = vp
        L.orc_window_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int, vp, C.c_int, vp]
        L.orc_window_destroy.argtypes = [vp]
        L.orc_set_image.argtypes = [vp, C.c_int, vp]
        L.orc_set_lin.argtypes = [vp, vp, vp]
        L.orc_set_state.argtypes = [vp, C.POINTER(Calib)] + [vp] * 8
        L.orc_linearize_all.restype = C.c_double
        L.orc_linearize_all.argtypes = [vp, vp, C.c_int]
        L.orc_apply_res.argtypes = [vp]
        L.orc_reset_oob.argtypes = [vp]
        L.orc_fix_linearization.argtypes = [vp, vp, C.c_int]
        L.orc_accumulate.argtypes = [vp] + [vp] * 6 + [c_int_p, c_int_p, C.c_int, C.c_int]
        L.orc_resubstitute.argtypes = [vp, vp, vp, C.c_int]
        L.orc_calc_lenergy.restype = C.c_double
        L.orc_calc_lenergy.argtypes = [vp]
        L.orc_accumulate_marg.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, c_int_p]
        for name in ("orc_J", "orc_Jnew", "orc_res", "orc_pts", "orc_new_state", "orc_new_energy",
                     "orc_new_energy_wo", "orc_center", "orc_JpJdF", "orc_res_toZeroF"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp]
        L.orc_point_field.restype = vp
        L.orc_point_field.argtypes = [vp, C.c_int]
        L.orc_host_init.argtypes = [vp, vp, vp, vp, vp]
        L.orc_optimize.restype = C.c_float
        L.orc_optimize.argtypes = [vp, C.c_int, C.c_int, c_int_p]
        L.orc_optimize_ex.restype = C.c_float
        L.orc_optimize_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p]
        L.orc_num_good_residuals.restype = vp
        L.orc_num_good_residuals.argtypes = [vp]
        L.orc_host_init_ex.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_host_get_calib_value.argtypes = [vp, vp, vp]
        L.orc_host_get_evalpt.argtypes = [vp, C.c_int, vp]
        L.orc_tracker_set_truth_mode.argtypes = [vp, C.c_int]
        L.orc_gn_iteration.restype = C.c_int
        L.orc_gn_iteration.argtypes = [vp, C.c_int, C.c_int]
        L.orc_host_get_frame.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.orc_host_get_calib.argtypes = [vp, vp]
        L.orc_host_precalc.argtypes = [vp]
        L.orc_host_set_truth_mode.argtypes = [vp, C.c_int]
        for name in ("orc_host_get_precalc", "orc_host_get_adHTdeltaF", "orc_host_get_adHost",
                     "orc_host_get_adTarget", "orc_host_get_lastX"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp]
        L.orc_host_get_HM.argtypes = [vp, vp, vp]
        L.orc_immature_init.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.orc_immature_trace.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
        L.orc_immature_activate.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp]
        L.orc_camera_parse.argtypes = [C.c_char_p, vp]
        L.orc_undistort_setup.argtypes = [vp, vp, vp, vp, vp]
        L.orc_photometric_setup.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp]
        L.orc_undistort_frame.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, vp]
        L.orc_pixsel_make_hists.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_pixsel_select.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_float, vp, vp]
        L.orc_pixsel_make_maps.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_float, C.c_int, C.c_float, vp, vp]
        L.orc_activate_select.argtypes = [C.c_int] * 4 + [vp, vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int, vp, vp, vp, vp, vp, vp]
        L.orc_next_min_act_dist.argtypes = [C.c_float, C.c_int, C.c_float]
        L.orc_next_min_act_dist.restype = C.c_float
        L.orc_host_get_frame_prior.argtypes = [vp, C.c_int, vp, vp]
        L.orc_host_drop_points.argtypes = [vp, vp, C.c_int]
        L.orc_host_marginalize_points.argtypes = [vp, vp, C.c_int, vp]
        L.orc_host_marginalize_frame_prior.argtypes = [vp, C.c_int, vp, vp]
        L.orc_se3_exp12.argtypes = [vp, vp]
        L.orc_se3_log12.argtypes = [vp, vp]
        L.orc_se3_adj12.argtypes = [vp, vp]
        L.orc_se3_mul12.argtypes = [vp, vp, vp]
        L.orc_se3_inv12.argtypes = [vp, vp]
        L.orc_solve_ldlt.restype = C.c_int
        L.orc_solve_ldlt.argtypes = [vp, vp, vp, C.c_int]
        L.orc_pyr_levels.restype = C.c_int
        L.orc_pyr_levels.argtypes = [C.c_int, C.c_int]
        L.orc_make_images.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
        L.orc_tracker_create.restype = vp
        L.orc_tracker_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int]
        L.orc_tracker_destroy.argtypes = [vp]
        L.orc_tracker_set_ref.argtypes = [vp, C.POINTER(Calib), vp, C.c_int, vp, vp, vp, vp, vp]
        L.orc_tracker_get_pc.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.orc_tracker_scale_depth.argtypes = [vp, C.c_float]
        L.orc_tracker_calc_res.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_float, vp]
        L.orc_tracker_calc_gs.argtypes = [vp, C.c_int, C.c_float, C.c_float, vp, vp]
        L.orc_tracker_calc_res_scale.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, vp]
        L.orc_tracker_calc_gs_scale.argtypes = [vp, C.c_int, vp, vp, C.c_float, vp, vp]
        L.orc_tracker_set_points3d.argtypes = [vp, C.POINTER(Calib), C.c_int, vp, vp]
        L.orc_tracker_last_inners.argtypes = [vp, vp]
        L.orc_tracker_warp_n.restype = C.c_int
        L.orc_tracker_warp_n.argtypes = [vp]
        L.orc_tracker_track.restype = C.c_int
        L.orc_tracker_track.argtypes = [vp, vp, C.c_float, C.c_float, vp, vp, vp, C.c_int, vp, vp, vp]
        L.orc_tracker_optimize_scale.restype = C.c_float
        L.orc_tracker_optimize_scale.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    return L


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _view(ptr, dtype, shape):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def pyr_levels(w, h):
    return lib().orc_pyr_levels(w, h)


def make_images(img, gammaB=None):
    """FrameHessian::makeImages -> ([dI per level (hl,wl,3)], [absSquaredGrad per level])."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape
    lv = pyr_levels(w, h)
    dI = [np.zeros((h >> l, w >> l, 3), dtype=np.float32) for l in range(lv)]
    ab = [np.zeros((h >> l, w >> l), dtype=np.float32) for l in range(lv)]
    pd = (C.c_void_p * lv)(*[d.ctypes.data for d in dI])
    pa = (C.c_void_p * lv)(*[a.ctypes.data for a in ab])
    gb = None if gammaB is None else np.ascontiguousarray(gammaB, dtype=np.float32)
    lib().orc_make_images(_p(img), w, h, _p(gb), lv, pd, pa)
    return dI, ab


def immature_init(prm, host_dI0, u, v):
    """ImmaturePoint constructor over pixel positions (u, v) of a host frame (level-0 dI, (h, w, 3))."""
    from sos_slam_amd.records import IMMATURE_DTYPE
    dI = np.ascontiguousarray(host_dI0, dtype=np.float32)
    h, w = dI.shape[:2]
    u = np.ascontiguousarray(u, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.int32)
    out = np.zeros(len(u), dtype=IMMATURE_DTYPE)
    lib().orc_immature_init(C.byref(prm), _p(dI), w, h, len(u), _p(u), _p(v), _p(out))
    return out


def immature_trace(prm, frame_dI0, pts, KRKi, Kt, aff):
    """ImmaturePoint::traceOn over all points against one frame; returns the updated records."""
    dI = np.ascontiguousarray(frame_dI0, dtype=np.float32)
    h, w = dI.shape[:2]
    pts = np.ascontiguousarray(pts).copy()
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (KRKi, Kt, aff)]
    lib().orc_immature_trace(C.byref(prm), _p(dI), w, h, len(pts), _p(pts), *[_p(x) for x in a])
    return pts


def immature_activate(prm, calib, frame_dI0, pairs, pts, host_of):
    """FullSystem::optimizeImmaturePoint over all candidates; frame_dI0[f] = level-0 dI (h, w, 3) of frame idx f."""
    from sos_slam_amd.records import ACTIVATION_DTYPE, PAIR_TFM_DTYPE
    imgs = [np.ascontiguousarray(x, dtype=np.float32) for x in frame_dI0]
    n = len(imgs)
    h, w = imgs[0].shape[:2]
    ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in imgs])
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_TFM_DTYPE)
    assert pairs.size == n * n
    pts = np.ascontiguousarray(pts)
    ho = np.ascontiguousarray(host_of, dtype=np.int32)
    out = np.zeros(len(pts), dtype=ACTIVATION_DTYPE)
    lib().orc_immature_activate(C.byref(prm), C.byref(calib), w, h, n, ptrs, _p(pairs), len(pts), _p(pts), _p(ho), _p(out))
    return out


def _select_args(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, host_flagged):
    KRKi = np.ascontiguousarray(KRKi, dtype=np.float32).reshape(-1, 9)
    Kt = np.ascontiguousarray(Kt, dtype=np.float32).reshape(-1, 3)
    n = len(KRKi)
    au, av, aid = [np.ascontiguousarray(act[k], dtype=np.float32) for k in ("u", "v", "idepth_scaled")]
    ah = np.ascontiguousarray(act["host"], dtype=np.int32)
    cand = np.ascontiguousarray(cand)
    ch = np.ascontiguousarray(cand_host, dtype=np.int32)
    ct = np.ascontiguousarray(cand_type, dtype=np.float32)
    hf = np.ascontiguousarray(host_flagged, dtype=np.uint8)
    dec = np.zeros(len(cand), dtype=np.int8)
    dist = np.zeros((h1, w1), dtype=np.float32)
    keep = (KRKi, Kt, au, av, aid, ah, cand, ch, ct, hf)
    args = [w1, h1, n, newest, _p(KRKi), _p(Kt), len(au), _p(au), _p(av), _p(aid), _p(ah), float(min_dist), float(min_quality),
            len(cand), _p(cand), _p(ch), _p(ct), _p(hf), _p(dec), _p(dist)]
    return args, dec, dist, keep


def activate_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, host_flagged):
    """Candidate loop of FullSystem::activatePointsMT; returns (decision int8[nCand], fwdWarpedIDDistFinal (h1, w1))."""
    args, dec, dist, keep = _select_args(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, host_flagged)
    rc = lib().orc_activate_select(*args)
    assert rc == 0
    return dec, dist


def next_min_act_dist(cur, n_points, desired):
    return float(lib().orc_next_min_act_dist(cur, n_points, desired))


class Undistorter:
    """Undistort + PhotometricUndistorter restatement (orc_undistort.c)."""

    def __init__(self, text, G=None, vignette=None, photometric_mode=2):
        from sos_slam_amd.records import CameraModel
        self.cam = CameraModel()
        if lib().orc_camera_parse(text.encode(), C.byref(self.cam)) != 0:
            raise ValueError("camera file rejected")
        m = self.cam
        self.K = np.zeros(4, np.float64)
        self.remapX = np.zeros((m.h, m.w), np.float32)
        self.remapY = np.zeros_like(self.remapX)
        pt = C.c_int(0)
        if lib().orc_undistort_setup(C.byref(m), _p(self.K), _p(self.remapX), _p(self.remapY), C.byref(pt)) != 0:
            raise ValueError("no rectification found")
        self.passthrough = bool(pt.value)
        self.mode = photometric_mode
        self.valid = 0
        self.G = self.vinv = None
        if G is not None and vignette is not None:
            self.G = np.ascontiguousarray(G, dtype=np.float32).copy()
            v = np.ascontiguousarray(vignette, dtype=np.float32)
            self.vinv = np.zeros(v.size, np.float32)
            self.valid = lib().orc_photometric_setup(_p(self.G), len(self.G), _p(v), v.size, photometric_mode, _p(self.vinv))

    def frame(self, raw, exposure, factor=1.0):
        raw = np.ascontiguousarray(raw)
        out = np.zeros((self.cam.h, self.cam.w), np.float32)
        lib().orc_undistort_frame(C.byref(self.cam), _p(self.remapX), _p(self.remapY), int(self.passthrough), _p(self.G), int(self.valid),
                                  _p(self.vinv), self.mode, _p(raw), raw.dtype.itemsize, exposure, factor, _p(out))
        return out


class PixelSelector:
    """PixelSelector restatement over one frame's level-0 dI and absSquaredGrad levels 0..2 (orc_make_images output)."""

    def __init__(self, prm, pattern, w, h):
        self.prm, self.w, self.h = prm, w, h
        self.pattern = np.ascontiguousarray(pattern, dtype=np.uint8)
        self.current_potential = 3
        self.ths = self.sm = None

    def make_hists(self, absg0):
        n = (self.w // 32) * (self.h // 32)
        self.ths, self.sm = np.zeros(n, np.float32), np.zeros(n, np.float32)
        a = np.ascontiguousarray(absg0, dtype=np.float32)
        lib().orc_pixsel_make_hists(C.byref(self.prm), _p(a), self.w, self.h, _p(self.ths), _p(self.sm))
        return self.ths, self.sm

    def _imgs(self, dI, absg):
        return [np.ascontiguousarray(dI[0], dtype=np.float32)] + [np.ascontiguousarray(absg[l], dtype=np.float32) for l in range(3)]

    def select(self, dI, absg, pot, th_factor=1.0):
        im = self._imgs(dI, absg)
        m = np.zeros((self.h, self.w), np.float32)
        n = np.zeros(3, np.int32)
        lib().orc_pixsel_select(C.byref(self.prm), _p(im[0]), _p(im[1]), _p(im[2]), _p(im[3]), self.w, self.h, _p(self.pattern),
                                _p(self.sm), pot, th_factor, _p(m), _p(n))
        return m, n

    def make_maps(self, dI, absg, density, recursions_left=1, th_factor=1.0):
        im = self._imgs(dI, absg)
        m = np.zeros((self.h, self.w), np.float32)
        pot = C.c_int(self.current_potential)
        lib().orc_pixsel_make_maps.restype = C.c_int
        num = lib().orc_pixsel_make_maps(C.byref(self.prm), _p(im[0]), _p(im[1]), _p(im[2]), _p(im[3]), self.w, self.h,
                                         _p(self.pattern), _p(self.sm), density, recursions_left, th_factor, C.byref(pot), _p(m))
        self.current_potential = pot.value
        return m, int(num)


class OracleWindow:
    """One EnergyFunctional window inside the oracle."""

    def __init__(self, params: dict, n: int, points: np.ndarray, resid: np.ndarray, fast: bool = False):
        from sos_slam_amd import synth  # dtypes only
        self._synth = synth
        self.L = lib_fast() if fast else lib()
        self.params = Params.from_dict(params)
        self.n, self.P, self.R = n, len(points), len(resid)
        pts = np.ascontiguousarray(points)
        res = np.ascontiguousarray(resid)
        self.h = self.L.orc_window_create(C.byref(self.params), n, self.P, _p(pts), self.R, _p(res))
        self._keep = []

    def close(self):
        if self.h:
            self.L.orc_window_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- inputs
    def set_image(self, frame, dI):
        dI = np.ascontiguousarray(dI, dtype=np.float32)
        self._keep.append(dI)
        self.L.orc_set_image(self.h, frame, _p(dI))

    def set_lin(self, res_toZeroF, linJ):
        self.L.orc_set_lin(self.h, _p(res_toZeroF), _p(linJ))

    def set_state(self, calib=None, precalc=None, adHTdeltaF=None, cDeltaF=None, adHost=None, adTarget=None,
                  idepth=None, idepth_zero=None, deltaF=None):
        self.L.orc_set_state(self.h, C.byref(calib) if calib is not None else None, _p(precalc),
                             _p(adHTdeltaF), _p(cDeltaF), _p(adHost), _p(adTarget), _p(idepth),
                             _p(idepth_zero), _p(deltaF))

    # --- kernel-level calls
    def linearize(self, frameEnergyTH, nthreads=1):
        th = np.ascontiguousarray(frameEnergyTH, dtype=np.float32)
        return self.L.orc_linearize_all(self.h, _p(th), nthreads)

    def apply_res(self):
        self.L.orc_apply_res(self.h)

    def reset_oob(self):
        self.L.orc_reset_oob(self.h)

    def fix_linearization(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.L.orc_fix_linearization(self.h, _p(idx), len(idx))

    def accumulate(self, fp64_truth=False, nthreads=1):
        dim = 4 + 8 * self.n
        out = [np.zeros((dim, dim)), np.zeros(dim), np.zeros((dim, dim)), np.zeros(dim), np.zeros((dim, dim)),
               np.zeros(dim)]
        ra, rl = C.c_int32(0), C.c_int32(0)
        self.L.orc_accumulate(self.h, *[_p(o) for o in out], C.byref(ra), C.byref(rl), int(fp64_truth), nthreads)
        return dict(H_A=out[0], b_A=out[1], H_L=out[2], b_L=out[3], H_sc=out[4], b_sc=out[5], resInA=ra.value,
                    resInL=rl.value)

    def accumulate_marg(self, point_idx):
        dim = 4 + 8 * self.n
        idx = np.ascontiguousarray(point_idx, dtype=np.int32)
        out = [np.zeros((dim, dim)), np.zeros(dim), np.zeros((dim, dim)), np.zeros(dim)]
        rm = C.c_int32(0)
        self.L.orc_accumulate_marg(self.h, _p(idx), len(idx), *[_p(o) for o in out], C.byref(rm))
        return dict(M=out[0], Mb=out[1], Msc=out[2], Mbsc=out[3], resInM=rm.value)

    def resubstitute(self, x, nthreads=1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        step = np.zeros(self.P, dtype=np.float32)
        self.L.orc_resubstitute(self.h, _p(x), _p(step), nthreads)
        return step

    def calc_lenergy(self):
        return self.L.orc_calc_lenergy(self.h)

    # --- views
    def J(self):
        return _view(self.L.orc_J(self.h), self._synth.RAWJAC_DTYPE, (self.R,))

    def Jnew(self):
        return _view(self.L.orc_Jnew(self.h), self._synth.RAWJAC_DTYPE, (self.R,))

    def res(self):
        return _view(self.L.orc_res(self.h), self._synth.RESID_DTYPE, (self.R,))

    def pts(self):
        return _view(self.L.orc_pts(self.h), self._synth.POINT_DTYPE, (self.P,))

    def new_state(self):
        return _view(self.L.orc_new_state(self.h), np.int32, (self.R,))

    def new_energy(self):
        return _view(self.L.orc_new_energy(self.h), np.float32, (self.R,))

    def new_energy_wo(self):
        return _view(self.L.orc_new_energy_wo(self.h), np.float32, (self.R,))

    def center(self):
        return _view(self.L.orc_center(self.h), np.float32, (self.R, 3))

    def JpJdF(self):
        return _view(self.L.orc_JpJdF(self.h), np.float32, (self.R, 8))

    def res_toZeroF(self):
        return _view(self.L.orc_res_toZeroF(self.h), np.float32, (self.R, 8))

    def point_field(self, which):
        names = ["idepth_hessian", "HdiF", "bdSumF", "Hdd_accAF", "bd_accAF", "Hcd_accAF", "Hdd_accLF",
                 "bd_accLF", "Hcd_accLF", "step", "maxRelBaseline"]
        k = names.index(which)
        shape = (self.P, 4) if which.startswith("Hcd") else (self.P,)
        return _view(self.L.orc_point_field(self.h, k), np.float32, shape)

    # --- host level
    def host_init(self, frames, K, HM=None, bM=None):
        fr = np.ascontiguousarray(frames)
        Kd = np.ascontiguousarray(K, dtype=np.float64)
        self.L.orc_host_init(self.h, _p(fr), _p(Kd), _p(None if HM is None else np.ascontiguousarray(HM)),
                             _p(None if bM is None else np.ascontiguousarray(bM)))

    def set_truth_mode(self, on=True):
        self.L.orc_host_set_truth_mode(self.h, int(on))

    def host_precalc(self):
        self.L.orc_host_precalc(self.h)

    def optimize(self, iters=6, nthreads=1):
        it = C.c_int32(0)
        rmse = self.L.orc_optimize(self.h, iters, nthreads, C.byref(it))
        return rmse, it.value

    def optimize_ex(self, iters=6, force_accept=True, nthreads=1):
        """optimize() with setting_forceAceptStep selectable; returns (rmse, iterations, rejected steps)."""
        it, rej = C.c_int32(0), C.c_int32(0)
        rmse = self.L.orc_optimize_ex(self.h, iters, nthreads, int(force_accept), C.byref(it), C.byref(rej))
        return rmse, it.value, rej.value

    def set_imu(self, S=None, cal=None, frames=None, HM=None, bM=None):
        """IMU branch of the host loop's solveSystemF (orc_host_set_imu); records as in sos_slam_amd.records, kept alive here."""
        self.L.orc_host_set_imu.argtypes = [C.c_void_p] * 6
        self.L.orc_host_set_imu.restype = None
        if S is None:
            self._imu = None
            self.L.orc_host_set_imu(self.h, None, None, None, None, None)
            return
        from sos_slam_amd.records import ImuFrame
        arr = (ImuFrame * len(frames))(*frames)
        hm = np.ascontiguousarray(HM, dtype=np.float64)
        bm = np.ascontiguousarray(bM, dtype=np.float64)
        self._imu = (S, cal, arr, hm, bm)
        self.L.orc_host_set_imu(self.h, C.addressof(S), C.addressof(cal), C.addressof(arr), _p(hm), _p(bm))

    def imu_state(self):
        S, cal, arr, _, _ = self._imu
        return cal.scale, np.array([list(f.state_imu) for f in arr])

    def num_good_residuals(self):
        return _view(self.L.orc_num_good_residuals(self.h), np.int32, (self.P,))

    def host_init_ex(self, frames_ex, calib_value, calib_value_zero, HM, bM):
        """rolling-window init: frames_ex = FRAME_INIT_EX_DTYPE records (evalPT pose, state, state_zero, ...)"""
        fr = np.ascontiguousarray(frames_ex)
        v = np.ascontiguousarray(calib_value, dtype=np.float64)
        vz = np.ascontiguousarray(calib_value_zero, dtype=np.float64)
        self.L.orc_host_init_ex(self.h, _p(fr), _p(v), _p(vz), _p(np.ascontiguousarray(HM, dtype=np.float64)),
                                _p(np.ascontiguousarray(bM, dtype=np.float64)))

    def calib_value(self):
        v, vz = np.zeros(4), np.zeros(4)
        self.L.orc_host_get_calib_value(self.h, _p(v), _p(vz))
        return v, vz

    def evalpt(self, f):
        c = np.zeros(12)
        self.L.orc_host_get_evalpt(self.h, f, _p(c))
        return c

    def gn_iteration(self, iteration=0, nthreads=1):
        return self.L.orc_gn_iteration(self.h, iteration, nthreads)

    def frame(self, f):
        c2w, st, sz = np.zeros(12), np.zeros(10), np.zeros(10)
        th = C.c_float(0)
        self.L.orc_host_get_frame(self.h, f, _p(c2w), _p(st), _p(sz), C.byref(th))
        return dict(camToWorld=c2w, state=st, state_zero=sz, frameEnergyTH=th.value)

    def calib_value_scaled(self):
        v = np.zeros(4)
        self.L.orc_host_get_calib(self.h, _p(v))
        return v

    def precalc(self):
        return _view(self.L.orc_host_get_precalc(self.h), self._synth.PRECALC_DTYPE, (self.n * self.n,))

    def adHTdeltaF(self):
        return _view(self.L.orc_host_get_adHTdeltaF(self.h), np.float32, (self.n * self.n, 8))

    def adHost(self):
        return _view(self.L.orc_host_get_adHost(self.h), np.float64, (self.n * self.n, 8, 8))

    def adTarget(self):
        return _view(self.L.orc_host_get_adTarget(self.h), np.float64, (self.n * self.n, 8, 8))

    def lastX(self):
        return _view(self.L.orc_host_get_lastX(self.h), np.float64, (4 + 8 * self.n,))

    def get_prior(self):
        dim = 4 + 8 * self.n
        HM, bM = np.zeros((dim, dim)), np.zeros(dim)
        self.L.orc_host_get_HM(self.h, _p(HM), _p(bM))
        return HM, bM

    def frame_prior(self, f):
        pr, dp = np.zeros(8), np.zeros(8)
        self.L.orc_host_get_frame_prior(self.h, f, _p(pr), _p(dp))
        return pr, dp

    def drop_points(self, idx):
        a = np.ascontiguousarray(idx, dtype=np.int32)
        self.L.orc_host_drop_points(self.h, _p(a), len(a))

    def marginalize_points(self, idx):
        """flagPointsForRemoval (explicit list) + dropPointsF + marginalizePointsF; returns 1 = marginalised,
        0 = dropped per listed point."""
        a = np.ascontiguousarray(idx, dtype=np.int32)
        flag = np.zeros(len(a), dtype=np.int32)
        self.L.orc_host_marginalize_points(self.h, _p(a), len(a), _p(flag))
        return flag

    def marginalize_frame_prior(self, frame_idx):
        dim = 4 + 8 * self.n - 8
        HM, bM = np.zeros((dim, dim)), np.zeros(dim)
        self.L.orc_host_marginalize_frame_prior(self.h, frame_idx, _p(HM), _p(bM))
        return HM, bM


def window_from_synth(win, nthreads=1, fast=False):
    """OracleWindow with images (built by the oracle's makeImages) and host state of a synth.Window."""
    ow = OracleWindow(win.params, win.n, win.points, win.resid, fast=fast)
    ow.dI = []
    for i in range(win.n):
        dI, _ = make_images(win.images[i])
        ow.dI.append(dI)
        ow.set_image(i, dI[0])
    ow.host_init(win.frames, win.K, win.HM, win.bM)
    return ow


class OracleTracker:
    def __init__(self, params: dict, w: int, h: int):
        self.L = lib()
        self.params = Params.from_dict(params)
        self.w, self.h = w, h
        self.levels = pyr_levels(w, h)
        self.t = self.L.orc_tracker_create(C.byref(self.params), w, h)
        self._keep = []

    def close(self):
        if self.t:
            self.L.orc_tracker_destroy(self.t)
            self.t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pyr(self, dI_levels):
        arrs = [np.ascontiguousarray(d, dtype=np.float32) for d in dI_levels]
        self._keep.append(arrs)
        return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs]), arrs

    def set_ref(self, calib, ref_dI_levels, u, v, idepth, hdi):
        ptrs, _ = self._pyr(ref_dI_levels)
        self._ref_ptrs = ptrs
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (u, v, idepth, hdi)]
        pc_n = np.zeros(self.levels, dtype=np.int32)
        self.L.orc_tracker_set_ref(self.t, C.byref(calib), ptrs, len(a[0]), *[_p(x) for x in a], _p(pc_n))
        self.pc_n = pc_n
        return pc_n

    def get_pc(self, lvl):
        n = int(self.pc_n[lvl])
        out = [np.zeros(n, dtype=np.float32) for _ in range(4)]
        self.L.orc_tracker_get_pc(self.t, lvl, *[_p(o) for o in out])
        return out

    def scale_depth(self, s):
        self.L.orc_tracker_scale_depth(self.t, s)

    def calc_res(self, lvl, new_dI, RKi, t, affLL, cutoff):
        rs = np.zeros(6)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (new_dI, RKi, t, affLL)]
        self.L.orc_tracker_calc_res(self.t, lvl, *[_p(x) for x in a], cutoff, _p(rs))
        return rs

    def calc_gs(self, lvl, a, b0):
        H, b = np.zeros((8, 8)), np.zeros(8)
        self.L.orc_tracker_calc_gs(self.t, lvl, a, b0, _p(H), _p(b))
        return H, b

    def calc_res_scale(self, lvl, stereo_dI, RKi, t, K1, scale, cutoff):
        rs = np.zeros(6)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (stereo_dI, RKi, t, K1)]
        self.L.orc_tracker_calc_res_scale(self.t, lvl, *[_p(x) for x in a], scale, cutoff, _p(rs))
        return rs

    def calc_gs_scale(self, lvl, t, K1, scale):
        H, b = C.c_float(0), C.c_float(0)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (t, K1)]
        self.L.orc_tracker_calc_gs_scale(self.t, lvl, _p(a[0]), _p(a[1]), scale, C.byref(H), C.byref(b))
        return H.value, b.value

    def warp_n(self):
        return self.L.orc_tracker_warp_n(self.t)

    def set_truth_mode(self, on=True):
        """yardstick: fp64 accumulation of the calcRes / calcGSSSE sums"""
        self.L.orc_tracker_set_truth_mode(self.t, int(on))

    def track(self, new_dI_levels, ref_ab, new_ab, ref_aff, lastToNew12, aff2, coarsest, minRes=None):
        ptrs, _ = self._pyr(new_dI_levels)
        T = np.ascontiguousarray(lastToNew12, dtype=np.float64).copy()
        aff = np.ascontiguousarray(aff2, dtype=np.float64).copy()
        ra = np.ascontiguousarray(ref_aff, dtype=np.float64)
        mr = np.full(5, np.nan) if minRes is None else np.ascontiguousarray(minRes, dtype=np.float64)
        lr, fl = np.zeros(5), np.zeros(3)
        ok = self.L.orc_tracker_track(self.t, ptrs, ref_ab, new_ab, _p(ra), _p(T), _p(aff), coarsest, _p(mr),
                                      _p(lr), _p(fl))
        return ok, T, aff, lr, fl

    def set_points3d(self, calib, xyz, colors):
        """PoseEstimator template: xyz (n, 3), colors (levels, n)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        col = np.ascontiguousarray(colors, dtype=np.float32)
        assert col.shape == (self.levels, len(xyz))
        self._n3d = len(xyz)
        self.L.orc_tracker_set_points3d(self.t, C.byref(calib), len(xyz), _p(xyz), _p(col))

    def pose_estimate(self, new_dI_levels, matched_ab, new_ab, refToNew12, coarsest, loop_direct_thres, inner_percent=90):
        """PoseEstimator::estimate: the LM loop with zero reference affine parameters and no abort thresholds, then the
        three acceptance tests (src/LoopClosure/PoseEstimator.cpp:455-485)."""
        aff_good, T, aff, lr, fl = self.track(new_dI_levels, matched_ab, new_ab, (0.0, 0.0), refToNew12, (0.0, 0.0), coarsest,
                                              minRes=np.full(5, np.inf))
        inn = np.zeros(6, np.int32)
        self.L.orc_tracker_last_inners(self.t, _p(inn))
        pose_error = np.float32(lr[0])
        inlier_percent = int(np.float32(100) * np.float32(inn[0]) / np.float32(self._n3d))
        ok = bool(aff_good) and bool(pose_error < np.float32(loop_direct_thres)) and inlier_percent > inner_percent
        return ok, T, float(pose_error), inlier_percent

    def optimize_scale(self, stereo_dI_levels, tfm12, K1, scale, coarsest):
        ptrs, _ = self._pyr(stereo_dI_levels)
        s = C.c_float(scale)
        tf = np.ascontiguousarray(tfm12, dtype=np.float64)
        k1 = np.ascontiguousarray(K1, dtype=np.float32)
        r = self.L.orc_tracker_optimize_scale(self.t, ptrs, _p(tf), _p(k1), C.byref(s), coarsest)
        return r, s.value


# ---- FullSystem::trackNewCoarse: hypothesis list and the loop over it (FS/FullSystem.cpp:150-283) -------------
def _se3(fn, *args):
    L = lib()
    out = np.zeros(12)
    getattr(L, fn)(*[_p(np.ascontiguousarray(a, dtype=np.float64)) for a in args], _p(out))
    return out


def se3_mul12(A, B):
    return _se3("orc_se3_mul12", A, B)


def se3_inv12(A):
    return _se3("orc_se3_inv12", A)


def se3_exp12(a6):
    return _se3("orc_se3_exp12", a6)


def se3_log12(T):
    out = np.zeros(6)
    lib().orc_se3_log12(_p(np.ascontiguousarray(T, dtype=np.float64)), _p(out))
    return out


_ROT_SIGNS = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (-1, 1, 0),
              (0, -1, 1), (-1, 0, 1), (1, -1, 0), (0, 1, -1), (1, 0, -1), (-1, -1, 0), (0, -1, -1), (-1, 0, -1), (-1, -1, -1),
              (-1, -1, 1), (-1, 1, -1), (-1, 1, 1), (1, -1, -1), (1, -1, 1), (1, 1, -1), (1, 1, 1)]


def _quat_se3(w, x, y, z):
    """Sophus::SE3(Quaterniond(w, x, y, z), 0): SO3's constructor normalises, Eigen's toRotationMatrix on the unit quaternion."""
    q = np.array([w, x, y, z], dtype=np.float64)
    q = q / np.sqrt(np.sum(q * q))
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return np.concatenate([R.reshape(-1), np.zeros(3)])


def make_track_tries(slast_2_sprelast, lastF_2_slast, imu=None, poses_valid=True):
    """lastF_2_fh_tries, FS/FullSystem.cpp:150-213."""
    ident = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
    if not poses_valid:
        return np.array([ident])
    fh_2_slast = np.asarray(slast_2_sprelast, dtype=np.float64)
    inv = se3_inv12(fh_2_slast)
    tries = []
    if imu is not None:
        tries.append(np.asarray(imu, dtype=np.float64))
    tries.append(se3_mul12(inv, lastF_2_slast))
    tries.append(se3_mul12(se3_mul12(inv, inv), lastF_2_slast))
    tries.append(se3_mul12(se3_inv12(se3_exp12(se3_log12(fh_2_slast) * 0.5)), lastF_2_slast))
    tries.append(np.asarray(lastF_2_slast, dtype=np.float64).copy())
    tries.append(ident.copy())
    const = se3_mul12(inv, lastF_2_slast)
    rot_delta = np.float32(0.02)
    while float(rot_delta) < 0.05:   # `float rot_delta; rot_delta < 0.05; rot_delta += 0.01`: the sum and the comparison are
        for rs in _ROT_SIGNS:        # double, the store rounds to float -> 0.02, 0.03, 0.04, then 0.05000000074 ends the loop
            d = [float(np.float32(r) * rot_delta) for r in rs]
            tries.append(se3_mul12(const, _quat_se3(1.0, *d)))
        rot_delta = np.float32(float(rot_delta) + 0.01)
    return np.array(tries)


def track_new_coarse(tracker, new_dI, ref_ab, new_ab, ref_aff, tries, aff_last, coarsest, last_coarse_rmse, retrack_threshold=1.5):
    """The loop of FS/FullSystem.cpp:219-283 around OracleTracker.track, one try after the other."""
    achieved = np.full(5, np.nan)
    flow = np.array([100.0, 100.0, 100.0])
    best_T = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
    best_aff = np.zeros(2)
    have, chosen, its = False, -1, 0
    for i, T0 in enumerate(tries):
        ok, T, aff, cur, fl = tracker.track(new_dI, ref_ab, new_ab, ref_aff, T0, aff_last, coarsest, minRes=achieved.copy())
        its += 1
        if ok and np.isfinite(np.float32(cur[0])) and not (cur[0] >= achieved[0]):
            flow, best_aff, best_T, have, chosen = fl.copy(), aff.copy(), T.copy(), True, i
        if have:
            for q in range(5):
                if (not np.isfinite(np.float32(achieved[q]))) or achieved[q] > cur[q]:
                    achieved[q] = cur[q]
        if have and achieved[0] < last_coarse_rmse[0] * retrack_threshold:
            break
    if not have:
        flow = np.zeros(3)
        best_aff = np.asarray(aff_last, dtype=np.float64).copy()
        best_T = np.asarray(tries[0], dtype=np.float64).copy()
    return dict(lastF_2_fh=best_T, aff=best_aff, achievedRes=achieved, flow=flow, tryIterations=its, chosen=chosen, haveOneGood=have)


def optimize_scale_kf(tracker, stereo_dI, tfm12, K1, tracking_ref_scale, coarsest, thres, state):
    """FullSystem::optimizeScale (FS/FullSystem.cpp:1117-1177) around OracleTracker.optimize_scale, one guess after the other.
    state = [scaleTrapped, scale_opt_fails] (updated in place).  Returns (new_scale or -1, scale_error)."""
    if thres <= 0:
        return 1.0, -1.0
    new_scale, err = np.float32(1.0), np.float32(-1.0)
    if state[0]:
        e, s = tracker.optimize_scale(stereo_dI, tfm12, K1, float(np.float32(tracking_ref_scale)), coarsest)
        new_scale, err = np.float32(s), np.float32(e)
    else:
        for g in (0.1, 0.2, 0.5, 1, 2, 5, 10):
            e, s = tracker.optimize_scale(stereo_dI, tfm12, K1, float(np.float32(g)), coarsest)
            if e > 0 and (err < 0 or err > e):
                new_scale, err = np.float32(s), np.float32(e)
    ok = 0 < err < thres
    state[1] = 0 if ok else state[1] + 1
    if state[1] > 5:
        state[0] = 0
    if not ok:
        return -1.0, float(err)
    if not state[0]:
        state[0] = 1
    return float(new_scale), float(err)


def imu():
    """The oracle's IMU / spline factor assembly (orc_imu_*), same call surface as sos_slam_amd.host.imu()."""
    from sos_slam_amd.host import _ImuApi
    return _ImuApi(lib(), "orc_imu_")
