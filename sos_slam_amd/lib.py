"""ctypes binding of csrc/libsos_slam_hip.so (the C-ABI of include/sos_slam.h).

There is NO CPU fallback: `load()` raises when the library has not been built or cannot be loaded,
and every context creation raises when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from .records import Calib, Params
from .synth import RAWJAC_DTYPE

_LIB = None

# every symbol include/sos_slam.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "sos_ctx_create", "sos_ctx_destroy", "sos_ctx_synchronize", "sos_ctx_pyr_levels", "sos_make_pyramid",
    "sos_frame_upload_dI", "sos_frame_download_level", "sos_frame_release", "sos_ba_create", "sos_ba_destroy",
    "sos_ba_set_window", "sos_ba_set_state", "sos_ba_linearize", "sos_ba_apply_res", "sos_ba_reset_oob",
    "sos_ba_fix_linearization", "sos_ba_accumulate", "sos_ba_accumulate_local", "sos_ba_acc_buffer",
    "sos_ba_stitch", "sos_ba_gn_accumulate", "sos_ba_gn_accumulate_begin", "sos_ba_gn_step", "sos_ba_gn_resub", "sos_ba_get_point_hessian", "sos_ba_resubstitute", "sos_ba_calc_lenergy",
    "sos_ba_accumulate_marg", "sos_ba_update_point_priors", "sos_ba_set_prefetch", "sos_tracker_set_gs_hint", "sos_immature_init", "sos_immature_trace", "sos_immature_trace_all", "sos_immature_activate", "sos_pixsel_create", "sos_pixsel_destroy", "sos_pixsel_make_hists", "sos_pixsel_select", "sos_pixsel_make_maps", "sos_pixsel_list", "sos_camera_parse", "sos_undistort_create", "sos_undistort_destroy", "sos_undistort_get", "sos_undistort_frame", "sos_rccl_load", "sos_rccl_unique_id", "sos_comm_create", "sos_comm_destroy", "sos_comm_size",
    "sos_comm_rank", "sos_ba_set_comm", "sos_ba_newest_capacity", "sos_ba_gather_energies", "sos_ba_allreduce_f64", "sos_ba_get_jacobian", "sos_ba_get_residual_flags", "sos_ba_get_JpJdF",
    "sos_ba_get_res_toZeroF", "sos_ba_time_kernel", "sos_gn_solve_system", "sos_tracker_create", "sos_tracker_destroy",
    "sos_tracker_set_ref", "sos_tracker_set_points3d", "sos_tracker_scale_depth", "sos_tracker_get_pc", "sos_tracker_calc_res",
    "sos_tracker_lm_profile", "sos_tracker_calc_gs", "sos_tracker_calc_res_scale", "sos_tracker_calc_gs_scale", "sos_backend_name",
]


class SosError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.HIP_LIB


def load():
    """Load the HIP library; raises SosError if it is missing (no silent fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise SosError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise SosError(f"cannot load {path}: {e}") from e
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.sos_backend_name.restype = C.c_char_p
    L.sos_ctx_create.argtypes = [ci, vp, ci, ci, C.POINTER(vp)]
    L.sos_ctx_destroy.argtypes = [vp]
    L.sos_ctx_synchronize.argtypes = [vp]
    L.sos_ctx_pyr_levels.argtypes = [vp]
    L.sos_make_pyramid.argtypes = [vp, ci, vp, vp]
    L.sos_frame_upload_dI.argtypes = [vp, ci, vp]
    L.sos_frame_download_level.argtypes = [vp, ci, ci, vp, vp]
    L.sos_frame_release.argtypes = [vp, ci]
    L.sos_ba_create.argtypes = [vp, C.POINTER(Params), C.POINTER(vp)]
    L.sos_ba_destroy.argtypes = [vp]
    L.sos_ba_set_window.argtypes = [vp, ci, vp, ci, vp, ci, vp, vp, vp]
    L.sos_ba_set_state.argtypes = [vp, C.POINTER(Calib)] + [vp] * 8
    L.sos_ba_linearize.argtypes = [vp, vp, C.POINTER(C.c_double), vp, vp, vp, vp]
    L.sos_ba_apply_res.argtypes = [vp]
    L.sos_ba_reset_oob.argtypes = [vp]
    L.sos_ba_fix_linearization.argtypes = [vp, vp, ci]
    L.sos_ba_accumulate.argtypes = [vp] + [vp] * 6 + [C.POINTER(ci), C.POINTER(ci)]
    L.sos_ba_accumulate_local.argtypes = [vp]
    L.sos_ba_acc_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.sos_ba_stitch.argtypes = [vp] + [vp] * 6 + [C.POINTER(ci), C.POINTER(ci)]
    L.sos_ba_gn_accumulate.argtypes = [vp, vp, vp, vp, vp, C.POINTER(ci), C.POINTER(ci)]
    L.sos_ba_gn_step.argtypes = [vp, vp, cf, C.POINTER(Calib), vp, vp, vp, vp, ci, C.POINTER(C.c_double), vp, C.POINTER(ci), vp]
    L.sos_ba_get_point_hessian.argtypes = [vp, vp, vp, vp]
    L.sos_ba_resubstitute.argtypes = [vp, vp, vp]
    L.sos_ba_calc_lenergy.argtypes = [vp, C.POINTER(C.c_double)]
    L.sos_ba_accumulate_marg.argtypes = [vp, vp, ci, vp, vp, vp, vp, C.POINTER(ci)]
    L.sos_ba_update_point_priors.argtypes = [vp, vp, vp, ci]
    L.sos_ba_set_prefetch.argtypes = [vp, ci]
    L.sos_tracker_set_gs_hint.argtypes = [vp, ci, C.c_float]
    L.sos_immature_init.argtypes = [vp, vp, ci, ci, vp, vp, vp]
    L.sos_immature_trace.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    L.sos_immature_trace_all.argtypes = [vp, vp, ci, ci, vp, vp, ci, vp, vp, vp]
    L.sos_immature_activate.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp, vp, vp]
    L.sos_immset_create.argtypes = [vp, C.POINTER(vp)]
    L.sos_immset_destroy.argtypes = [vp]
    L.sos_immset_destroy.restype = None
    L.sos_immset_put.argtypes = [vp, ci, ci, vp]
    L.sos_immset_trace.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    L.sos_immset_count.argtypes = [vp, ci, C.POINTER(ci)]
    L.sos_immset_get.argtypes = [vp, ci, ci, vp]
    L.sos_pixsel_create.argtypes = [vp, vp, vp, C.POINTER(vp)]
    L.sos_pixsel_destroy.argtypes = [vp]
    L.sos_pixsel_make_hists.argtypes = [vp, ci, vp, vp]
    L.sos_pixsel_select.argtypes = [vp, ci, ci, C.c_float, vp, vp]
    L.sos_pixsel_make_maps.argtypes = [vp, ci, C.c_float, ci, C.c_float, vp, vp, vp]
    L.sos_pixsel_list.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    L.sos_camera_parse.argtypes = [C.c_char_p, vp]
    L.sos_undistort_create.argtypes = [vp, vp, vp, ci, vp, ci, C.POINTER(vp)]
    L.sos_undistort_destroy.argtypes = [vp]
    L.sos_undistort_get.argtypes = [vp, vp, vp, vp, vp]
    L.sos_undistort_frame.argtypes = [vp, vp, ci, C.c_float, C.c_float, ci, vp, vp]
    L.sos_rccl_load.argtypes = [C.c_char_p]
    L.sos_rccl_unique_id.argtypes = [vp]
    L.sos_comm_create.argtypes = [vp, ci, ci, ci, C.POINTER(vp)]
    L.sos_comm_destroy.argtypes = [vp]
    L.sos_comm_size.argtypes = [vp]
    L.sos_comm_rank.argtypes = [vp]
    L.sos_ba_set_comm.argtypes = [vp, vp]
    L.sos_ba_newest_capacity.argtypes = [vp, C.POINTER(ci)]
    L.sos_ba_gather_energies.argtypes = [vp, vp, ci, vp, C.POINTER(ci)]
    L.sos_ba_allreduce_f64.argtypes = [vp, vp, C.c_size_t]
    L.sos_ba_get_jacobian.argtypes = [vp, ci, ci, vp]
    L.sos_ba_get_residual_flags.argtypes = [vp, vp, vp, vp]
    L.sos_ba_get_JpJdF.argtypes = [vp, vp]
    L.sos_ba_get_res_toZeroF.argtypes = [vp, vp]
    L.sos_ba_time_kernel.argtypes = [vp, C.c_char_p, vp, ci, C.POINTER(cf)]
    L.sos_tracker_create.argtypes = [vp, C.POINTER(Params), C.POINTER(vp)]
    L.sos_tracker_destroy.argtypes = [vp]
    L.sos_tracker_set_ref.argtypes = [vp, C.POINTER(Calib), ci, ci, vp, vp, vp, vp, vp]
    L.sos_tracker_set_points3d.argtypes = [vp, C.POINTER(Calib), ci, vp, vp]
    L.sos_tracker_scale_depth.argtypes = [vp, cf]
    L.sos_tracker_get_pc.argtypes = [vp, ci, vp, vp, vp, vp]
    L.sos_tracker_calc_res.argtypes = [vp, ci, ci, vp, vp, vp, cf, vp]
    L.sos_tracker_calc_gs.argtypes = [vp, ci, cf, cf, vp, vp]
    L.sos_tracker_calc_res_scale.argtypes = [vp, ci, ci, vp, vp, vp, cf, cf, vp]
    L.sos_tracker_calc_gs_scale.argtypes = [vp, ci, vp, vp, cf, vp, vp]
    _LIB = L
    return L


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc != 0:
        raise SosError(f"{what} failed with status {rc}")


class Context:
    """sos_ctx: device + stream + frame (pyramid) store."""

    def __init__(self, w: int, h: int, device: int = 0, stream: int | None = None):
        self.L = load()
        self.w, self.h = w, h
        self.h_ = C.c_void_p()
        _chk(self.L.sos_ctx_create(device, C.c_void_p(stream) if stream else None, w, h, C.byref(self.h_)),
             "sos_ctx_create (is an MI355X visible?)")
        self.levels = self.L.sos_ctx_pyr_levels(self.h_)

    def _adopt(self, child):
        """objects holding device state of this context are closed before it"""
        import weakref
        if not hasattr(self, "_children"):
            self._children = []
        self._children.append(weakref.ref(child))

    def close(self):
        for r in getattr(self, "_children", []):
            ch = r()
            if ch is not None:
                ch.close()
        self._children = []
        if getattr(self, "h_", None):
            self.L.sos_ctx_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _chk(self.L.sos_ctx_synchronize(self.h_), "sos_ctx_synchronize")

    def gn_solve_system(self, H_top, b_top, H_sc, b_sc, HM, bM, delta, reps: int = 1):
        """k_gn_solve (the solve of the device-resident Gauss-Newton loop) on a system from the host: the device counterpart of
        host.solve_system.  Returns (x, phases) with phases = microseconds (assemble, factorise, substitute, wall per launch)."""
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (H_top, b_top, H_sc, b_sc, HM, bM, delta)]
        n = (len(a[1]) - 4) // 8
        x = np.zeros(len(a[1]))
        ph = np.zeros(4)
        self.L.sos_gn_solve_system.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]
        _chk(self.L.sos_gn_solve_system(self.h_, n, *[_p(v) for v in a], _p(x), int(reps), _p(ph)), "sos_gn_solve_system")
        return x, ph

    def dbg_gs_row_update(self, a_in, nl):
        """debug: k_gn_solve's pivot-row update on one wave, the asm statements and the builtin form on the same input (sos_dbg_gs_row_update).
        a_in (64, 16), nl (64) -> (out_asm, out_ref), each (15, 64, 16): [K, lane, column]"""
        a = np.ascontiguousarray(a_in, dtype=np.float64).reshape(64, 16)
        m = np.ascontiguousarray(nl, dtype=np.float64).reshape(64)
        oa, orf = np.zeros((15, 64, 16)), np.zeros((15, 64, 16))
        self.L.sos_dbg_gs_row_update.argtypes = [C.c_void_p] * 5
        _chk(self.L.sos_dbg_gs_row_update(self.h_, _p(a), _p(m), _p(oa), _p(orf)), "sos_dbg_gs_row_update")
        return oa, orf

    # ---- immature points (ImmaturePoint constructor / traceOn)
    def immature_init(self, prm, host_slot: int, u, v):
        from .records import IMMATURE_DTYPE
        u = np.ascontiguousarray(u, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.int32)
        out = np.zeros(len(u), dtype=IMMATURE_DTYPE)
        _chk(self.L.sos_immature_init(self.h_, C.byref(prm), host_slot, len(u), _p(u), _p(v), _p(out)), "sos_immature_init")
        return out

    def immature_trace_all(self, prm, frame_slot: int, pts, host_of, KRKi, Kt, aff, copy=True):
        pts = np.ascontiguousarray(pts)
        if copy:
            pts = pts.copy()
        ho = np.ascontiguousarray(host_of, dtype=np.int32)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (KRKi, Kt, aff)]
        _chk(self.L.sos_immature_trace_all(self.h_, C.byref(prm), frame_slot, len(pts), _p(pts), _p(ho), len(a[1]) // 3 if a[1].ndim == 1 else a[1].shape[0],
                                           *[_p(x) for x in a]), "sos_immature_trace_all")
        return pts

    def immature_activate(self, prm, calib, frame_slots, pairs, pts, host_of):
        """FullSystem::optimizeImmaturePoint over all candidates; returns ACTIVATION_DTYPE records."""
        from .records import ACTIVATION_DTYPE, PAIR_TFM_DTYPE
        slots = np.ascontiguousarray(frame_slots, dtype=np.int32)
        n = len(slots)
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_TFM_DTYPE)
        assert pairs.size == n * n
        pts = np.ascontiguousarray(pts)
        ho = np.ascontiguousarray(host_of, dtype=np.int32)
        out = np.zeros(len(pts), dtype=ACTIVATION_DTYPE)
        _chk(self.L.sos_immature_activate(self.h_, C.byref(prm), C.byref(calib), n, _p(slots), _p(pairs), len(pts), _p(pts),
                                          _p(ho), _p(out)), "sos_immature_activate")
        return out

    def immature_trace(self, prm, frame_slot: int, pts, KRKi, Kt, aff):
        pts = np.ascontiguousarray(pts).copy()
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (KRKi, Kt, aff)]
        _chk(self.L.sos_immature_trace(self.h_, C.byref(prm), frame_slot, len(pts), _p(pts), *[_p(x) for x in a]),
             "sos_immature_trace")
        return pts

    def make_pyramid(self, slot: int, img: np.ndarray, gammaB: np.ndarray | None = None):
        img = np.ascontiguousarray(img, dtype=np.float32)
        assert img.shape == (self.h, self.w)
        gb = None if gammaB is None else np.ascontiguousarray(gammaB, dtype=np.float32)
        _chk(self.L.sos_make_pyramid(self.h_, slot, _p(img), _p(gb)), "sos_make_pyramid")

    def upload_dI(self, slot: int, dI: np.ndarray):
        dI = np.ascontiguousarray(dI, dtype=np.float32)
        assert dI.shape == (self.h, self.w, 3)
        _chk(self.L.sos_frame_upload_dI(self.h_, slot, _p(dI)), "sos_frame_upload_dI")

    def download_level(self, slot: int, lvl: int):
        wl, hl = self.w >> lvl, self.h >> lvl
        dI = np.zeros((hl, wl, 3), dtype=np.float32)
        ag = np.zeros((hl, wl), dtype=np.float32)
        _chk(self.L.sos_frame_download_level(self.h_, slot, lvl, _p(dI), _p(ag)), "sos_frame_download_level")
        return dI, ag

    def release(self, slot: int):
        _chk(self.L.sos_frame_release(self.h_, slot), "sos_frame_release")


class ImmatureSet:
    """sos_immset: the immature points of the window's keyframes resident on the device between frames (traceNewCoarse without
    host traffic); records cross the bus at keyframe decisions only."""

    def __init__(self, ctx: Context):
        self.L, self.ctx = ctx.L, ctx
        self.h_ = C.c_void_p()
        _chk(self.L.sos_immset_create(ctx.h_, C.byref(self.h_)), "sos_immset_create")

    def close(self):
        if self.h_:
            self.L.sos_immset_destroy(self.h_)
            self.h_ = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def put(self, host_key: int, pts):
        pts = np.ascontiguousarray(pts)
        _chk(self.L.sos_immset_put(self.h_, int(host_key), len(pts), _p(pts) if len(pts) else None), "sos_immset_put")

    def count(self, host_key: int) -> int:
        n = C.c_int(0)
        _chk(self.L.sos_immset_count(self.h_, int(host_key), C.byref(n)), "sos_immset_count")
        return int(n.value)

    def get(self, host_key: int):
        from .records import IMMATURE_DTYPE
        out = np.zeros(self.count(host_key), dtype=IMMATURE_DTYPE)
        _chk(self.L.sos_immset_get(self.h_, int(host_key), len(out), _p(out) if len(out) else None), "sos_immset_get")
        return out

    def trace(self, prm, frame_slot: int, host_keys, KRKi, Kt, aff):
        keys = np.ascontiguousarray(host_keys, dtype=np.int32)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (KRKi, Kt, aff)]
        assert a[0].size == 9 * len(keys) and a[1].size == 3 * len(keys) and a[2].size == 2 * len(keys)
        _chk(self.L.sos_immset_trace(self.h_, C.byref(prm), int(frame_slot), len(keys), _p(keys), *[_p(x) for x in a]), "sos_immset_trace")


class BorrowedContext(Context):
    """A sos_ctx owned by somebody else (the facade's FullSystem): same methods, close() only closes the objects created
    on it."""

    def __init__(self, handle: int, w: int, h: int):
        self.L = load()
        self.w, self.h = w, h
        self.h_ = C.c_void_p(handle)
        self.levels = self.L.sos_ctx_pyr_levels(self.h_)

    def close(self):
        for r in getattr(self, "_children", []):
            ch = r()
            if ch is not None:
                ch.close()
        self._children = []
        self.h_ = None


class Backend:
    """sos_ba: device side of one EnergyFunctional."""

    def __init__(self, ctx: Context, params: dict):
        self.L = ctx.L
        self.ctx = ctx
        self.params = Params.from_dict(params)
        self.h_ = C.c_void_p()
        _chk(self.L.sos_ba_create(ctx.h_, C.byref(self.params), C.byref(self.h_)), "sos_ba_create")
        self.n = self.P = self.R = 0

    def close(self):
        if getattr(self, "h_", None):
            self.L.sos_ba_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_window(self, frame_slot, points, resid, res_toZeroF=None, lin_J=None):
        fs = np.ascontiguousarray(frame_slot, dtype=np.int32)
        pts = np.ascontiguousarray(points)
        res = np.ascontiguousarray(resid)
        self.n, self.P, self.R = len(fs), len(pts), len(res)
        rtz = None if res_toZeroF is None else np.ascontiguousarray(res_toZeroF, dtype=np.float32)
        lj = None if lin_J is None else np.ascontiguousarray(lin_J)
        _chk(self.L.sos_ba_set_window(self.h_, self.n, _p(fs), self.P, _p(pts), self.R, _p(res), _p(rtz), _p(lj)),
             "sos_ba_set_window")

    def set_state(self, calib=None, precalc=None, adHTdeltaF=None, cDeltaF=None, adHost=None, adTarget=None,
                  idepth=None, idepth_zero=None, deltaF=None):
        def f32(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.float32)

        def f64(a):
            return None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        pc = None if precalc is None else np.ascontiguousarray(precalc)
        _chk(self.L.sos_ba_set_state(self.h_, C.byref(calib) if calib is not None else None, _p(pc),
                                     _p(f32(adHTdeltaF)), _p(f32(cDeltaF)), _p(f64(adHost)), _p(f64(adTarget)),
                                     _p(f32(idepth)), _p(f32(idepth_zero)), _p(f32(deltaF))), "sos_ba_set_state")

    def linearize(self, frameEnergyTH, outputs=True):
        th = np.ascontiguousarray(frameEnergyTH, dtype=np.float32)
        E = C.c_double(0)
        if outputs:
            ns = np.zeros(self.R, dtype=np.uint8)
            ne = np.zeros(self.R, dtype=np.float32)
            nw = np.zeros(self.R, dtype=np.float32)
            ce = np.zeros((self.R, 3), dtype=np.float32)
            _chk(self.L.sos_ba_linearize(self.h_, _p(th), C.byref(E), _p(ns), _p(ne), _p(nw), _p(ce)),
                 "sos_ba_linearize")
            return dict(energy=E.value, newState=ns, newEnergy=ne, newEnergyWithOutlier=nw, center=ce)
        _chk(self.L.sos_ba_linearize(self.h_, _p(th), C.byref(E), None, None, None, None), "sos_ba_linearize")
        return dict(energy=E.value)

    def apply_res(self):
        _chk(self.L.sos_ba_apply_res(self.h_), "sos_ba_apply_res")

    def reset_oob(self):
        _chk(self.L.sos_ba_reset_oob(self.h_), "sos_ba_reset_oob")

    def fix_linearization(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        _chk(self.L.sos_ba_fix_linearization(self.h_, _p(idx), len(idx)), "sos_ba_fix_linearization")

    def _hb(self):
        dim = 4 + 8 * self.n
        return [np.zeros((dim, dim)), np.zeros(dim), np.zeros((dim, dim)), np.zeros(dim), np.zeros((dim, dim)),
                np.zeros(dim)]

    def accumulate(self):
        out = self._hb()
        ra, rl = C.c_int(0), C.c_int(0)
        _chk(self.L.sos_ba_accumulate(self.h_, *[_p(o) for o in out], C.byref(ra), C.byref(rl)), "sos_ba_accumulate")
        return dict(H_A=out[0], b_A=out[1], H_L=out[2], b_L=out[3], H_sc=out[4], b_sc=out[5], resInA=ra.value,
                    resInL=rl.value)

    def accumulate_local(self):
        _chk(self.L.sos_ba_accumulate_local(self.h_), "sos_ba_accumulate_local")

    def acc_buffer(self):
        ptr, n = C.c_void_p(), C.c_size_t(0)
        _chk(self.L.sos_ba_acc_buffer(self.h_, C.byref(ptr), C.byref(n)), "sos_ba_acc_buffer")
        return ptr.value, n.value

    def stitch(self):
        out = self._hb()
        ra, rl = C.c_int(0), C.c_int(0)
        _chk(self.L.sos_ba_stitch(self.h_, *[_p(o) for o in out], C.byref(ra), C.byref(rl)), "sos_ba_stitch")
        return dict(H_A=out[0], b_A=out[1], H_L=out[2], b_L=out[3], H_sc=out[4], b_sc=out[5], resInA=ra.value,
                    resInL=rl.value)

    def point_hessian(self):
        a = [np.zeros(self.P, dtype=np.float32) for _ in range(3)]
        _chk(self.L.sos_ba_get_point_hessian(self.h_, *[_p(x) for x in a]), "sos_ba_get_point_hessian")
        return dict(idepth_hessian=a[0], HdiF=a[1], bdSumF=a[2])

    def resubstitute(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        step = np.zeros(self.P, dtype=np.float32)
        _chk(self.L.sos_ba_resubstitute(self.h_, _p(x), _p(step)), "sos_ba_resubstitute")
        return step

    def calc_lenergy(self):
        E = C.c_double(0)
        _chk(self.L.sos_ba_calc_lenergy(self.h_, C.byref(E)), "sos_ba_calc_lenergy")
        return E.value

    def accumulate_marg(self, point_idx):
        dim = 4 + 8 * self.n
        idx = np.ascontiguousarray(point_idx, dtype=np.int32)
        out = [np.zeros((dim, dim)), np.zeros(dim), np.zeros((dim, dim)), np.zeros(dim)]
        rm = C.c_int(0)
        _chk(self.L.sos_ba_accumulate_marg(self.h_, _p(idx), len(idx), *[_p(o) for o in out], C.byref(rm)),
             "sos_ba_accumulate_marg")
        return dict(M=out[0], Mb=out[1], Msc=out[2], Mbsc=out[3], resInM=rm.value)

    def update_point_priors(self, point_idx, priorF):
        idx = np.ascontiguousarray(point_idx, dtype=np.int32)
        pr = np.ascontiguousarray(priorF, dtype=np.float32)
        _chk(self.L.sos_ba_update_point_priors(self.h_, _p(idx), _p(pr), len(idx)), "sos_ba_update_point_priors")

    def jacobian(self, r, which=0):
        out = np.zeros(1, dtype=RAWJAC_DTYPE)
        _chk(self.L.sos_ba_get_jacobian(self.h_, int(r), which, _p(out)), "sos_ba_get_jacobian")
        return out[0]

    def residual_flags(self):
        f = np.zeros(self.R, dtype=np.uint32)
        s = np.zeros(self.R, dtype=np.int32)
        e = np.zeros(self.R, dtype=np.float32)
        _chk(self.L.sos_ba_get_residual_flags(self.h_, _p(f), _p(s), _p(e)), "sos_ba_get_residual_flags")
        return f, s, e

    def JpJdF(self):
        o = np.zeros((self.R, 8), dtype=np.float32)
        _chk(self.L.sos_ba_get_JpJdF(self.h_, _p(o)), "sos_ba_get_JpJdF")
        return o

    def res_toZeroF(self):
        o = np.zeros((self.R, 8), dtype=np.float32)
        _chk(self.L.sos_ba_get_res_toZeroF(self.h_, _p(o)), "sos_ba_get_res_toZeroF")
        return o

    def time_kernel(self, name: str, frameEnergyTH, iters: int = 100) -> float:
        th = np.ascontiguousarray(frameEnergyTH, dtype=np.float32)
        ms = C.c_float(0)
        _chk(self.L.sos_ba_time_kernel(self.h_, name.encode(), _p(th), iters, C.byref(ms)), f"sos_ba_time_kernel({name})")
        return ms.value


class Tracker:
    """sos_tracker: device side of one CoarseTracker / ScaleOptimizer."""
    _borrowed = None

    @property
    def _h(self):
        return self._borrowed if self._borrowed is not None else self.h_

    def __init__(self, ctx: Context, params: dict):
        self.L = ctx.L
        self.ctx = ctx
        self.params = Params.from_dict(params)
        self.h_ = C.c_void_p()
        _chk(self.L.sos_tracker_create(ctx.h_, C.byref(self.params), C.byref(self.h_)), "sos_tracker_create")
        self.pc_n = np.zeros(ctx.levels, dtype=np.int32)

    def close(self):
        if getattr(self, "h_", None):
            self.L.sos_tracker_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_ref(self, calib, refSlot, u, v, idepth, hdi):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (u, v, idepth, hdi)]
        _chk(self.L.sos_tracker_set_ref(self._h, C.byref(calib), refSlot, len(a[0]), *[_p(x) for x in a],
                                        _p(self.pc_n)), "sos_tracker_set_ref")
        return self.pc_n.copy()

    def scale_depth(self, s):
        _chk(self.L.sos_tracker_scale_depth(self._h, s), "sos_tracker_scale_depth")

    def get_pc(self, lvl):
        n = int(self.pc_n[lvl])
        out = [np.zeros(n, dtype=np.float32) for _ in range(4)]
        _chk(self.L.sos_tracker_get_pc(self._h, lvl, *[_p(o) for o in out]), "sos_tracker_get_pc")
        return out

    def set_points3d(self, calib, xyz, colors):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        col = np.ascontiguousarray(colors, dtype=np.float32)
        _chk(self.L.sos_tracker_set_points3d(self._h, C.byref(calib), len(xyz), _p(xyz), _p(col)), "sos_tracker_set_points3d")

    def calc_res(self, lvl, newSlot, RKi, t, affLL, cutoff):
        rs = np.zeros(6)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (RKi, t, affLL)]
        _chk(self.L.sos_tracker_calc_res(self._h, lvl, newSlot, *[_p(x) for x in a], cutoff, _p(rs)),
             "sos_tracker_calc_res")
        return rs

    def calc_gs(self, lvl, a, b0):
        H, b = np.zeros((8, 8)), np.zeros(8)
        _chk(self.L.sos_tracker_calc_gs(self._h, lvl, a, b0, _p(H), _p(b)), "sos_tracker_calc_gs")
        return H, b

    def calc_res_scale(self, lvl, stereoSlot, RKi, t, K1, scale, cutoff):
        rs = np.zeros(6)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (RKi, t, K1)]
        _chk(self.L.sos_tracker_calc_res_scale(self._h, lvl, stereoSlot, *[_p(x) for x in a], scale, cutoff, _p(rs)),
             "sos_tracker_calc_res_scale")
        return rs

    def calc_gs_scale(self, lvl, t, K1, scale):
        H, b = C.c_float(0), C.c_float(0)
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (t, K1)]
        _chk(self.L.sos_tracker_calc_gs_scale(self._h, lvl, _p(a[0]), _p(a[1]), scale, C.byref(H), C.byref(b)),
             "sos_tracker_calc_gs_scale")
        return H.value, b.value


class PixelSelector:
    """sos_pixsel: PixelSelector (FS/PixelSelector2.cpp) over the frames of a Context."""

    def __init__(self, ctx: Context, prm, pattern):
        self.L = load()
        self.ctx = ctx
        self.w, self.h = ctx.w, ctx.h
        pattern = np.ascontiguousarray(pattern, dtype=np.uint8)
        assert pattern.size == self.w * self.h
        self.h_ = C.c_void_p()
        _chk(self.L.sos_pixsel_create(ctx.h_, C.byref(prm), _p(pattern), C.byref(self.h_)), "sos_pixsel_create")
        ctx._adopt(self)
        self.current_potential = 3

    def close(self):
        if self.h_:
            # the garbage collector may finalise this object after its context was closed (weak references to objects being
            # collected are dead, so the context could not close it first): the context is gone, the handle is only dropped
            if getattr(self.ctx, "h_", None):
                self.L.sos_pixsel_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def make_hists(self, slot):
        n = (self.w // 32) * (self.h // 32)
        ths, sm = np.zeros(n, np.float32), np.zeros(n, np.float32)
        _chk(self.L.sos_pixsel_make_hists(self.h_, slot, _p(ths), _p(sm)), "sos_pixsel_make_hists")
        return ths, sm

    def select(self, slot, pot, th_factor=1.0, want_map=True):
        m = np.zeros((self.h, self.w), np.float32) if want_map else None
        n = np.zeros(3, np.int32)
        _chk(self.L.sos_pixsel_select(self.h_, slot, pot, th_factor, _p(m), _p(n)), "sos_pixsel_select")
        return m, n

    def make_maps(self, slot, density, recursions_left=1, th_factor=1.0, want_map=True):
        m = np.zeros((self.h, self.w), np.float32) if want_map else None
        pot = C.c_int32(self.current_potential)
        num = C.c_int32(0)
        _chk(self.L.sos_pixsel_make_maps(self.h_, slot, density, recursions_left, th_factor, C.byref(pot), _p(m), C.byref(num)),
             "sos_pixsel_make_maps")
        self.current_potential = pot.value
        return m, num.value

    def list(self, pattern_padding=2, capacity=1 << 16):
        u, v, t = np.zeros(capacity, np.int32), np.zeros(capacity, np.int32), np.zeros(capacity, np.float32)
        cnt = C.c_int32(0)
        _chk(self.L.sos_pixsel_list(self.h_, pattern_padding, capacity, _p(u), _p(v), _p(t), C.byref(cnt)), "sos_pixsel_list")
        k = min(cnt.value, capacity)
        return u[:k], v[:k], t[:k]


def camera_parse(text: str):
    """sos_camera_parse: the 4-line DSO camera file -> records.CameraModel (raises on the formats the reference rejects)."""
    from .records import CameraModel
    m = CameraModel()
    _chk(load().sos_camera_parse(text.encode(), C.byref(m)), "sos_camera_parse")
    return m


class Undistorter:
    """sos_undistort: Undistort + PhotometricUndistorter feeding the pyramid of a Context (created with the output size)."""

    def __init__(self, ctx: Context, cam, G=None, vignette=None, photometric_mode=2):
        self.L, self.ctx, self.cam = load(), ctx, cam
        g = None if G is None else np.ascontiguousarray(G, dtype=np.float32)
        v = None if vignette is None else np.ascontiguousarray(vignette, dtype=np.float32)
        self.h_ = C.c_void_p()
        _chk(self.L.sos_undistort_create(ctx.h_, C.byref(cam), _p(g), 0 if g is None else len(g), _p(v), photometric_mode, C.byref(self.h_)),
             "sos_undistort_create")
        ctx._adopt(self)

    def close(self):
        if self.h_:
            if getattr(self.ctx, "h_", None):   # see PixelSelector.close
                self.L.sos_undistort_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get(self):
        K = np.zeros(4, np.float32)
        rx = np.zeros((self.cam.h, self.cam.w), np.float32)
        ry = np.zeros_like(rx)
        pt = C.c_int32(0)
        _chk(self.L.sos_undistort_get(self.h_, _p(K), _p(rx), _p(ry), C.byref(pt)), "sos_undistort_get")
        return K, rx, ry, bool(pt.value)

    def frame(self, raw, exposure, slot, factor=1.0, gammaB=None, want_image=True):
        raw = np.ascontiguousarray(raw)
        assert raw.dtype in (np.uint8, np.uint16) and raw.shape == (self.cam.hOrg, self.cam.wOrg)
        out = np.zeros((self.cam.h, self.cam.w), np.float32) if want_image else None
        gb = None if gammaB is None else np.ascontiguousarray(gammaB, dtype=np.float32)
        _chk(self.L.sos_undistort_frame(self.h_, _p(raw), raw.dtype.itemsize, exposure, factor, slot, _p(gb), _p(out)), "sos_undistort_frame")
        return out
