"""Generates tests/golden/t6_golden.npz: a T6-size window (6 KF x 900 points, 320 x 240) with an OOB / outlier mix and
~10 % LINEARISED residuals (frozen Jacobians + res_toZeroF, the state flagPointsForRemoval leaves behind), plus the
CoarseTracker template (pc_* per level, calcRes / calcGSSSE sums at one pose) and immature-point records (constructor +
one traceOn) on the same frames.  Inputs are pinned by hash (the generator is deterministic), outputs are stored.
Like t3_golden.npz it is made BY THE ORACLE (the reference ships no vectors): it pins the C restatement
against accidental change and travels to the GPU box.   python tests/golden/make_golden_t6.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from sos_slam_amd import synth  # noqa: E402
from sos_slam_amd.records import Calib, TraceParams  # noqa: E402

WINDOW = dict(name="T6", noise_sigma=2.0, state_noise=1e-3, idepth_noise=0.02, extra_frames=1)
TH = 350.0
J_STRIDE = 7


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build(win):
    """the oracle-side state every consumer of the fixture rebuilds: returns (ow, out dict)"""
    ow = orc.window_from_synth(win)
    th = np.full(win.n, TH, np.float32)
    ow.reset_oob()
    E = ow.linearize(th)
    out = dict(in_images=sha(win.images), in_points=sha(win.points), in_resid=sha(win.resid), lin_energy=np.float64(E),
               new_state=ow.new_state().copy(), new_energy=ow.new_energy().copy(), new_energy_wo=ow.new_energy_wo().copy(),
               center=ow.center().copy(), Jnew_sub=ow.Jnew()[::J_STRIDE].copy())
    ow.apply_res()
    ow.accumulate()
    res = ow.res()
    # ~10 % of the residuals become linearised: every active residual of a third of host 0's and host 1's points
    pts_sel = np.flatnonzero((win.points["host"] <= 1) & (np.arange(win.P) % 5 < 2))
    ridx = np.flatnonzero(np.isin(res["point"], pts_sel) & ((res["flags"] & 1) != 0)).astype(np.int32)
    ow.fix_linearization(ridx)
    out.update(lin_idx=ridx, res_flags=ow.res()["flags"].copy(), res_state=ow.res()["state_state"].copy(),
               res_toZeroF=ow.res_toZeroF()[ridx].copy(), lin_J=ow.J()[ridx].copy(), JpJdF=ow.JpJdF().copy())
    a32, a64 = ow.accumulate(fp64_truth=False), ow.accumulate(fp64_truth=True)
    for k in ("H_A", "b_A", "H_L", "b_L", "H_sc", "b_sc"):
        out["acc32_" + k], out["acc64_" + k] = a32[k], a64[k]
    out["resInA"], out["resInL"] = np.int32(a32["resInA"]), np.int32(a32["resInL"])
    out["idepth_hessian"] = ow.point_field("idepth_hessian").copy()
    out["lenergy"] = np.float64(ow.calc_lenergy())
    x = np.linspace(-1e-3, 1e-3, 4 + 8 * win.n)
    out["resub_x"], out["resub_step"] = x, ow.resubstitute(x).copy()
    return ow, out, pts_sel


def tracker_part(win, ow, out):
    res = ow.res()
    sel = (res["target"] == win.n - 1) & ((res["flags"] & 3) == 1) & (res["state_state"] == synth.RES_IN)
    c = ow.center()[sel]
    hdi = ow.point_field("HdiF")[res["point"][sel]]
    calib = Calib.from_K(ow.calib_value_scaled())
    ot = orc.OracleTracker(win.params, win.w, win.h)
    pc_n = ot.set_ref(calib, ow.dI[win.n - 1], c[:, 0], c[:, 1], c[:, 2], hdi)
    out.update(trk_u=c[:, 0].copy(), trk_v=c[:, 1].copy(), trk_id=c[:, 2].copy(), trk_hdi=hdi.copy(), pc_n=pc_n.copy())
    for lvl in range(len(pc_n)):
        for nm, arr in zip(("u", "v", "idepth", "color"), ot.get_pc(lvl)):
            out[f"pc{lvl}_{nm}"] = arr
    # one calcRes / calcGSSSE per level at the rendered relative pose
    ref, new = win.frames[win.n - 1]["camToWorld"], win.extra_poses[0]
    T = synth.se3_mul12(synth.se3_inv12(new), ref)
    K = ow.calib_value_scaled()
    new_dI, _ = orc.make_images(win.extra_images[0])
    rs, Hs, bs = [], [], []
    for lvl in range(len(pc_n)):
        fx, fy = np.float32(K[0] / 2 ** lvl), np.float32(K[1] / 2 ** lvl)
        cx, cy = np.float32((K[2] + 0.5) / 2 ** lvl - 0.5), np.float32((K[3] + 0.5) / 2 ** lvl - 0.5)
        Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]], dtype=np.float32)
        RKi = (T[:9].reshape(3, 3).astype(np.float32) @ Ki).astype(np.float32)
        r = ot.calc_res(lvl, new_dI[lvl], RKi, T[9:].astype(np.float32), np.array([1.01, -0.5], np.float32), 20.0)
        H, b = ot.calc_gs(lvl, 1.01, 0.25)
        rs.append(r); Hs.append(H); bs.append(b)
    out.update(trk_T=T, trk_res=np.stack(rs), trk_H=np.stack(Hs), trk_b=np.stack(bs))
    ot.close()


def immature_part(win, ow, out):
    tprm = TraceParams.default()
    uu, vv = np.meshgrid(np.arange(8, win.w - 8, 9), np.arange(8, win.h - 8, 9))
    u, v = uu.reshape(-1).astype(np.int32), vv.reshape(-1).astype(np.int32)
    rec = orc.immature_init(tprm, ow.dI[2][0], u, v)
    K = np.array([[win.K[0], 0, win.K[2]], [0, win.K[1], win.K[3]], [0, 0, 1]], dtype=np.float32)
    T = synth.se3_mul12(synth.se3_inv12(win.frames[3]["camToWorld"]), win.frames[2]["camToWorld"])
    R, t = T[:9].reshape(3, 3).astype(np.float32), T[9:].astype(np.float32)
    KRKi = (K @ R @ np.linalg.inv(K).astype(np.float32)).astype(np.float32).reshape(-1)
    Kt = (K @ t).astype(np.float32)
    aff = np.array([1.0, 0.0], np.float32)
    traced = orc.immature_trace(tprm, ow.dI[3][0], rec, KRKi, Kt, aff)
    out.update(imm_u=u, imm_v=v, imm_init=rec, imm_KRKi=KRKi, imm_Kt=Kt, imm_aff=aff, imm_traced=traced)


def make_window():
    win = synth.make_window(**WINDOW)
    # a few grossly wrong inverse depths: their patterns leave the target images (the OOB branch of linearize)
    for f in ("idepth_scaled", "idepth_zero_scaled"):
        win.points[f][::29] *= np.float32(3.0)
    return win


def generate():
    win = make_window()
    ow, out, _ = build(win)
    tracker_part(win, ow, out)
    immature_part(win, ow, out)
    return win, ow, out


if __name__ == "__main__":
    win, ow, out = generate()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "t6_golden.npz")
    np.savez_compressed(path, **out)
    frac = len(out["lin_idx"]) / win.R
    print("wrote", path, os.path.getsize(path), "bytes; linearised fraction %.3f; states" % frac, np.bincount(out["new_state"], minlength=3))
