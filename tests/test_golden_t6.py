"""The second committed fixture (tests/golden/t6_golden.npz, made by tests/golden/make_golden_t6.py): a T6-size window with
an IN / OOB / OUTLIER mix and ~9 % linearised residuals, the CoarseTracker template and immature-point records on the same
frames -- against the oracle (CPU, every run) and against the HIP path (GPU)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import Calib, IMMATURE_DTYPE, TraceParams
from tests import helpers as hp
from tests.golden import make_golden_t6 as mk

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t6_golden.npz"))


def _window():
    win = mk.make_window()
    assert mk.sha(win.images) == str(G["in_images"]) and mk.sha(win.points) == str(G["in_points"]) and mk.sha(win.resid) == str(G["in_resid"])
    return win


def test_oracle_reproduces_the_t6_fixture():
    win, ow, out = mk.generate()
    assert mk.sha(win.images) == str(G["in_images"])
    for k in G.files:
        if k.startswith("in_"):
            continue
        a, b = out[k], G[k]
        if a.dtype.names:
            for f in a.dtype.names:
                if f != "pad":
                    assert np.array_equal(a[f], b[f], equal_nan=True), (k, f)
        else:
            assert np.array_equal(np.asarray(a), b, equal_nan=True), k
    states = np.bincount(G["new_state"], minlength=3)
    assert states[synth.RES_OOB] > 0 and states[synth.RES_OUTLIER] > 100 and states[synth.RES_IN] > 2000
    assert 0.05 < len(G["lin_idx"]) / win.R < 0.2
    ow.close()


@pytest.mark.gpu
def test_gpu_backend_matches_the_t6_fixture():
    from sos_slam_amd import lib
    win = _window()
    ow = orc.window_from_synth(win)          # host state only (precalc, adjoints)
    ctx, ba = hp.gpu_backend(win, ow)
    th = np.full(win.n, mk.TH, np.float32)
    ba.reset_oob()
    g = ba.linearize(th)
    assert np.array_equal(g["newState"].astype(np.int32), G["new_state"])
    assert np.array_equal(g["newEnergy"], G["new_energy"])
    assert np.array_equal(g["newEnergyWithOutlier"], G["new_energy_wo"])
    ok = G["new_state"] != synth.RES_OOB
    assert np.array_equal(g["center"][ok], G["center"][ok])
    assert abs(g["energy"] - float(G["lin_energy"])) <= 1e-12 * abs(float(G["lin_energy"]))
    for k, r in enumerate(range(0, win.R, mk.J_STRIDE)):
        if ok[r] and k % 5 == 0:
            assert hp.jac_equal(ba.jacobian(r, which=1), G["Jnew_sub"][k]), r
    ba.apply_res()
    ba.accumulate()
    ba.fix_linearization(G["lin_idx"])
    assert np.array_equal(ba.res_toZeroF()[G["lin_idx"]], G["res_toZeroF"])
    f, s, e = ba.residual_flags()
    assert np.array_equal(f & 3, G["res_flags"] & 3) and np.array_equal(s, G["res_state"])
    # the linearised residuals through a re-pack, as after flagPointsForRemoval: frozen J + res_toZeroF travel with the snapshot
    ctx2 = lib.Context(win.w, win.h)
    for i in range(win.n):
        ctx2.make_pyramid(i, win.images[i])
    ba2 = lib.Backend(ctx2, win.params)
    res_now = win.resid.copy()
    res_now["flags"] = G["res_flags"] & 7
    res_now["state_state"] = G["res_state"]
    rtz = np.zeros((win.R, 8), np.float32)
    rtz[G["lin_idx"]] = G["res_toZeroF"]
    linJ = np.zeros(win.R, dtype=synth.RAWJAC_DTYPE)
    linJ[G["lin_idx"]] = G["lin_J"]
    ba2.set_window(np.arange(win.n), win.points, res_now, rtz, linJ)
    hp.push_state(ba2, ow)
    ba2.linearize(th)
    ba2.apply_res()
    a = ba2.accumulate()
    assert a["resInA"] == int(G["resInA"]) and a["resInL"] == int(G["resInL"]) == len(G["lin_idx"])
    for k in ("H_A", "b_A", "H_L", "b_L", "H_sc", "b_sc"):
        assert hp.relerr(a[k], G["acc64_" + k]) < 1e-5, (k, hp.relerr(a[k], G["acc64_" + k]))
    assert np.array_equal(ba2.point_hessian()["idepth_hessian"], G["idepth_hessian"])
    assert abs(ba2.calc_lenergy() - float(G["lenergy"])) <= 1e-5 * abs(float(G["lenergy"]))
    assert np.array_equal(ba2.resubstitute(G["resub_x"]), G["resub_step"])
    for o in (ba2, ctx2, ba, ctx, ow):
        o.close()


@pytest.mark.gpu
def test_gpu_tracker_and_immature_match_the_t6_fixture():
    from sos_slam_amd import lib
    win = _window()
    ctx = lib.Context(win.w, win.h)
    for i in range(win.n):
        ctx.make_pyramid(i, win.images[i])
    new_slot = win.n
    ctx.make_pyramid(new_slot, win.extra_images[0])
    ow = orc.window_from_synth(win)
    calib = Calib.from_K(ow.calib_value_scaled())
    trk = lib.Tracker(ctx, win.params)
    pc_n = trk.set_ref(calib, win.n - 1, G["trk_u"], G["trk_v"], G["trk_id"], G["trk_hdi"])
    levels = len(G["pc_n"])
    assert np.array_equal(pc_n[:levels], G["pc_n"])
    for lvl in range(levels):
        for nm, arr in zip(("u", "v", "idepth", "color"), trk.get_pc(lvl)):
            assert np.array_equal(arr, G[f"pc{lvl}_{nm}"]), (lvl, nm)
    T = G["trk_T"]
    K = ow.calib_value_scaled()
    for lvl in range(levels):
        fx, fy = np.float32(K[0] / 2 ** lvl), np.float32(K[1] / 2 ** lvl)
        cx, cy = np.float32((K[2] + 0.5) / 2 ** lvl - 0.5), np.float32((K[3] + 0.5) / 2 ** lvl - 0.5)
        Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]], dtype=np.float32)
        RKi = (T[:9].reshape(3, 3).astype(np.float32) @ Ki).astype(np.float32)
        r = trk.calc_res(lvl, new_slot, RKi, T[9:].astype(np.float32), np.array([1.01, -0.5], np.float32), 20.0)
        ro = G["trk_res"][lvl]
        assert r[1] == ro[1] and r[5] == pytest.approx(ro[5], rel=1e-6, abs=1e-7)          # counts exact
        assert abs(r[0] - ro[0]) <= 1e-5 * abs(ro[0])                                      # fp32 sums: tree vs running order
        H, b = trk.calc_gs(lvl, 1.01, 0.25)
        assert hp.relerr(H, G["trk_H"][lvl]) < 1e-5 and hp.relerr(b, G["trk_b"][lvl]) < 1e-4
    trk.close()
    # immature points: constructor and one traceOn, bit for bit
    tprm = TraceParams.default()
    rec = ctx.immature_init(tprm, 2, G["imm_u"], G["imm_v"])
    for f in IMMATURE_DTYPE.names:
        if f != "pad":
            assert np.array_equal(rec[f], G["imm_init"][f], equal_nan=True), f
    traced = ctx.immature_trace(tprm, 3, rec, G["imm_KRKi"], G["imm_Kt"], G["imm_aff"])
    for f in IMMATURE_DTYPE.names:
        if f != "pad":
            assert np.array_equal(traced[f], G["imm_traced"][f], equal_nan=True), f
    assert len(np.unique(traced["lastTraceStatus"])) >= 3
    ctx.close()
    ow.close()
