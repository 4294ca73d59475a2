"""The C restatement of the hot path (oracle/orc_backend.c, fp32) against an independent NumPy fp64 reading of the same
reference functions (oracle/mirror_np.py): FrameFramePrecalc::set, PointFrameResidual::linearize, takeDataF and the
accumulated top / Schur systems restated as dense normal equations.  A misreading of the reference would have to be made
twice, in two different formulations, to get through here.  CPU only."""
import numpy as np
import pytest

from oracle import mirror_np as mir
from oracle import oracle as orc
from sos_slam_amd import synth

CASES = [("T3", {}), ("T4", {}), ("T4", dict(state_noise=3e-3, idepth_noise=0.05, noise_sigma=4.0)), ("T6", {}),
         ("T4", dict(w=154, h=46, P=300))]


def _inputs(win, ow):
    n = win.n
    evalPT = np.stack([ow.evalpt(f) for f in range(n)])
    pre = np.stack([ow.frame(f)["camToWorld"] for f in range(n)])
    st = np.stack([ow.frame(f)["state"] for f in range(n)])
    sz = np.stack([ow.frame(f)["state_zero"] for f in range(n)])
    aff = np.stack([st[:, 6] * synth.SCALE_A, st[:, 7] * synth.SCALE_B], axis=1)
    return evalPT, pre, aff, sz[:, 7] * synth.SCALE_B


@pytest.mark.parametrize("name,kw", CASES)
def test_precalc_linearize_and_system_agree_with_the_numpy_mirror(name, kw):
    win = synth.make_window(name, **kw)
    ow = orc.window_from_synth(win)
    n = win.n
    K = ow.calib_value_scaled()
    evalPT, pre, aff, b0 = _inputs(win, ow)
    # ---- FrameFramePrecalc::set
    pm = mir.precalc(evalPT, pre, K, np.ones(n), aff, b0)
    pc = ow.precalc()
    offd = np.array([h != t for t in range(n) for h in range(n)])
    for key, field, shape in (("KRKi", "PRE_KRKiTll", (3, 3)), ("Kt", "PRE_KtTll", (3,)), ("R0", "PRE_RTll_0", (3, 3)),
                              ("t0", "PRE_tTll_0", (3,)), ("aff", "PRE_aff_mode", (2,))):
        a, b = pm[key], pc[field].reshape((n * n,) + shape).astype(np.float64)
        # K R K^-1 and K t are float products of entries up to fx, cx: their round-off scales with the focal length
        tol = 2e-6 * (float(np.max(K)) if key in ("KRKi", "Kt") else max(np.abs(b).max(), 1.0))
        assert np.abs(a - b)[offd].max() <= tol, (key, np.abs(a - b)[offd].max(), tol)
    assert np.abs(pm["b0"] - pc["PRE_b0_mode"]).max() <= 1e-6 * max(np.abs(pm["b0"]).max(), 1.0)
    # ---- linearize (the mirror gets the oracle's fp32 precalc records, so only the linearisation itself is compared)
    pcf = dict(KRKi=pc["PRE_KRKiTll"].reshape(-1, 3, 3), Kt=pc["PRE_KtTll"], R0=pc["PRE_RTll_0"].reshape(-1, 3, 3), t0=pc["PRE_tTll_0"],
               aff=pc["PRE_aff_mode"], b0=pc["PRE_b0_mode"])
    th = np.array([ow.frame(f)["frameEnergyTH"] for f in range(n)], np.float32)
    if kw:
        th[:] = 300.0          # bring the outlier threshold into the energy distribution of the perturbed windows
    ow.reset_oob()
    ow.linearize(th)
    pts, res = ow.pts().copy(), ow.res().copy()
    m = mir.linearize(pts, res, pcf, K, [d[0] for d in ow.dI], th, win.params)
    ns, en, wo = ow.new_state(), ow.new_energy(), ow.new_energy_wo()
    thmax = np.maximum(th[res["host"]], th[res["target"]])
    near = (np.abs(m["energy_wo"] - thmax) <= 1e-3 * thmax) | (m["energy_wo"] < 0)
    same = m["new_state"] == ns
    assert same[~near].all(), np.flatnonzero(~same & ~near)[:10]
    assert (~same).sum() <= 2
    ok = same & (ns != mir.RES_OOB)
    assert ok.sum() > 0.5 * len(res)
    assert np.abs(m["energy_wo"][ok] - wo[ok]).max() <= 2e-4 * np.abs(wo[ok]).max()
    assert np.abs(m["energy"][ok] - en[ok]).max() <= 2e-4 * np.abs(en[ok]).max()
    cen = ow.center()
    cok = same & m["center_ok"]
    assert np.abs(m["center"][cok] - cen[cok]).max() <= 1e-4 * np.abs(cen[cok]).max()
    Jn = ow.Jnew()
    for f in ("resF", "Jpdxi", "Jpdc", "Jpdd", "JIdx", "JabF"):
        a, b = m["J"][f][ok], Jn[f][ok].astype(np.float64)
        assert np.abs(a - b).max() <= 3e-4 * np.abs(b).max(), (f, np.abs(a - b).max(), np.abs(b).max())
    for f in ("JIdx2", "JabJIdx", "Jab2"):
        a, b = m["J"][f][ok].reshape(-1, 4), Jn[f][ok].astype(np.float64)
        assert np.abs(a - b).max() <= 3e-4 * np.abs(b).max(), f
    if kw.get("state_noise"):
        assert (ns == mir.RES_OUTLIER).sum() > 3        # the perturbed window exercises the outlier branch
    # ---- applyRes + takeDataF, then the accumulated system against the dense normal equations
    ow.apply_res()
    act = (ow.res()["flags"] & 1) != 0
    assert np.array_equal(act, (ns == mir.RES_IN) & (res["state_state"] != mir.RES_OOB))
    both = act & same
    jp = ow.JpJdF()
    assert np.abs(m["JpJdF"][both] - jp[both]).max() <= 5e-4 * np.abs(jp[both]).max()
    # the mirror's own Jacobians (fp64) of exactly the oracle's active set
    t = ow.accumulate(fp64_truth=True)
    d = mir.dense_system(m["J"], both, res, pts, n, ow.adHost(), ow.adTarget(), np.zeros(4))
    if (act & ~same).sum() == 0:
        for k in ("H_A", "H_sc"):
            assert np.linalg.norm(d[k] - t[k]) <= 2e-4 * np.linalg.norm(t[k]), (k, np.linalg.norm(d[k] - t[k]) / np.linalg.norm(t[k]))
        for k in ("b_A", "b_sc"):
            assert np.linalg.norm(d[k] - t[k]) <= 2e-3 * np.linalg.norm(t[k]), (k, np.linalg.norm(d[k] - t[k]) / np.linalg.norm(t[k]))
        idh = ow.point_field("idepth_hessian").astype(np.float64)
        assert np.abs(d["idepth_hessian"] - idh).max() <= 5e-4 * idh.max()
        hdi = ow.point_field("HdiF").astype(np.float64)
        assert np.abs(d["HdiF"] - hdi).max() <= 5e-4 * hdi.max()
        # the reduced system both readings would solve: same Gauss-Newton direction
        dim = 4 + 8 * n
        reg = np.eye(dim) * 1e-3 * np.abs(np.diag(t["H_A"])).max()
        xo = np.linalg.solve(t["H_A"] - t["H_sc"] + win.HM + reg, t["b_A"] - t["b_sc"])
        xm = np.linalg.solve(d["H_A"] - d["H_sc"] + win.HM + reg, d["b_A"] - d["b_sc"])
        assert np.linalg.norm(xo - xm) <= 5e-3 * np.linalg.norm(xo)
        # ---- resubstituteF_MT: the point steps of that direction, relative-coordinate bookkeeping against the dense second block row
        step_o = ow.resubstitute(xo).astype(np.float64)
        step_m = mir.resubstitute(d, xo)
        assert np.array_equal(step_o == 0, step_m == 0) or (np.abs(step_o[step_m == 0]).max() == 0)
        print(name, "resubstitute: max |oracle - dense| / max |step| =", np.abs(step_o - step_m).max() / np.abs(step_m).max())
        assert np.abs(step_o - step_m).max() <= 3e-4 * np.abs(step_m).max(), (np.abs(step_o - step_m).max(), np.abs(step_m).max())
        assert np.abs(step_m).max() > 0
    ow.close()


@pytest.mark.parametrize("name,kw", [("T6", {}), ("W7", {}), ("T6", dict(w=154, h=46))])
def test_tracker_residual_and_system_agree_with_the_numpy_mirror(name, kw):
    """CoarseTracker::calcResPose / calcGSSSEPose (FS/CoarseTracker.cpp:554-764): the C restatement (orc_tracker.c; per-pixel fp32,
    sums in fp64 -- its truth mode) against the vectorised fp64 reading in oracle/mirror_np.py, on every pyramid level, at the
    true relative pose and at a perturbed one (pixels near the image border / the cutoff change sides between fp32 and fp64:
    counts may differ by a few, sums by their share)."""
    from sos_slam_amd.records import Calib
    from sos_slam_amd.synth import se3_exp12, se3_mul12
    win = synth.make_window(name, extra_frames=1, **kw)
    ow = orc.window_from_synth(win)
    ow.optimize(3)
    res = ow.res()
    sel = (res["target"] == win.n - 1) & ((res["flags"] & 0x101) == 1) & (res["state_state"] == synth.RES_IN)
    c = ow.center()[sel]
    hdi = ow.point_field("HdiF")[res["point"][sel]]
    K0 = ow.calib_value_scaled()
    t = orc.OracleTracker(win.params, win.w, win.h)
    t.set_truth_mode(True)
    t.set_ref(Calib.from_K(K0), ow.dI[win.n - 1], c[:, 0], c[:, 1], c[:, 2], hdi)
    new_dI, _ = orc.make_images(win.extra_images[0])
    ref, new = win.frames[win.n - 1]["camToWorld"], win.extra_poses[0]
    Rr, tr, Rn, tn = ref[:9].reshape(3, 3), ref[9:], new[:9].reshape(3, 3), new[9:]
    T0 = np.concatenate([(Rn.T @ Rr).reshape(-1), Rn.T @ (tr - tn)])
    huber, cutoff = float(win.params["huberTH"]), float(win.params["coarseCutoffTH"])
    for T in (T0, se3_mul12(se3_exp12(np.array([0.01, -0.008, 0.004, 0.004, -0.003, 0.002])), T0)):
        for lvl in range(len(t.pc_n)):
            if t.pc_n[lvl] == 0:
                continue
            fx, fy = np.float32(K0[0] / 2 ** lvl), np.float32(K0[1] / 2 ** lvl)
            cx, cy = np.float32((K0[2] + 0.5) / 2 ** lvl - 0.5), np.float32((K0[3] + 0.5) / 2 ** lvl - 0.5)
            Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]], dtype=np.float32)
            RKi = (T[:9].reshape(3, 3).astype(np.float32) @ Ki).astype(np.float32)
            aff = np.array([1.02, -0.7], np.float32)
            rs = t.calc_res(lvl, new_dI[lvl], RKi, T[9:].astype(np.float32), aff, cutoff)
            H, b = t.calc_gs(lvl, float(aff[0]), 0.3)
            pu, pv, pid, pcol = t.get_pc(lvl)
            m = mir.tracker_res_gs(pu, pv, pid, pcol, new_dI[lvl], (fx, fy, cx, cy), T[:9], T[9:], aff.astype(np.float64), 0.3, huber, cutoff,
                                   lvl == 0)
            nE = int(rs[1])
            assert abs(m["numTermsInE"] - nE) <= 2 and nE > 0.5 * t.pc_n[lvl], (lvl, m["numTermsInE"], nE)
            assert abs(m["numSaturated"] - round(rs[5] * nE)) <= 2
            share = 3.0 * (2 * huber * cutoff) / max(rs[0], 1.0)          # what three border / cutoff pixels can move
            assert abs(m["E"] - rs[0]) <= (2e-5 + share) * rs[0], (lvl, m["E"], rs[0])
            if lvl == 0:
                assert m["flowT"] == pytest.approx(rs[2], rel=1e-4) and m["flowRT"] == pytest.approx(rs[4], rel=1e-4)
            # the oracle returns H, b with the SCALE_* factors applied (FS/CoarseTracker.cpp:597-609): undo them
            sc = np.array([1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 10.0, 1000.0])
            Hm, bm = m["H"] * np.outer(sc, sc), m["b"] * sc
            d = np.sqrt(np.abs(np.diag(H)))
            assert np.abs((Hm - H) / np.outer(d, d)).max() <= 2e-4 + share, (lvl, np.abs((Hm - H) / np.outer(d, d)).max())
            assert np.abs((bm - b) / d).max() <= (2e-4 + share) * max(np.abs(b / d).max(), 1e-3) + 1e-6


@pytest.mark.parametrize("idx", [0, 2, 4])
def test_marginalize_frame_agrees_with_the_numpy_mirror(idx):
    """EnergyFunctional::marginalizeFrame (visual part, OB/EnergyFunctional.cpp:788-859): the C restatement against a NumPy reading
    written as index selection + scaled Schur complement (the reference moves blocks in place)."""
    win = synth.make_window("T6")
    ow = orc.window_from_synth(win)
    ow.optimize(3)
    alive = np.flatnonzero(win.points["host"] == 1)[:40].astype(np.int32)
    ow.marginalize_points(alive)                      # a prior with real structure: the window's own + marginalised points
    H0, b0 = ow.get_prior()
    pr, dp = ow.frame_prior(idx)
    Ho, bo = ow.marginalize_frame_prior(idx)
    Hm, bm = mir.marginalize_frame(H0, b0, idx, pr, dp)
    s = 1.0 / np.sqrt(np.abs(np.diag(Hm)) + 10)
    assert np.abs((Ho - Hm) * np.outer(s, s)).max() <= 1e-9 * max(np.abs(Hm * np.outer(s, s)).max(), 1.0)
    assert np.abs((bo - bm) * s).max() <= 1e-9 * max(np.abs(bm * s).max(), 1.0)
    assert np.abs(H0).max() > 0


def test_solve_system_agrees_with_the_numpy_mirror():
    """solveSystemF (IMU off) of the oracle against oracle/mirror_np.solve_system on every Gauss-Newton iteration of optimize() at T6
    and W7, and on every solve of a rolling chain (priors out of real marginalisations)."""
    import ctypes as C
    from sos_slam_amd import synth
    from tests import rolling
    Lo = orc.lib()
    vp = C.c_void_p
    TAP = C.CFUNCTYPE(None, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp)
    Lo.orc_set_solve_tap.argtypes = [TAP]
    errs = []

    def arr(p, shape):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape).copy()

    def tap(n, H, b, Hsc, bsc, HM, bM, delta, lam, x):
        d = 4 + 8 * n
        xm = mir.solve_system(arr(H, (d, d)), arr(b, (d,)), arr(Hsc, (d, d)), arr(bsc, (d,)), arr(HM, (d, d)), arr(bM, (d,)), arr(delta, (d,)), lam)
        xo = arr(x, (d,))
        errs.append(float(np.abs(xo - xm).max() / np.abs(xm).max()))

    cb = TAP(tap)
    Lo.orc_set_solve_tap(cb)
    try:
        for name in ("T6", "W7"):
            ow = orc.window_from_synth(synth.make_window(name))
            ow.optimize(6)
        n_win = len(errs)
        sc = rolling.Scenario(n_frames=14)
        ch = rolling.OracleChain(sc)
        ch.bootstrap()
        while ch.next_frame < sc.n_frames:
            ch.step()
    finally:
        Lo.orc_set_solve_tap(C.cast(None, TAP))
    e = np.array(errs)
    print(f"{n_win} window solves, {len(e) - n_win} chain solves: max {e.max():.1e}, median {np.median(e):.1e}")
    assert n_win >= 4 and len(e) - n_win >= 20
    assert e.max() < 1e-8 and np.median(e) < 1e-10


@pytest.mark.parametrize("name", ["T6", "W7"])
def test_set_state_agrees_with_the_matrix_exponential(name):
    """FrameHessian::setState (FS/HessianBlocks.h:217-230): PRE_camToWorld = SE3::exp(c2w_leftEps) * camToWorld_evalPT with
    c2w_leftEps = (SCALE_XI_TRANS * state[0:3], SCALE_XI_ROT * state[3:6]) -- the oracle's Sophus restatement (closed-form exp, left
    increment on camToWorld) against scipy's matrix exponential of the 4 x 4 twist, on the states three Gauss-Newton iterations leave."""
    from scipy.linalg import expm
    win = synth.make_window(name)
    ow = orc.window_from_synth(win)
    ow.optimize(3)
    moved = 0
    for f in range(win.n):
        ev, fr = ow.evalpt(f), ow.frame(f)
        E, T = np.eye(4), np.eye(4)
        E[:3, :3], E[:3, 3] = ev[:9].reshape(3, 3), ev[9:]
        T[:3, :3], T[:3, 3] = fr["camToWorld"][:9].reshape(3, 3), fr["camToWorld"][9:]
        v, w = synth.SCALE_XI_TRANS * fr["state"][:3], synth.SCALE_XI_ROT * fr["state"][3:6]
        X = np.zeros((4, 4))
        X[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
        X[:3, 3] = v
        assert np.abs(expm(X) @ E - T).max() < 1e-13
        moved += int(np.abs(fr["state"][:6]).max() > 1e-5)
        assert np.abs(T[:3, :3] @ T[:3, :3].T - np.eye(3)).max() < 1e-13
    assert moved >= win.n - 2        # (the first keyframe carries the gauge, the newest one had its evaluation point reset)
    ow.close()


def test_fix_linearization_is_the_jacobian_times_the_absolute_deltas():
    """EFResidual::fixLinearizationF (OB/EnergyFunctionalStructs.cpp:75-103) with EnergyFunctional::setDeltaF's adHTdeltaF
    (OB/EnergyFunctional.cpp:163-194):  resF - res_toZeroF  must be the residual's Jacobian row in ABSOLUTE coordinates
    (calibration | host frame | target frame | inverse depth -- the rows mirror_np.dense_system builds) times the absolute deltas
    (value - value_zero, state - state_zero of the two frames, idepth - idepth_zero).  fp64 matrix products against the oracle's
    float bookkeeping in relative coordinates."""
    win = synth.make_window("T6")
    ow = orc.window_from_synth(win)
    ow.optimize(3)
    n = win.n
    res, pts = ow.res().copy(), ow.pts().copy()
    J = ow.J().copy()
    act = np.flatnonzero((res["flags"] & 1) != 0)[::3][:400]
    v, vz = ow.calib_value()
    cdelta = (v - vz).astype(np.float32).astype(np.float64)
    dstate = np.stack([ow.frame(f)["state"][:8] - ow.frame(f)["state_zero"][:8] for f in range(n)])
    assert np.abs(dstate).max() > 1e-5
    adH, adT = ow.adHost(), ow.adTarget()
    # setDeltaF: adHTdeltaF[h + n t] = delta_h^T adHost + delta_t^T adTarget
    adHT = ow.adHTdeltaF().astype(np.float64)
    for h in range(n):
        for t in range(n):
            k = h + n * t
            want = dstate[h].astype(np.float32).astype(np.float64) @ adH[k] + dstate[t].astype(np.float32).astype(np.float64) @ adT[k]
            assert np.abs(adHT[k] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-6), (h, t)
    resF = J["resF"][act].astype(np.float64)
    ow.fix_linearization(act)
    rtz = ow.res_toZeroF()[act].astype(np.float64)
    worst = 0.0
    for j, r in enumerate(act):
        h, t, p = int(res["host"][r]), int(res["target"][r]), int(res["point"][r])
        k = h + n * t
        JI = J["JIdx"][r].astype(np.float64)                              # (2, 8)
        rel8 = np.concatenate([JI.T @ J["Jpdxi"][r].astype(np.float64), J["JabF"][r].astype(np.float64).T], axis=1)     # 8 pixels x 8 relative
        row_c = JI.T @ J["Jpdc"][r].astype(np.float64)
        row_h, row_t = rel8 @ adH[k].T, rel8 @ adT[k].T
        row_d = JI.T @ J["Jpdd"][r].astype(np.float64)
        lin = row_c @ cdelta + row_h @ dstate[h] + row_t @ dstate[t] + row_d * float(pts["deltaF"][p])
        got = resF[j] - rtz[j]
        worst = max(worst, np.abs(got - lin).max() / max(np.abs(lin).max(), 1e-3))
    print("fixLinearizationF vs J_abs * delta_abs: worst relative difference", worst)
    assert worst < 2e-3
    assert ((ow.res()["flags"][act] & synth.RF_LINEARIZED) != 0).all() if hasattr(synth, "RF_LINEARIZED") else True
    ow.close()
