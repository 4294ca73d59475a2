"""Self-consistency known-answer tests of the oracle (CPU): the reference pins nothing on this path
(SURVEY.md 4, 8(c)), so the restatement is checked against first principles:

  * finite differences of the photometric residual w.r.t. the relative pose and the inverse depth
  * H_top - H_sc / b_top - b_sc equal the Schur complement of the explicitly assembled dense system
  * resubstitute equals the eliminated block of the dense solve
  * the reference's tiered fp32 accumulators agree with fp64 accumulation to fp32 round-off
"""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import Calib


@pytest.fixture(scope="module")
def t4():
    win = synth.make_window("T4", state_noise=0.0)  # state == state_zero: FEJ point = current point
    ow = orc.window_from_synth(win)
    ow.reset_oob()
    th = np.full(win.n, 1e9, np.float32)  # everything IN
    ow.linearize(th)
    ow.apply_res()
    return win, ow, th


def _resF(ow, th):
    ow.reset_oob()
    ow.linearize(th)
    return np.array(ow.Jnew()["resF"], dtype=np.float64), ow.new_state().copy()


def _center(ow, th):
    ow.reset_oob()
    ow.linearize(th)
    return ow.center().astype(np.float64).copy(), ow.new_state().copy(), np.array(ow.Jnew()["resF"], dtype=np.float64)


def test_fd_idepth(t4):
    """Geometry: d(Ku,Kv)/d idepth = Jpdd (FS/Residuals.cpp:117-120), exact to FD accuracy.  Photometry:
    d resF / d idepth ~ JIdx^T Jpdd with the Huber / gradient weights frozen -- loose, because DSO pairs a
    bilinear intensity with a central-difference gradient image."""
    win, ow, th = t4
    J = ow.J().copy()  # applyRes swapped the fresh Jacobian into EFResidual::J
    pts0 = ow.pts().copy()
    eps = 1e-3
    out = []
    for sgn in (+1, -1):
        idp = (pts0["idepth_scaled"] * (1 + sgn * eps)).astype(np.float32)
        ow.set_state(idepth=idp, idepth_zero=idp)
        out.append(_center(ow, th))
    ow.set_state(idepth=pts0["idepth_scaled"].copy(), idepth_zero=pts0["idepth_zero_scaled"].copy())
    (cp, sp, rp), (cm, sm, rm) = out
    res = ow.res()
    ok = (sp == 0) & (sm == 0)
    did = (2 * eps * pts0["idepth_scaled"][res["point"]]).astype(np.float64)
    fd_geo = (cp[:, :2] - cm[:, :2]) / did[:, None]
    an_geo = J["Jpdd"].astype(np.float64)
    assert np.allclose(fd_geo[ok], an_geo[ok], rtol=2e-3, atol=2e-2)
    fd = (rp - rm) / did[:, None]
    an = J["JIdx"][:, 0, :] * J["Jpdd"][:, 0:1] + J["JIdx"][:, 1, :] * J["Jpdd"][:, 1:2]
    big = np.abs(an) > 20.0
    rel = np.abs(fd - an)[ok[:, None] & big] / np.abs(an)[ok[:, None] & big]
    assert np.median(rel) < 0.3
    assert np.corrcoef(fd[ok[:, None] & big], an[ok[:, None] & big])[0, 1] > 0.95


def test_fd_relative_pose(t4):
    """d(Ku,Kv)/d xi of a left increment on the host->target transform = Jpdxi (FS/Residuals.cpp:145-157)."""
    from tests.test_oracle_math import se3_exp
    win, ow, th = t4
    ow.reset_oob()
    ow.linearize(th)
    J = ow.Jnew().copy()
    pc0 = ow.precalc().copy()
    n = win.n
    K = win.K
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]])
    Ki = np.linalg.inv(Km)
    res = ow.res()
    h, t = 0, 1
    sel = (res["host"] == h) & (res["target"] == t)
    assert sel.sum() > 10
    pidx = h + n * t
    R0 = pc0["PRE_RTll_0"][pidx].reshape(3, 3).astype(np.float64)
    t0 = pc0["PRE_tTll_0"][pidx].astype(np.float64)
    for k in range(6):
        rs = []
        eps = 1e-3
        for sgn in (+1, -1):
            d = np.zeros(6)
            d[k] = sgn * eps
            T = se3_exp(d)
            Rn = T[:9].reshape(3, 3) @ R0
            tn = T[:9].reshape(3, 3) @ t0 + T[9:]
            pc = pc0.copy()
            pc["PRE_RTll_0"][pidx] = Rn.reshape(-1).astype(np.float32)
            pc["PRE_tTll_0"][pidx] = tn.astype(np.float32)
            pc["PRE_KRKiTll"][pidx] = (Km @ Rn @ Ki).reshape(-1).astype(np.float32)
            pc["PRE_KtTll"][pidx] = (Km @ tn).astype(np.float32)
            ow.set_state(precalc=pc)
            rs.append(_center(ow, th))
        ow.set_state(precalc=pc0)
        (cp, sp, _), (cm, sm, _) = rs
        ok = sel & (sp == 0) & (sm == 0)
        fd = (cp[ok, :2] - cm[ok, :2]) / (2 * eps)
        an = J["Jpdxi"][ok, :, k].astype(np.float64)
        assert np.allclose(fd, an, rtol=5e-3, atol=5e-2), k


def _dense_system(win, ow):
    """Explicit normal equations over [calib 4 | poses 8n | idepth P] from the per-residual Jacobians."""
    n, P = win.n, win.P
    J = ow.J()
    res = ow.res()
    adH, adT = ow.adHost(), ow.adTarget()
    dim = 4 + 8 * n
    H = np.zeros((dim + P, dim + P))
    b = np.zeros(dim + P)
    for r in range(len(res)):
        if not (res["flags"][r] & 1):
            continue
        h, t, p = int(res["host"][r]), int(res["target"][r]), int(res["point"][r])
        j = J[r]
        JI = np.stack([j["JIdx"][0], j["JIdx"][1]], axis=1).astype(np.float64)        # 8 x 2
        Jrel = np.concatenate([JI @ np.stack([j["Jpdxi"][0], j["Jpdxi"][1]]).astype(np.float64),
                               np.stack([j["JabF"][0], j["JabF"][1]], axis=1).astype(np.float64)], axis=1)  # 8 x 8
        row = np.zeros((8, dim + P))
        row[:, 0:4] = JI @ np.stack([j["Jpdc"][0], j["Jpdc"][1]]).astype(np.float64)
        k = h + n * t
        row[:, 4 + 8 * h:12 + 8 * h] += Jrel @ adH[k].T
        row[:, 4 + 8 * t:12 + 8 * t] += Jrel @ adT[k].T
        row[:, dim + p] = JI @ j["Jpdd"].astype(np.float64)
        rF = j["resF"].astype(np.float64)
        H += row.T @ row
        b += row.T @ rF
    return H, b


def test_schur_complement_and_resubstitute():
    win = synth.make_window("T3")
    ow = orc.window_from_synth(win)
    ow.reset_oob()
    ow.linearize(np.full(win.n, 1e9, np.float32))
    ow.apply_res()
    acc = ow.accumulate(fp64_truth=True)
    Hd, bd = _dense_system(win, ow)
    dim = 4 + 8 * win.n
    # points without active residuals do not enter (Hdi = 0); the oracle clamps H_dd at 1e-10
    Hdd = np.diag(Hd)[dim:].copy()
    act = Hdd > 0
    Hdi = np.zeros_like(Hdd)
    Hdi[act] = 1.0 / Hdd[act]
    Hpd = Hd[:dim, dim:]
    S_H = Hpd @ np.diag(Hdi) @ Hpd.T
    S_b = Hpd @ (Hdi * bd[dim:])
    Htop = acc["H_A"] + acc["H_L"]
    scale = np.abs(Htop).max()
    # (the per-residual 2x2 shorthands JIdx2 / JabJIdx / Jab2 are fp32 sums: agreement to fp32 round-off)
    assert np.abs(Htop - Hd[:dim, :dim]).max() < 1e-6 * scale
    assert np.abs(acc["b_A"] + acc["b_L"] - bd[:dim]).max() < 1e-6 * np.abs(bd[:dim]).max()
    assert np.abs(acc["H_sc"] - S_H).max() < 1e-6 * scale           # Hdi is an fp32 reciprocal in the oracle
    assert np.abs(acc["b_sc"] - S_b).max() < 1e-6 * np.abs(S_b).max()
    # the reference's tiered fp32 accumulators agree with fp64 accumulation to fp32 round-off
    a32 = ow.accumulate(fp64_truth=False)
    for k in ("H_A", "b_A", "H_sc", "b_sc"):
        assert np.linalg.norm(a32[k] - acc[k]) < 1e-5 * np.linalg.norm(acc[k])
    # resubstitute == eliminated block of the dense solve:  x_d = Hdd^-1 (b_d - H_dp x_p), step = -x_d
    rng = np.random.default_rng(1)
    xp = rng.normal(0, 1e-3, dim)
    step = ow.resubstitute(xp)
    xd = Hdi * (bd[dim:] - Hpd.T @ xp)
    assert np.allclose(step[act], -xd[act], rtol=2e-4, atol=1e-7)


def test_energy_threshold_statistic():
    """setNewFrameEnergyTH (FS/FullSystemOptimize.cpp:84-124): order statistic of the newest frame's energies."""
    win = synth.make_window("T4")
    ow = orc.window_from_synth(win)
    ow.optimize(1)
    th = ow.frame(win.n - 1)["frameEnergyTH"]
    assert th > 0 and np.isfinite(th)
    # older frames keep their threshold
    assert ow.frame(0)["frameEnergyTH"] == pytest.approx(512.0)


@pytest.mark.parametrize("name", ["T4", "T6", "W7"])
def test_energy_threshold_is_the_order_statistic_formula(name):
    """setNewFrameEnergyTH restated in NumPy from the energies of the last linearisation (every residual towards the newest keyframe
    with state_NewEnergyWithOutlier >= 0, those the final linearizeAll(true) then dropped included):
        th = ((26 w + sqrt(e[k]) fac (1 - w)) overall)^2,   k = (int)(frameEnergyTHN * count)
    with the index formed in FLOAT as the reference does (a float setting times the count): 0.7f * 170 = 119.0, where the double product
    118.99999999999999 would truncate to 118 -- T4 (170 residuals) and W7 (1450 -> 1015 vs 1014) sit on such counts."""
    win = synth.make_window(name)
    ow = orc.window_from_synth(win)
    ow.optimize(2)
    res, wo, p = ow.res(), ow.new_energy_wo(), win.params
    e = np.sort(wo[(res["target"] == win.n - 1) & (wo >= 0)])
    k = int(np.float32(p["frameEnergyTHN"]) * np.float32(len(e)))
    th = np.float32(np.sqrt(np.float32(e[k]))) * np.float32(p["frameEnergyTHFacMedian"])
    th = np.float32(np.float32(26.0) * np.float32(p["frameEnergyTHConstWeight"]) + th * np.float32(1 - p["frameEnergyTHConstWeight"]))
    th = np.float32(th * th) * np.float32(p["overallEnergyTHWeight"]) ** 2
    assert ow.frame(win.n - 1)["frameEnergyTH"] == pytest.approx(float(th), rel=1e-6)
    if name in ("T4", "W7"):
        assert int(p["frameEnergyTHN"] * len(e)) == k - 1 and e[k] > e[k - 1]


def test_marginalize_frame_is_dense_schur_complement():
    """marginalizeFrame (OB/EnergyFunctional.cpp:730-889, IMU off) = eliminating the frame's 8 variables (with its
    prior added) from HM / bM; the reference's Jacobi scaling and symmetrisation do not change the result."""
    win = synth.make_window("T4")
    rng = np.random.default_rng(11)
    dim = 4 + 8 * win.n
    A = rng.normal(size=(dim, dim + 5))
    HM = A @ A.T * 1e3 + np.diag(rng.uniform(1, 1e6, dim))
    bM = rng.normal(size=dim) * 1e2
    win.HM, win.bM = HM, bM
    ow = orc.window_from_synth(win)
    for f in (0, 2, win.n - 2):
        Hn, bn = ow.marginalize_frame_prior(f)
        fr = ow.frame(f)
        keep = np.r_[0:4 + 8 * f, 4 + 8 * (f + 1):dim]
        drop = np.r_[4 + 8 * f:4 + 8 * (f + 1)]
        # frame prior of a non-first frame: 0 on the pose, affine priors on a,b (FS/HessianBlocks.h:283-303)
        Hd = HM[np.ix_(drop, drop)].copy()
        prior = ow.frame_prior(f)
        Hd += np.diag(prior[0])
        bd = bM[drop] + prior[0] * prior[1]
        B = HM[np.ix_(keep, drop)]
        H_ref = HM[np.ix_(keep, keep)] - B @ np.linalg.solve(Hd, B.T)
        b_ref = bM[keep] - B @ np.linalg.solve(Hd, bd)
        assert np.abs(Hn - H_ref).max() < 1e-9 * np.abs(H_ref).max()
        assert np.abs(bn - b_ref).max() < 1e-9 * np.abs(b_ref).max()
        assert np.array_equal(Hn, Hn.T)
        assert fr is not None
    ow.close()


def test_marginalize_points_is_additive():
    """marginalizePointsF adds margWeightFac * (M - Msc) per point: marginalising a set in one call or in two
    gives the same prior up to fp32 summation order, and leaves the other points untouched."""
    win = synth.make_window("T4")
    sel = np.flatnonzero(win.points["host"] == 0)

    def run(batches):
        ow = orc.window_from_synth(win)
        ow.optimize(3)
        flags = [ow.marginalize_points(b) for b in batches]
        HM, bM = ow.get_prior()
        removed = (ow.res()["flags"] & 0x100) != 0
        pts_of_removed = np.unique(ow.res()["point"][removed])
        ow.close()
        return HM, bM, np.concatenate(flags), pts_of_removed

    H1, b1, f1, rm1 = run([sel])
    H2, b2, f2, rm2 = run([sel[: len(sel) // 2], sel[len(sel) // 2:]])
    assert np.array_equal(f1, f2) and f1.sum() > 5
    assert np.abs(H1).max() > 0
    assert np.abs(H1 - H2).max() < 1e-5 * np.abs(H1).max()
    assert np.abs(b1 - b2).max() < 1e-5 * max(np.abs(b1).max(), 1e-12)
    observed = np.unique(win.resid["point"])
    assert np.isin(sel[np.isin(sel, observed)], rm1).all()


def test_adjoints_are_the_derivatives_of_the_relative_pose():
    """EnergyFunctional::setAdjointsF (OB/EnergyFunctional.cpp:42-84), pose block: with left increments on camToWorld
    (FS/HessianBlocks.h:228) the host-to-target transform moves by a LEFT increment  delta = Ad(worldToTarget) (xi_host - xi_target),
    so adHost = +Ad^T, adTarget = -Ad^T with the rows scaled by SCALE_XI_TRANS / SCALE_XI_ROT (stored transposed: row = the
    frame's own coordinate).  Checked by central differences of log(T(xi) T(0)^-1) with scipy's expm / logm -- no Sophus, no closed
    form of the adjoint -- on the evaluation points of a window after three iterations."""
    from scipy.linalg import expm, logm
    win = synth.make_window("T6")
    ow = orc.window_from_synth(win)
    ow.optimize(3)
    n = win.n
    E = []
    for f in range(n):
        ev, M = ow.evalpt(f), np.eye(4)
        M[:3, :3], M[:3, 3] = ev[:9].reshape(3, 3), ev[9:]
        E.append(M)
    adH, adT = ow.adHost(), ow.adTarget()
    scale = np.array([synth.SCALE_XI_TRANS] * 3 + [synth.SCALE_XI_ROT] * 3)

    def twist(x):
        X = np.zeros((4, 4))
        X[:3, :3] = [[0, -x[5], x[4]], [x[5], 0, -x[3]], [-x[4], x[3], 0]]
        X[:3, 3] = x[:3]
        return X

    def rel(h, t, sh, st):            # worldToTarget * hostToWorld with the (unscaled) state increments sh, st
        return np.linalg.inv(expm(twist(scale * st)) @ E[t]) @ (expm(twist(scale * sh)) @ E[h])

    def delta(T, T0):
        L = np.real(logm(T @ np.linalg.inv(T0)))
        return np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])

    eps = 1e-6
    worst = 0.0
    for h, t in ((0, 3), (2, 5), (4, 1), (5, 0)):
        T0 = rel(h, t, np.zeros(6), np.zeros(6))
        k = h + n * t
        for which, ad in (("host", adH[k]), ("target", adT[k])):
            fd = np.zeros((6, 6))       # fd[i, j] = d delta_j / d state_i
            for i in range(6):
                e = np.zeros(6)
                e[i] = eps
                p = rel(h, t, e, np.zeros(6)) if which == "host" else rel(h, t, np.zeros(6), e)
                m = rel(h, t, -e, np.zeros(6)) if which == "host" else rel(h, t, np.zeros(6), -e)
                fd[i] = (delta(p, T0) - delta(m, T0)) / (2 * eps)
            worst = max(worst, np.abs(fd - ad[:6, :6]).max() / np.abs(ad[:6, :6]).max())
            assert np.abs(fd - ad[:6, :6]).max() < 1e-6 * np.abs(ad[:6, :6]).max(), (h, t, which)
            assert np.all(ad[:6, 6:] == 0) and np.all(ad[6:, :6] == 0)       # poses and affine parameters do not mix
        assert np.array_equal(adH[k][:6, :6], -adT[k][:6, :6])
    print("adjoints vs central differences:", worst)
    ow.close()
