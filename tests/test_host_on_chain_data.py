"""The facade's host functions that need no device, on the data of a RUNNING chain instead of synthetic inputs: every candidate
selection of activatePointsMT and every IMU-form frame marginalisation of the oracle's rolling visual-inertial chain is handed to
the facade's implementation as well (sosf_activate_select with its bitmap distance transform, sosf_imu_marginalize_frame with its
blocked Schur update).  Priors out of real marginalisations span 1e8 .. 1e-19; candidate sets carry the trace states real traces
leave.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import host
from tests import rolling


def test_selection_and_imu_marginalisation_on_the_data_of_a_chain(monkeypatch):
    seen = dict(select=0, optimize=0, marg=0, worst_H=0.0, worst_b=0.0)
    orc_select = orc.activate_select

    def select_both(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged):
        d_o, D_o = orc_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged)
        d_f, D_f = host.activate_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged)
        assert np.array_equal(d_o, d_f), (seen["select"], int((d_o != d_f).sum()))
        assert np.array_equal(D_o, D_f)
        seen["select"] += 1
        seen["optimize"] += int((d_o == 1).sum())
        return d_o, D_o

    monkeypatch.setattr(rolling.orc, "activate_select", select_both)
    orc_imu = orc.imu
    fac = host.imu()

    class Both:
        def __init__(self):
            self.o = orc_imu()

        def __getattr__(self, name):
            return getattr(self.o, name)

        def marginalize_frame(self, S, cal, recs, idx, delta, pr, dp, HM, bM, marg_weight=0.25):
            Ho, bo = self.o.marginalize_frame(S, cal, recs, idx, delta, pr, dp, HM, bM, marg_weight=marg_weight)
            Hf, bf = fac.marginalize_frame(S, cal, recs, idx, delta, pr, dp, HM, bM, marg_weight=marg_weight)
            seen["marg"] += 1
            seen["worst_H"] = max(seen["worst_H"], np.abs(Ho - Hf).max() / np.abs(Ho).max())
            seen["worst_b"] = max(seen["worst_b"], np.abs(bo - bf).max() / np.abs(bo).max())
            assert np.array_equal(Hf, Hf.T)
            return Ho, bo

    monkeypatch.setattr(rolling.orc, "imu", Both)
    sc = rolling.Scenario(n_frames=20, vio=True)
    ch = rolling.OracleChain(sc)
    ch.bootstrap()
    while ch.next_frame < sc.n_frames:
        ch.step()
    print(seen)
    assert seen["select"] >= 14 and seen["optimize"] > 500 and seen["marg"] >= 6
    assert seen["worst_H"] < 1e-9 and seen["worst_b"] < 1e-9


def test_vio_front_end_on_the_data_of_a_chain():
    """propagateImuState / initializeImu / updateVel / tryTrapScale (FS/HessianBlocks.cpp:225-429) of the facade on every call the
    oracle's chain makes (NumPy front-end, oracle/imu_frontend.py): sample lists as the hand-over on frame marginalisation leaves
    them, biases and spline states as the optimisation leaves them."""
    import copy
    worst = dict(propagate=0.0, initialize=0.0, update_vel=0.0, trap=0.0)
    calls = dict(propagate=0, initialize=0, update_vel=0, trap=0)

    def rel(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))

    class Both(rolling.OracleChain):
        host = host
        _fe = rolling.DeviceChain._fe
        _shell = rolling.DeviceChain._shell

        def _snap(self):
            return copy.deepcopy((self.imu_state, self.imu_zero, {f: s["vel"].copy() for f, s in self.shells.items()}, dict(self.cal),
                                  np.array(self.scale_queue), self.scale_qi))

        def _restore(self, s):
            self.imu_state, self.imu_zero = copy.deepcopy(s[0]), copy.deepcopy(s[1])
            for f, v in s[2].items():
                self.shells[f]["vel"] = v.copy()
            self.cal = dict(s[3])
            self.scale_queue, self.scale_qi = np.array(s[4]), s[5]

        def _both(self, name, what, *a):
            before = self._snap()
            r_f = getattr(rolling.DeviceChain, name)(self, *a)
            after_f = self._snap()
            self._restore(before)
            r_o = getattr(rolling.OracleChain, name)(self, *a)
            after_o = self._snap()
            calls[what] += 1
            for f in after_o[0]:
                worst[what] = max(worst[what], rel(after_f[0][f], after_o[0][f]), rel(after_f[1][f], after_o[1][f]), rel(after_f[2][f], after_o[2][f]))
            worst[what] = max(worst[what], rel([after_f[3]["scale"], after_f[3]["scale_zero"]], [after_o[3]["scale"], after_o[3]["scale_zero"]]))
            assert after_f[3]["trapped"] == after_o[3]["trapped"] and after_f[5] == after_o[5]
            worst[what] = max(worst[what], rel(after_f[4], after_o[4]))
            assert r_f == r_o or (r_f is None and r_o is None)
            return r_o

        def vio_propagate(self, *a): return self._both("vio_propagate", "propagate", *a)
        def vio_initialize(self, *a): return self._both("vio_initialize", "initialize", *a)
        def vio_update_vel(self, *a): return self._both("vio_update_vel", "update_vel", *a)
        def vio_try_trap(self, *a): return self._both("vio_try_trap", "trap", *a)

    sc = rolling.Scenario(n_frames=24, vio=True)
    ch = Both(sc)
    ch.bootstrap()
    while ch.next_frame < sc.n_frames:
        ch.step()
    print(calls, worst)
    assert calls["propagate"] >= 14 and calls["initialize"] == 1 and calls["update_vel"] >= 18 and calls["trap"] >= 5
    assert max(worst.values()) < 1e-9, worst


def test_imu_solve_against_an_independent_kkt_in_numpy():
    """A third reading of the IMU branch of solveSystemF (OB/EnergyFunctional.cpp:1053-1148), independent of the two C implementations:
    the KKT system assembled in NumPy from the pieces (getImuHessian, expandHbtoFitImu, prior around the expanded delta, (1 + lambda)
    on the diagonal, H_sc / (1 + lambda), constraint rows, removal of the unconstrained states) and solved in extended precision, on
    every system the oracle's rolling chain solves (lower triangles mirrored: what Eigen's ldlt() reads).
    Two things it pins that a synthetic scene does not: the H_M d2 term of the right-hand side with d2's IMU part (scale trapped), and
    `1.0f / (1 + lambda)` being a DOUBLE quotient (a float literal over a double sum) -- taken as a float quotient it is 1.3e-8 off,
    which moves the step by 2e-4: the solve amplifies perturbations of H_sc by 1e4."""
    import ctypes as C
    from sos_slam_amd.records import ImuCalib, ImuFrame, ImuSettings, imu_dim
    oapi, fac = orc.imu(), host.imu()
    Lo = orc.lib()
    vp = C.c_void_p
    TAP = C.CFUNCTYPE(None, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp, C.c_double, vp)
    Lo.orc_set_imu_solve_tap.argtypes = [TAP]
    LD = np.longdouble

    def arr(p, shape):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape).copy()

    def low(A):
        return np.tril(A) + np.tril(A, -1).T

    def kkt(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta, lam):
        n = len(frames)
        dI, d0 = imu_dim(n), 4 + 8 * n
        H, b, J, r, sv = oapi.hessian(S, cal, frames)
        He, be = oapi.expand(n, H_top, b_top)
        Hs, bs = oapi.expand(n, H_sc, b_sc)
        d2 = np.zeros(dI)
        d2[:4] = delta[:4]
        if cal.scale_trapped:
            d2[4] = cal.scale - cal.scale_zero
        for i in range(n):
            d2[5 + 29 * i:5 + 29 * i + 8] = delta[4 + 8 * i:12 + 8 * i]
            if cal.scale_trapped:
                d2[5 + 29 * i + 8:5 + 29 * (i + 1)] = np.array(frames[i].state_imu[:]) - np.array(frames[i].state_imu_zero[:])
        Hf = (He + H + HM).astype(LD)
        bf = (be + b + bM).astype(LD) + HM.astype(LD) @ d2.astype(LD)
        Hf[np.diag_indices(dI)] *= LD(1 + lam)
        Hf -= Hs.astype(LD) * LD(1.0 / (1 + lam))
        bf -= bs
        keep = [k for k in range(dI) if k < 4 or (k == 4 and not S.enable_scale_opt) or (k >= 5 and (k - 5) % 29 < (29 if sv[(k - 5) // 29] else 14))]
        ms, c = len(keep), len(r)
        K = np.zeros((ms + c, ms + c), LD)
        K[:ms, :ms] = Hf[np.ix_(keep, keep)]
        K[:ms, ms:] = J[:, keep].T
        K[ms:, :ms] = J[:, keep]
        rhs = np.concatenate([bf[keep], r.astype(LD)])
        nz = np.abs(K).sum(1) > 0                      # all-zero constraint rows (velocity rows without a valid successor)
        K2, r2 = K[np.ix_(nz, nz)], rhs[nz]
        s = 1 / np.sqrt(np.abs(np.diag(K2)).astype(np.float64) + 10)
        Ks = K2 * s[:, None] * s[None, :]
        x = np.linalg.solve(Ks.astype(np.float64), (r2 * s).astype(np.float64)).astype(LD)
        for _ in range(6):                               # iterative refinement with extended-precision residuals
            x = x + np.linalg.solve(Ks.astype(np.float64), (r2 * s - Ks @ x).astype(np.float64)).astype(LD)
        assert float(np.abs(r2 * s - Ks @ x).max()) < 1e-14 * max(float(np.abs(r2 * s).max()), 1.0)
        full = np.zeros(ms + c, LD)
        full[nz] = x * s
        xs, imu_step, scale_step = np.zeros(d0), np.zeros((n, 21)), 0.0
        for j, k in enumerate(keep):
            if k < 4:
                xs[k] = full[j]
            elif k == 4:
                scale_step = -float(full[j])
            elif (k - 5) % 29 < 8:
                xs[4 + 8 * ((k - 5) // 29) + (k - 5) % 29] = full[j]
            else:
                imu_step[(k - 5) // 29, (k - 5) % 29 - 8] = -float(full[j])
        return xs, scale_step, imu_step

    rows = []

    def tap(S, Cal, n, F, H, b, Hsc, bsc, HM, bM, delta, lam, x, scale_step, step_imu):
        d0, dI = 4 + 8 * n, imu_dim(n)
        cal = C.cast(Cal, C.POINTER(ImuCalib)).contents
        Sx = C.cast(S, C.POINTER(ImuSettings)).contents
        frames = list((ImuFrame * n).from_address(F))
        args = [low(arr(H, (d0, d0))), arr(b, (d0,)), low(arr(Hsc, (d0, d0))), arr(bsc, (d0,)), low(arr(HM, (dI, dI))), arr(bM, (dI,)), arr(delta, (d0,))]
        xt, st, it = kkt(Sx, cal, frames, *args, lam)
        xo, so, io = oapi.solve(Sx, cal, frames, *args, lam=lam)
        xf, sf, if_ = fac.solve(Sx, cal, frames, *args, lam=lam)
        sc = np.abs(xt).max()
        si = max(np.abs(it).max(), 1e-300)
        rows.append((np.abs(xo - xt).max() / sc, np.abs(xf - xt).max() / sc, np.abs(io - it).max() / si, np.abs(if_ - it).max() / si,
                     abs(so - st) / max(abs(st), 1e-12), abs(sf - st) / max(abs(st), 1e-12), int(cal.scale_trapped)))

    cb = TAP(tap)
    Lo.orc_set_imu_solve_tap(cb)
    try:
        sc = rolling.Scenario(n_frames=28, vio=True)
        ch = rolling.OracleChain(sc)
        ch.bootstrap()
        while ch.next_frame < sc.n_frames:
            ch.step()
    finally:
        Lo.orc_set_imu_solve_tap(C.cast(None, TAP))
    r = np.array(rows)
    print("solves %d (trapped %d): oracle vs NumPy KKT x max %.1e median %.1e, imu %.1e, scale %.1e | facade x max %.1e median %.1e, imu %.1e, scale %.1e"
          % (len(r), int(r[:, 6].sum()), r[:, 0].max(), np.median(r[:, 0]), r[:, 2].max(), r[:, 4].max(), r[:, 1].max(), np.median(r[:, 1]),
             r[:, 3].max(), r[:, 5].max()))
    assert len(r) >= 60 and r[:, 6].sum() >= 20
    assert np.median(r[:, 0]) < 1e-9 and np.median(r[:, 1]) < 1e-9
    # the oracle's elimination pivots like Eigen's (largest diagonal first): on the poorly conditioned systems right after the IMU
    # initialisation it is the less accurate of the two; the facade's threshold pivoting stays at rounding level throughout
    assert r[:, 0].max() < 1e-4 and r[:, 2].max() < 1e-4 and r[:, 4].max() < 1e-4
    assert r[:, 1].max() < 1e-8 and r[:, 3].max() < 1e-8 and r[:, 5].max() < 1e-8


def test_frame_marginalisation_prior_on_the_data_of_a_chain(monkeypatch):
    """The facade's prior algebra of EnergyFunctional::marginalizeFrame (visual form, sosf_marginalize_frame_prior = what
    sosf_marginalize_frame runs on the system's prior) against the oracle's and against the NumPy mirror: on a window with a
    structured prior, and on every frame marginalisation of a rolling chain."""
    from oracle import mirror_np as mir
    from sos_slam_amd import synth
    seen = dict(n=0, worst_o=0.0, worst_m=0.0)
    orig = orc.OracleWindow.marginalize_frame_prior

    def scaled_err(A, B, bA, bB):
        s = 1.0 / np.sqrt(np.abs(np.diag(B)) + 10)
        return max(np.abs((A - B) * np.outer(s, s)).max() / max(np.abs(B * np.outer(s, s)).max(), 1.0),
                   np.abs((bA - bB) * s).max() / max(np.abs(bB * s).max(), 1.0))

    def both(self, idx):
        H0, b0 = self.get_prior()
        pr, dp = self.frame_prior(idx)
        Ho, bo = orig(self, idx)
        Hf, bf = host.marginalize_frame_prior(H0, b0, idx, pr, dp)
        Hm, bm = mir.marginalize_frame(H0, b0, idx, pr, dp)
        assert np.array_equal(Hf, Hf.T)
        seen["n"] += 1
        seen["worst_o"] = max(seen["worst_o"], scaled_err(Hf, Ho, bf, bo))
        seen["worst_m"] = max(seen["worst_m"], scaled_err(Hf, Hm, bf, bm))
        return Ho, bo

    monkeypatch.setattr(orc.OracleWindow, "marginalize_frame_prior", both)
    win = synth.make_window("T6")
    for idx in (0, 2, 4):
        ow = orc.window_from_synth(win)
        ow.optimize(3)
        ow.marginalize_points(np.flatnonzero(win.points["host"] == 1)[:40].astype(np.int32))
        ow.marginalize_frame_prior(idx)
        ow.close()
    sc = rolling.Scenario(n_frames=20)
    ch = rolling.OracleChain(sc)
    ch.bootstrap()
    while ch.next_frame < sc.n_frames:
        ch.step()
    print(seen)
    assert seen["n"] >= 3 + 8
    assert seen["worst_o"] < 1e-10 and seen["worst_m"] < 1e-9


def test_visual_solve_of_the_facade_on_the_data_of_a_chain():
    """sosf_solve_system (= what every Gauss-Newton iteration of the facade runs between the device's accumulation and the step:
    prior right-hand side, (1 + lambda), H_sc / (1 + lambda), Jacobi scaling, blocked LDL^T with threshold pivoting) against the NumPy
    mirror in extended precision, on every system the oracle solves at T6 / W7 and along a rolling chain.  The facade reads upper
    triangles, the mirror (as Eigen) lower ones: H_top and H_sc go in transposed; HM, of which bM + HM delta uses all, mirrored."""
    import ctypes as C
    from oracle import mirror_np as mir
    from sos_slam_amd import synth
    Lo = orc.lib()
    vp = C.c_void_p
    TAP = C.CFUNCTYPE(None, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp)
    Lo.orc_set_solve_tap.argtypes = [TAP]
    errs = []

    def arr(p, shape):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape).copy()

    def tap(n, H, b, Hsc, bsc, HM, bM, delta, lam, x):
        d = 4 + 8 * n
        Hh, Hs, M = arr(H, (d, d)), arr(Hsc, (d, d)), arr(HM, (d, d))
        M = np.tril(M) + np.tril(M, -1).T
        rest = (arr(b, (d,)), arr(bsc, (d,)), arr(bM, (d,)), arr(delta, (d,)))
        xm = mir.solve_system(Hh, rest[0], Hs, rest[1], M, rest[2], rest[3], lam)
        xf = host.solve_system(Hh.T, rest[0], Hs.T, rest[1], M, rest[2], rest[3], lam)
        errs.append(float(np.abs(xf - xm).max() / np.abs(xm).max()))

    cb = TAP(tap)
    Lo.orc_set_solve_tap(cb)
    try:
        for name in ("T6", "W7"):
            orc.window_from_synth(synth.make_window(name)).optimize(6)
        sc = rolling.Scenario(n_frames=14)
        ch = rolling.OracleChain(sc)
        ch.bootstrap()
        while ch.next_frame < sc.n_frames:
            ch.step()
    finally:
        Lo.orc_set_solve_tap(C.cast(None, TAP))
    e = np.array(errs)
    print(f"{len(e)} systems: facade vs NumPy mirror max {e.max():.1e}, median {np.median(e):.1e}")
    assert len(e) >= 25 and e.max() < 1e-8 and np.median(e) < 1e-10


@pytest.mark.parametrize("name", ["T6", "W7", "W12"])
def test_per_keyframe_host_math_of_the_facade_equals_the_oracle(name):
    """setEvalPT / setState, FrameFramePrecalc::set of every pair, setAdjointsF and setDeltaF's adHTdeltaF as the FACADE computes them
    (sosf_host_frame_math: the functions the system runs, on frames built for the occasion) against the oracle's, from the states three
    Gauss-Newton iterations leave: everything the device is handed per step that is not an image or a point.  Bit for bit."""
    from sos_slam_amd import synth
    win = synth.make_window(name)
    ow = orc.window_from_synth(win, nthreads=4)
    ow.optimize(3)
    n = win.n
    ev = np.stack([ow.evalpt(f) for f in range(n)])
    fr = [ow.frame(f) for f in range(n)]
    v, vz = ow.calib_value()
    c2w, pc, adH, adT, adHT = host.host_frame_math(ev, [f["state_zero"] for f in fr], [f["state"] for f in fr], win.frames["ab_exposure"], v, vz)
    assert np.array_equal(c2w, np.stack([f["camToWorld"] for f in fr]))
    po = ow.precalc()
    offd = np.array([h != t for t in range(n) for h in range(n)])
    for field in po.dtype.names:
        if field == "pad":
            continue
        assert np.array_equal(np.asarray(pc[field])[offd], np.asarray(po[field])[offd]), field
    assert np.array_equal(adH, ow.adHost()) and np.array_equal(adT, ow.adTarget())
    assert np.array_equal(adHT, ow.adHTdeltaF())
    assert np.abs(adHT).max() > 0 and np.abs(np.stack([f["state"] for f in fr])[:, :6]).max() > 1e-5
    ow.close()


@pytest.mark.parametrize("name", ["T4", "T6", "W7"])
def test_energy_threshold_of_the_facade_equals_the_oracle(name):
    """setNewFrameEnergyTH of the facade (sosf_new_frame_energy_th: the function the system runs) on the energies the oracle's last
    linearisation left, against the threshold the oracle set from them -- at counts where the float and the double product of the
    index differ (170 at T4, 1450 at W7)."""
    from sos_slam_amd import synth
    win = synth.make_window(name)
    ow = orc.window_from_synth(win)
    ow.optimize(2)
    res, wo, p = ow.res(), ow.new_energy_wo(), win.params
    e = wo[(res["target"] == win.n - 1) & (wo >= 0)]
    th = host.new_frame_energy_th(e, p["frameEnergyTHN"], p["frameEnergyTHFacMedian"], p["frameEnergyTHConstWeight"], p["overallEnergyTHWeight"])
    assert th == ow.frame(win.n - 1)["frameEnergyTH"]
    assert host.new_frame_energy_th(e[:0]) == 12 * 12 * 8
    ow.close()


def test_flag_frames_decision_of_the_facade_on_the_data_of_a_chain():
    """flagFramesForMarginalization as the facade decides it (sosf_flag_frames: the function the system runs on its own numbers) against the
    oracle chain's restatement, on every keyframe of a rolling sequence: which keyframes leave, and in which order, hangs on it."""
    seen = dict(calls=0, flagged=0, by_distance=0)

    class Both(rolling.OracleChain):
        def flag_frames(self, num_immature):
            fr = self.frames
            n = len(fr)
            before = np.array([f.flagged for f in fr], np.uint8)
            ids = [f.frameID for f in fr]
            n_in = [len(f.points) + ni for f, ni in zip(fr, num_immature)]
            n_out = [f.n_marg + f.n_out for f in fr]
            back = fr[-1]
            ref0 = [np.exp(f.state[6] * rolling.SCALE_A - back.state[6] * rolling.SCALE_A) for f in fr]
            dist = np.array([[self._distance(h, t) for t in fr] for h in fr], np.float32)
            want = super().flag_frames(num_immature)
            got = host.flag_frames(ids, n_in, n_out, ref0, dist) | before
            assert np.array_equal(got.astype(bool), want), (ids, got, want)
            seen["calls"] += 1
            seen["flagged"] += int(want.sum())
            seen["by_distance"] += int(n - int(want.sum()) < rolling.MAX_FRAMES and want.sum() > 0 and n >= rolling.MAX_FRAMES)
            return want

    for kw in (dict(n_frames=26), dict(n_frames=4 + 3 * 7, kf_every=3, step=0.07 / 3, rot=0.008 / 3)):
        sc = rolling.Scenario(**kw)
        ch = Both(sc)
        ch.bootstrap()
        while ch.next_frame < sc.n_frames:
            if ch.step() is None:
                break
    print(seen)
    assert seen["calls"] >= 25 and seen["flagged"] >= 15
