"""The visual-inertial solve with first-estimate Jacobians on a KEPT factor (sosf_imu_solve_prepare / _finish, csrc/host/sos_imu.cpp):
once the scale is trapped everything of the KKT matrix of OB/EnergyFunctional.cpp:1062-1140 except the visual block is constant over
the iterations of one optimize(), so the IMU states and constraint multipliers are eliminated once.  CPU only: against the oracle's
literal solve, against the facade's own literal form, and the rules under which the kept factor is dropped."""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import host
from sos_slam_amd.records import imu_dim
from tests.test_imu_assembly import _scene


def _system(n, seed=3, dense_prior=True):
    rng = np.random.default_rng(seed)
    d0, dI = 4 + 8 * n, imu_dim(n)
    A = rng.normal(size=(d0, d0 + 4))
    H_top = A @ A.T * 50 + np.eye(d0) * 200
    B = rng.normal(size=(d0, 6))
    H_sc = B @ B.T
    b_top, b_sc, delta = rng.normal(size=d0) * 10, rng.normal(size=d0), rng.normal(size=d0) * 1e-3
    Mq = rng.normal(size=(dI, 8))
    HM, bM = (Mq @ Mq.T if dense_prior else 0) + np.eye(dI) * 5, rng.normal(size=dI)
    return H_top, b_top, H_sc, b_sc, np.ascontiguousarray(HM), bM, delta


def _close(a, b, tol):
    xa, sa, ia = a
    xb, sb, ib = b
    sc = max(np.abs(xb).max(), np.abs(ib).max(), 1e-9)
    return np.abs(xa - xb).max() < tol * sc and abs(sa - sb) < tol * max(abs(sb), 1e-9) and np.abs(ia - ib).max() < tol * sc


@pytest.fixture
def api():
    f = host.imu()
    before = f.solve_mode(1)
    f.solve_stats(reset=True)
    yield f
    f.solve_mode(before)


@pytest.mark.parametrize("scale_opt", [False, True])
@pytest.mark.parametrize("n", [5, 8])
def test_kept_factor_follows_the_iterations(api, n, scale_opt):
    """Six 'iterations': the IMU states, the scale, delta and the visual system move, the linearisation points do not -- one factor,
    five solves on it, each equal to the oracle's literal solve of the same inputs."""
    S, cal, frames, keep = _scene(n=n, trapped=True, scale_opt=scale_opt, seed=n)
    H_top, b_top, H_sc, b_sc, HM, bM, delta = _system(n)
    rng = np.random.default_rng(11)
    fo = orc.imu()
    for it in range(6):
        xo = fo.solve(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta)
        xc = api.solve(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta)
        assert _close(xc, xo, 1e-8), it
        api.solve_mode(0)
        xl = api.solve(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta)
        api.solve_mode(1)
        assert _close(xc, xl, 1e-8), it
        # doStepFromBackup: states and scale step, the visual system is re-accumulated at the new states
        for f in frames:
            for k in range(21):
                f.state_imu[k] += 1e-5 * rng.normal()
            R = np.array(f.camToWorld[:9]).reshape(3, 3)
            w = rng.normal(0, 1e-3, 3)
            Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            f.camToWorld[:9] = list((R @ (np.eye(3) + Kx)).reshape(-1))
            for k in range(3):
                f.camToWorld[9 + k] += 1e-3 * rng.normal()
        cal.scale += 1e-6 * rng.normal()
        H_top = H_top * (1 + 0.02 * rng.random()) + np.diag(rng.random(len(b_top)))
        b_top = b_top + rng.normal(size=len(b_top))
        H_sc = H_sc * (1 + 0.02 * rng.random())
        delta = delta + 1e-4 * rng.normal(size=len(delta))
    kept, rebuilt, literal = api.solve_stats()
    assert (kept, rebuilt, literal) == (5, 1, 6)        # (the six literal solves are the mode-0 calls above)


def test_untrapped_scale_takes_the_literal_form(api):
    S, cal, frames, keep = _scene(trapped=False)
    sysm = _system(len(frames))
    xo = orc.imu().solve(S, cal, frames, *sysm)
    for _ in range(2):
        assert _close(api.solve(S, cal, frames, *sysm), xo, 1e-8)
    assert api.solve_stats() == (0, 0, 2)


def test_what_drops_the_kept_factor(api):
    S, cal, frames, keep = _scene(n=6, trapped=True)
    H_top, b_top, H_sc, b_sc, HM, bM, delta = _system(6)
    fo = orc.imu()

    def both():
        xo = fo.solve(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta)
        xc = api.solve(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta)
        assert _close(xc, xo, 1e-8)
        return api.solve_stats(reset=True)

    assert both() == (0, 1, 0)
    assert both() == (1, 0, 0)
    keep[4][3, 1:] += 0.01                       # a measurement: right-hand side only
    assert both() == (1, 0, 0)
    bM[7] += 1.0                                 # the prior's vector likewise
    assert both() == (1, 0, 0)
    HM[40, 91] += 0.5; HM[91, 40] += 0.5         # ONE off-diagonal pair of the prior
    assert both() == (0, 1, 0)
    assert both() == (1, 0, 0)
    frames[2].state_imu_zero[15] += 1e-7         # a linearisation point
    assert both() == (0, 1, 0)
    keep[1][5, 0] -= 1e-4                        # a sample's timestamp (its Jacobian's time)
    assert both() == (0, 1, 0)
    frames[3].evalPT_R[1] += 1e-9
    assert both() == (0, 1, 0)
    frames[4].timestamp += 1e-6
    # the fourth change in a row: the inputs move with every call, the literal form serves until they repeat
    assert both() == (0, 0, 1)
    assert both() == (0, 1, 0)
    assert both() == (1, 0, 0)
    S.weight_imu[0] *= 1.5
    assert both() == (0, 1, 0)
    cal.scale_zero *= 1.0001
    assert both() == (0, 1, 0)
    cal.scale *= 1.0001                          # the current scale: residuals and delta only
    assert both() == (1, 0, 0)


def test_named_prior(api):
    """prior_id != 0: same name, same pointer, same diagonal = the same prior without comparing dim^2 values"""
    S, cal, frames, keep = _scene(n=6, trapped=True)
    H_top, b_top, H_sc, b_sc, HM, bM, delta = _system(6)
    fo = orc.imu()

    def both(pid):
        xo = fo.solve(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta)
        xc = api.solve_two_calls(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta, prior_id=pid)
        assert _close(xc, xo, 1e-8)
        return api.solve_stats(reset=True)

    assert both(7) == (0, 1, 0)
    assert both(7) == (1, 0, 0)
    HM[10:20, 10:20] += np.eye(10)               # the facade writes its prior and renames it
    assert both(8) == (0, 1, 0)
    assert both(8) == (1, 0, 0)
    HM[30, 30] += 1.0                            # a write that was NOT announced still shows on the diagonal
    assert both(8) == (0, 1, 0)
    assert both(0) == (0, 1, 0)                  # unnamed after named: compared by value from now on
    assert both(0) == (1, 0, 0)


def test_two_calls_protocol(api):
    S, cal, frames, keep = _scene(n=5, trapped=True)
    sysm = _system(5)
    x = np.zeros(4 + 8 * 5)
    import ctypes as C
    ss, si = C.c_double(0), np.zeros((5, 21))
    a = [np.ascontiguousarray(v) for v in sysm[:4]]
    from sos_slam_amd.host import _p
    assert api.L.sosf_imu_solve_finish(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(x), C.byref(ss), _p(si)) != 0      # nothing prepared
    seen = []
    assert api.L.sosf_imu_solve_prepared_form() == -1
    xc = api.solve_two_calls(S, cal, frames, *sysm, between=lambda: seen.append(api.L.sosf_imu_solve_prepared_form()))
    assert seen == [1] and _close(xc, orc.imu().solve(S, cal, frames, *sysm), 1e-8)
    assert api.L.sosf_imu_solve_finish(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(x), C.byref(ss), _p(si)) != 0      # consumed


@pytest.mark.parametrize("n,m", [(60, 0), (60, 60), (60, 37), (401, 300), (193, 150)])
def test_partial_factorisation_equals_the_full_solve(n, m):
    """ldlt_partial_*: a quasi-definite leading block (states, then multipliers with a zero diagonal and one all-zero constraint row)
    eliminated with pivots from that block only; the trailing block solved on its own"""
    rng = np.random.default_rng(n + m)
    B = rng.normal(size=(n, n + 4))
    A = B @ B.T / n + np.eye(n)
    k = min(12, m // 3)
    if k:
        s0 = m - k
        J = np.zeros((k, n))
        J[np.arange(k), rng.permutation(s0)[:k]] = 1.0     # independent rows within the leading block, as the spline constraints are
        J[:, m:] = 0.3 * rng.normal(size=(k, n - m)) * (rng.random((k, n - m)) < 0.1)
        J[k - 1] = 0.0
        A[s0:m, :] = J
        A[:, s0:m] = J.T
        A[s0:m, s0:m] = 0
    b = rng.normal(size=n)
    if k:
        b[m - 1] = 0.0
    xr = host.ldlt_solve(A, b, 1)
    xp = host.ldlt_partial_solve(A, b, m)
    assert np.abs(xp - xr).max() < 1e-9 * np.abs(xr).max()


def test_baseline_size_against_the_oracle(api):
    """W12 (twelve keyframes, consistent IMU samples): dimension 353 + 66 constraints"""
    from sos_slam_amd import synth
    win = synth.make_window("W12")
    S, cal, fr, keep = synth.make_imu_records(win, consistent=True)
    HMi, bMi = synth.expand_prior_imu(win)
    H_top, b_top, H_sc, b_sc, _, _, delta = _system(win.n, seed=1)
    rng = np.random.default_rng(2)
    Mq = rng.normal(size=(HMi.shape[0], 8))
    HM = np.ascontiguousarray(HMi + Mq @ Mq.T)
    xo = orc.imu().solve(S, cal, fr, H_top, b_top, H_sc, b_sc, HM, bMi, delta)
    for _ in range(2):
        assert _close(api.solve(S, cal, fr, H_top, b_top, H_sc, b_sc, HM, bMi, delta), xo, 1e-7)
    assert api.solve_stats() == (1, 1, 0)


def test_states_without_information_are_left_to_the_literal_form(api):
    """A valid spline with no IMU sample and a prior that says nothing about its states: the leading block is singular beyond its
    all-zero constraint rows, so what the step is there is decided by the literal form's pivoting over the whole system."""
    S, cal, frames, keep = _scene(n=5, trapped=True)
    frames[1].n_imu = 0
    H_top, b_top, H_sc, b_sc, HM, bM, delta = _system(5, dense_prior=False)
    g0 = 4 + 1 + 29 * 1 + 8 + 6
    HM[g0:g0 + 15, :] = 0
    HM[:, g0:g0 + 15] = 0
    xs = []
    for mode in (0, 1, 1):
        api.solve_mode(mode)
        xs.append(api.solve(S, cal, frames, H_top, b_top, H_sc, b_sc, HM, bM, delta))
    api.solve_mode(1)
    for a in xs[1:]:
        assert all(np.array_equal(u, v) for u, v in zip(a, xs[0]))
    kept, rebuilt, literal = api.solve_stats()
    assert kept == 0 and rebuilt == 1 and literal == 3       # examined once, not again for the same inputs


@pytest.mark.parametrize("stereo,n_frames", [(False, 26), (True, 12)])
def test_kept_factor_on_the_systems_of_a_running_chain(api, stereo, n_frames):
    """Every IMU solve of the oracle's rolling visual-inertial chain (real priors out of point / frame marginalisations, entries of
    1e8 beside entries of 1e-19; six iterations per keyframe on one linearisation) handed to the facade's solve as well: the kept
    factor serves the iterations after the first of every optimize() once the scale is trapped, and its steps are the oracle's."""
    import ctypes as C
    from tests import rolling
    Lo = orc.lib()
    vp = C.c_void_p
    TAP = C.CFUNCTYPE(None, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp, C.c_double, vp)
    Lo.orc_set_imu_solve_tap.argtypes = [TAP]
    api.L.sosf_imu_solve.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp, vp, vp]
    worst = dict(x=0.0, imu=0.0, scale=0.0, kept_vs_literal=0.0)
    seen = dict(trapped=0, untrapped=0)

    def tap(S, Cal, n, F, H, b, Hsc, bsc, HM, bM, delta, lam, x, scale_step, step_imu):
        d0 = 4 + 8 * n
        xo = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), (d0,)).copy()
        so = np.ctypeslib.as_array(C.cast(step_imu, C.POINTER(C.c_double)), (21 * n,)).copy()
        # The oracle's H_sc carries the reference's fp32-level asymmetry (OB/AccumulatedSCHessian.cpp:128-139 fills block (j, k) and
        # block (k, j) from float accumulators rounded as (w L_p) R_q and (w R_q) L_p) and its solve reads the LOWER triangle, as
        # Eigen's ldlt() does (oracle/orc_math.h); the facade reads the UPPER one (all the device delivers).  Handing over the
        # transposes gives both the same numbers -- the two halves differ by 1e-7 of H_sc, which moves the step by 1e-4: the
        # sensitivity the GPU tests price with their fp64-truth yardstick.
        Ht = np.ascontiguousarray(np.ctypeslib.as_array(C.cast(H, C.POINTER(C.c_double)), (d0, d0)).T)
        Hst = np.ascontiguousarray(np.ctypeslib.as_array(C.cast(Hsc, C.POINTER(C.c_double)), (d0, d0)).T)
        res = []
        for mode in (1, 0):     # the kept factor where it applies, then the literal form of the same inputs (not counted below)
            api.solve_mode(mode)
            xf, sf, ss = np.zeros(d0), np.zeros(21 * n), C.c_double(0)
            rc = api.L.sosf_imu_solve(S, Cal, n, F, Ht.ctypes.data, b, Hst.ctypes.data, bsc, HM, bM, delta, lam, xf.ctypes.data, C.addressof(ss),
                                      sf.ctypes.data)
            assert rc == 0
            res.append((xf, sf, ss.value))
        api.solve_mode(1)
        (xf, sf, ssv), (xl, sl, ssl) = res
        from sos_slam_amd.records import ImuCalib
        trapped = C.cast(Cal, C.POINTER(ImuCalib)).contents.scale_trapped
        seen["trapped" if trapped else "untrapped"] += 1
        sc = max(np.abs(xo).max(), 1e-12)
        worst["x"] = max(worst["x"], np.abs(xf - xo).max() / sc)
        worst["imu"] = max(worst["imu"], np.abs(sf - so).max() / max(np.abs(so).max(), 1e-12))
        worst["scale"] = max(worst["scale"], abs(ssv - scale_step) / max(abs(scale_step), 1e-9 * sc, 1e-15))
        worst["kept_vs_literal"] = max(worst["kept_vs_literal"], np.abs(xf - xl).max() / sc, np.abs(sf - sl).max() / max(np.abs(sl).max(), 1e-12))

    cb = TAP(tap)
    Lo.orc_set_imu_solve_tap(cb)
    try:
        sc = rolling.Scenario(n_frames=n_frames, vio=True, stereo=stereo)
        ch = rolling.OracleChain(sc)
        ch.bootstrap()
        while ch.next_frame < sc.n_frames:
            ch.step()
    finally:
        Lo.orc_set_imu_solve_tap(C.cast(None, TAP))
    kept, rebuilt, literal = api.solve_stats()
    total = seen["trapped"] + seen["untrapped"]
    literal -= total            # the mode-0 repeats
    print(f"stereo {stereo}: solves trapped / untrapped {seen}, kept {kept} rebuilt {rebuilt} literal {literal}, worst relative difference {worst}")
    assert seen["trapped"] >= 12 and kept >= seen["trapped"] // 2 and kept + rebuilt + literal == total
    assert literal == seen["untrapped"]
    # the kept factor is the literal form to rounding on every system of the chain; the facade's solve and the oracle's (threshold
    # pivoting / full pivoting on the diagonal) part by up to 1e-6 on the poorly conditioned systems right after the IMU initialisation
    assert worst["kept_vs_literal"] < 1e-8, worst
    assert worst["x"] < 1e-5 and worst["imu"] < 1e-5 and worst["scale"] < 1e-5, worst
