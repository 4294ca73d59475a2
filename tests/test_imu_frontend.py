"""VIO front-end around the IMU assembly (FS/HessianBlocks.cpp:253-429), CPU: the facade's C++ (sosf_imu_propagate_state /
update_vel / initialize / try_trap_scale) against the independent NumPy restatement in oracle/imu_frontend.py, and
known-answer tests on data generated from the reference's own spline model (a cubic in the translation and in the so3
vector): the fit recovers the generating coefficients, the injected gyroscope bias and the metric scale."""
import numpy as np
import pytest

from oracle import imu_frontend as ref
from sos_slam_amd.records import ImuCalib, ImuFrame, ImuSettings, ImuShell


def _settings(rng, scale_opt=False):
    S = ImuSettings()
    S.weight_imu[:] = list(np.eye(6).reshape(-1))
    S.weight_imu_bias[:] = list(np.eye(6).reshape(-1))
    S.gravity[:] = [0.05, 9.78, 0.3]
    S.rot_imu_cam[:] = list(ref.so3_exp(rng.normal(0, 0.5, 3)).reshape(-1))
    S.maxImuInterval = 0.5
    S.enable_scale_opt = int(scale_opt)
    return S, dict(gravity=np.array(S.gravity[:]), rot_imu_cam=np.array(S.rot_imu_cam[:]))


def _frame(ts, c2w, state, imu, keep):
    f = ImuFrame()
    f.timestamp = ts
    f.camToWorld[:] = list(c2w)
    f.evalPT_R[:] = list(c2w[:9])
    f.state_imu[:] = list(state)
    f.state_imu_zero[:] = list(state)
    f.trackingRefIsPrev = 1
    keep.append(np.ascontiguousarray(imu, dtype=np.float64).reshape(-1, 7))
    f.n_imu = len(keep[-1])
    f.imu = keep[-1].ctypes.data if len(keep[-1]) else None
    return f


def _shell(ts, c2w, vel=(0, 0, 0)):
    s = ImuShell()
    s.timestamp = ts
    s.camToWorld[:] = list(c2w)
    s.velInWorld[:] = list(vel)
    return s


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_propagate_state_matches_restatement(seed):
    from sos_slam_amd import host
    fe = host.ImuFrontEnd()
    rng = np.random.default_rng(seed)
    S, Sd = _settings(rng)
    cal = ImuCalib(1.3 / 200.0, 1.3 / 200.0, 0, 1)
    keep = []
    last_ts, ts = 5.0, 5.11
    last_c2w = np.concatenate([ref.so3_exp(rng.normal(0, 0.3, 3)).reshape(-1), rng.normal(0, 1, 3)])
    c2w = np.concatenate([ref.so3_exp(rng.normal(0, 0.3, 3)).reshape(-1), rng.normal(0, 1, 3)])
    m = 14
    imu = np.zeros((m, 7))
    imu[:, 0] = np.sort(rng.uniform(last_ts, ts, m))
    imu[:, 1:4] = rng.normal(0, 0.5, (m, 3)) + [0, 9.8, 0]
    imu[:, 4:7] = rng.normal(0, 0.2, (m, 3))
    st0 = rng.normal(0, 1e-3, 21)
    f = _frame(ts, c2w, st0, imu, keep)
    sh, lsh = _shell(ts, c2w), _shell(last_ts, last_c2w, rng.normal(0, 0.3, 3))
    bias = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.01, 3)])
    fe.propagate_state(S, cal, f, sh, lsh, bias)
    sc_r, vel_r = ref.propagate_imu_state(Sd, cal.scale, ts, imu, last_ts, last_c2w[:9].reshape(3, 3), np.array(lsh.velInWorld[:]), bias,
                                          ref.scaled_of(st0))
    got = np.array(f.state_imu[:])
    assert np.all(np.isfinite(got))
    want = ref.state_of(sc_r)
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    assert np.array_equal(np.array(f.state_imu_zero[:]), got)                       # setImuStateZero
    assert np.abs(np.array(sh.velInWorld[:]) - vel_r).max() <= 1e-9 * max(1.0, np.abs(vel_r).max())
    assert np.allclose(ref.scaled_of(got)[:6], bias, rtol=1e-7)                     # imu_bias = last_imu_bias (float inverse constants)
    # the accelerometer fit IS the two-column least squares: the zero column only poisons row 0 of Eigen's "inverse"
    ss = 200.0 * cal.scale
    tt = imu[:, 0] - ts
    A2 = np.stack([np.full(m, 2 * ss), 6 * tt * ss], axis=1)
    R = last_c2w[:9].reshape(3, 3)
    rhs, tprev = [], last_ts
    for row in imu:
        R = R @ ref.so3_exp((row[4:7] - bias[3:]) * (row[0] - tprev))
        tprev = row[0]
        rhs.append(R @ Sd["rot_imu_cam"].reshape(3, 3).T @ (row[1:4] - bias[:3]) - Sd["gravity"])
    x2 = np.linalg.lstsq(A2, np.array(rhs), rcond=None)[0]
    sc = ref.scaled_of(got)
    assert np.allclose(sc[9:12], x2[0], rtol=1e-6, atol=1e-9) and np.allclose(sc[15:18], x2[1], rtol=1e-6, atol=1e-9)
    # updateVel
    fe.update_vel(f, sh, lsh)
    assert np.allclose(np.array(sh.velInWorld[:]), ref.update_vel(ref.scaled_of(got), ts, c2w[9:], last_ts, last_c2w[9:]), rtol=1e-12)


def test_eigen_singular_inverse_semantics():
    """rows 1, 2 of the "inverse" of [[0,0,0],[0,a,b],[0,b,c]] are [0 | inverse of the 2 x 2 block]; row 0 is inf / NaN"""
    M = np.array([[0, 0, 0], [0, 4.0, 1.0], [0, 1.0, 3.0]])
    with np.errstate(all="ignore"):
        Mi = ref.eigen_dynamic_inverse(M)
    assert np.allclose(Mi[1:, 1:], np.linalg.inv(M[1:, 1:])) and np.all(Mi[1:, 0] == 0)
    assert not np.any(np.isfinite(Mi[0]))
    R = np.random.default_rng(3).normal(size=(3, 3))
    R = R @ R.T + np.eye(3)
    assert np.allclose(ref.eigen_dynamic_inverse(R), np.linalg.inv(R))


def _model_scene(rng, scale_true=1.7, bias_g=(0.01, -0.02, 0.005), scale_opt=False):
    """five keyframes on a trajectory that IS the reference's model: translation and so3 vector cubic in (t - t_base)"""
    S, Sd = _settings(rng, scale_opt)
    Ric, g = Sd["rot_imu_cam"].reshape(3, 3), Sd["gravity"]
    ts = np.array([10.0, 10.12, 10.23, 10.37, 10.5])
    l0, q0, c0 = rng.normal(0, 0.3, 6), rng.normal(0, 0.3, 6), rng.normal(0, 0.3, 6)
    Rb, tb = ref.so3_exp(rng.normal(0, 0.3, 3)), rng.normal(0, 1, 3)
    c2w = []
    for t in ts - ts[4]:
        xi = l0 * t + q0 * t * t + c0 * t ** 3
        c2w.append(np.concatenate([(Rb @ ref.so3_exp(xi[3:])).reshape(-1), tb + xi[:3]]))
    c2w = np.array(c2w)
    imus, keep = [], []
    for i in range(5):
        m = 10
        tt = np.sort(rng.uniform(ts[i - 1] if i else ts[0] - 0.1, ts[i], m)) - ts[4]
        rows = np.zeros((m, 7))
        rows[:, 0] = tt + ts[4]
        for k, t in enumerate(tt):
            w = l0[3:] * t + q0[3:] * t * t + c0[3:] * t ** 3
            rot_ti_w = Ric @ ref.so3_exp(w).T @ Rb.T
            acc_w = 2 * q0[:3] + 6 * t * c0[:3]
            rows[k, 1:4] = rot_ti_w @ (scale_true * acc_w + g)
            rows[k, 4:7] = Ric @ (l0[3:] + 2 * t * q0[3:] + 3 * t * t * c0[3:]) + np.array(bias_g)
        imus.append(rows)
    frames = [_frame(ts[i], c2w[i], rng.normal(0, 1e-4, 21), imus[i], keep) for i in range(5)]
    shells = [_shell(ts[i], c2w[i]) for i in range(5)]
    return S, Sd, ts, c2w, imus, frames, shells, keep, (l0, q0, c0, Rb)


@pytest.mark.parametrize("scale_opt", [False, True])
def test_initialize_imu_matches_restatement_and_recovers_the_model(scale_opt):
    from sos_slam_amd import host
    fe = host.ImuFrontEnd()
    rng = np.random.default_rng(11)
    S, Sd, ts, c2w, imus, frames, shells, keep, (l0, q0, c0, Rb) = _model_scene(rng, scale_opt=scale_opt)
    cal = ImuCalib(1.0 / 200.0, 1.0 / 200.0, 0, 0)
    st_in = np.array([ref.scaled_of(np.array(f.state_imu[:])) for f in frames])
    ok, fo, so = fe.initialize(S, cal, frames, shells)
    r = ref.initialize_imu(Sd, 1.0 / 200.0, scale_opt, ts, c2w, c2w[4, :9], imus, st_in)
    assert ok and r["ok"] and cal.imu_initialized == 1
    st = np.array([np.array(f.state_imu[:]) for f in fo])
    want = np.array([ref.state_of(x) for x in r["scaled"]])
    assert np.abs(st - want).max() <= 1e-9 * np.abs(want).max()          # states: both sides apply the float INVERSE constants
    got = np.array([ref.scaled_of(x) for x in st])                        # scaled again: 2e-8 relative off (0.01f * 100 != 1)
    assert np.abs(np.array([s.velInWorld[:] for s in so]) - r["vel"]).max() <= 1e-8
    for f in fo:
        assert np.array_equal(np.array(f.state_imu_zero[:]), np.array(f.state_imu[:]))
    # known answers: the generating spline, the injected bias, the metric scale
    assert np.allclose(got[4, 6:9], l0[3:], rtol=1e-6, atol=1e-8) and np.allclose(got[4, 9:15], q0, rtol=1e-6, atol=1e-7)
    assert np.allclose(got[4, 15:21], c0, rtol=1e-6, atol=1e-6)
    assert np.allclose(got[:, 3:6], [0.01, -0.02, 0.005], atol=1e-9)
    assert np.all(got[:, 0:3] == 0)
    if scale_opt:
        assert cal.scale == 1.0 / 200.0                       # left to the optimiser
    else:
        assert abs(cal.scale * 200.0 - 1.7) < 1e-6 and cal.scale_zero == cal.scale   # 0.005f * 200 != 1
        assert abs(r["scale_scaled"] - 1.7) < 1e-8


def test_initialize_imu_fails_on_negative_scale():
    from sos_slam_amd import host
    fe = host.ImuFrontEnd()
    rng = np.random.default_rng(5)
    S, Sd, ts, c2w, imus, frames, shells, keep, _ = _model_scene(rng, scale_true=-0.8)
    cal = ImuCalib(1.0 / 200.0, 1.0 / 200.0, 0, 0)
    ok, fo, so = fe.initialize(S, cal, frames, shells)
    assert not ok and cal.imu_initialized == 0
    assert cal.scale * 200.0 == pytest.approx(-0.8, abs=1e-6)     # setScaleScaledZero ran before the test, as in the reference


def test_try_trap_scale():
    from sos_slam_amd import host
    fe = host.ImuFrontEnd()
    queue, qi = np.linspace(-10, -100, 10), 0                      # Vec10::LinSpaced(10, -10, -100)
    rq, rqi = queue.copy(), 0
    cal = ImuCalib(0.005, 0.005, 0, 1)
    trapped_at = None
    for it in range(14):
        cal.scale = 0.005 + 1e-6 * np.cos(it)
        queue, qi = fe.try_trap_scale(cal, queue, qi, 1e-4)
        zero, trapped, rq, rqi = ref.try_trap_scale(cal.scale, rq, rqi, 1e-4)
        assert np.array_equal(queue, rq) and qi == rqi
        assert bool(cal.scale_trapped) == trapped or trapped_at is not None
        if trapped and trapped_at is None:
            trapped_at = it
        if trapped:
            assert cal.scale_zero == pytest.approx(zero, rel=1e-14)
        elif trapped_at is None:
            assert cal.scale_zero == cal.scale
    assert trapped_at == 9                                          # the ten LinSpaced entries have to be flushed out first


def test_consistent_imu_records_of_the_bench_window_have_small_residuals():
    """synth.make_imu_records(consistent=True) (bench.py --imu): samples an IMU would deliver on make_window's trajectory.  At the
    trajectory's own poses the spline constraints and the IMU residuals of the oracle's assembly vanish up to the small-angle terms
    the zero spline states leave out -- orders of magnitude below what arbitrary samples produce."""
    from oracle import oracle as orc
    from sos_slam_amd import synth

    class W:
        n = 7
    for consistent in (True, False):
        S, cal, frames, keep = synth.make_imu_records(W, consistent=consistent)
        for i, f in enumerate(frames):
            R = synth.so3_exp(0.01 * i * np.array([0.3, 1.0, 0.2]))
            t = 0.08 * i * np.array([1.0, 0.1, 0.05])
            f.camToWorld[:] = list(R.reshape(-1)) + list(t)
            f.evalPT_R[:] = list(R.reshape(-1))
        H, b, J, r, sv = orc.imu().hessian(S, cal, frames)
        if consistent:
            assert list(sv) == [0] + [1] * (W.n - 1)
            rot_rows = np.abs(r).max()
            b_c = np.abs(b).max()
        else:
            b_a = np.abs(b).max()
    # rotation constraint: the spline's linear-rotation state is zero while the keyframes turn by 0.0108 rad -> r = that angle;
    # velocity constraint and IMU residuals: zero (constant velocity, gravity only)
    assert rot_rows < 0.02
    assert b_c < 0.05 * b_a, (b_c, b_a)
