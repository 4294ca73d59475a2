"""The IMU branch of the facade's solveSystemF on the device-accumulated H / b (sosf_set_imu): with IMU terms switched
off (zero weights, no valid spline) and the expanded prior it must reproduce the plain solve of the same window; with
real IMU records the step it takes satisfies the spline constraints and moves the IMU states / scale."""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import ImuCalib, ImuFrame, ImuSettings, imu_dim
from tests.test_imu_assembly import _rot

pytestmark = pytest.mark.gpu


def _records(win, weights=True, valid=True, seed=0):
    rng = np.random.default_rng(seed)
    n = win.n
    S = ImuSettings()
    S.weight_imu[:] = list((np.eye(6) * (4.0 if weights else 0.0)).reshape(-1))
    S.weight_imu_bias[:] = list((np.eye(6) * (10.0 if weights else 0.0)).reshape(-1))
    S.gravity[:] = [0, 9.81, 0]
    S.rot_imu_cam[:] = list(_rot(np.array([0.1, -0.2, 0.05])).reshape(-1))
    S.maxImuInterval = 0.5
    S.enable_scale_opt = 0
    cal = ImuCalib(1.0 / 200, 1.0 / 200, 1, 1)
    frames, keep = [], []
    for i in range(n):
        f = ImuFrame()
        f.timestamp = 1.0 + 0.1 * i
        st = rng.normal(0, 1e-4, 21)
        f.state_imu[:] = list(st)
        f.state_imu_zero[:] = list(st)
        f.trackingRefIsPrev = 1 if valid else 0
        imu = np.zeros((8, 7))
        imu[:, 0] = f.timestamp - np.linspace(0.07, 0.0, 8)
        imu[:, 1:4] = [0.2, 9.6, -0.1]
        imu[:, 4:7] = rng.normal(0, 0.02, (8, 3))
        keep.append(imu)
        f.n_imu = 8
        f.imu = imu.ctypes.data
        frames.append(f)
    return S, cal, frames, keep


def test_imu_branch_wiring():
    from sos_slam_amd import host
    win = synth.make_window("T6")
    n = win.n
    d0, dI = 4 + 8 * n, imu_dim(n)
    idx = np.array([k if k < 4 else 5 + 29 * ((k - 4) // 8) + (k - 4) % 8 for k in range(d0)])
    HMi, bMi = np.zeros((dI, dI)), np.zeros(dI)
    HMi[np.ix_(idx, idx)] = win.HM
    bMi[idx] = win.bM
    # ---- IMU terms off: the branch (expansion, prior, Schur complement, compaction to 14 states per keyframe, LDLT on a
    # system with exact-zero pivots for the bias states, split) reproduces the plain solve
    plain, imu = host.System.from_window(win), host.System.from_window(win)
    plain.prepare(); imu.prepare()
    S, cal, frames, keep = _records(win, weights=False, valid=False)
    imu.set_imu(S, cal, frames, HMi, bMi)
    plain.gn_iteration(0); imu.gn_iteration(0)
    xa, xb = plain.lastX(), imu.lastX()
    assert np.abs(xa - xb).max() <= 1e-7 * np.abs(xa).max() + 1e-12, np.abs(xa - xb).max()
    ss, st, _, _ = imu.imu_state()
    assert ss == 0 or abs(ss) < 1e-12
    assert np.all(st == 0)
    plain.close(); imu.close()
    # ---- with IMU records: constraints of the KKT system hold for the step taken, states and scale move
    sysm = host.System.from_window(win)
    sysm.prepare()
    S, cal, frames, keep = _records(win)
    sysm.set_imu(S, cal, frames, HMi + np.eye(dI) * 1e-3, bMi)
    st0 = np.array([list(f.state_imu) for f in frames])
    sysm.gn_iteration(0)
    x = sysm.lastX()
    ss, st, st_new, scale = sysm.imu_state()
    assert np.isfinite(x).all() and np.abs(st).max() > 0 and ss != 0
    assert np.allclose(st_new, st0 + st) and np.isclose(scale, 1.0 / 200 + ss)
    # the constraints as the oracle assembles them from the records the facade filled in (poses of the solve)
    S2, cal2, frames2, keep2 = _records(win)
    arr = sysm._imu[2]
    for i in range(n):
        frames2[i].camToWorld[:] = list(arr[i].camToWorld)
        frames2[i].evalPT_R[:] = list(arr[i].evalPT_R)
        frames2[i].state_imu[:] = list(st0[i])
        frames2[i].state_imu_zero[:] = list(st0[i])
    H, b, J, r, sv = orc.imu().hessian(S2, cal2, frames2)
    assert list(sv) == [0] + [1] * (n - 1) and len(r) == 6 * (n - 2) + 3
    full = np.zeros(dI)
    full[:4] = x[:4]
    full[4] = -ss
    for i in range(n):
        full[5 + 29 * i:5 + 29 * i + 8] = x[4 + 8 * i:12 + 8 * i]
        full[5 + 29 * i + 8:5 + 29 * (i + 1)] = -st[i]
    assert np.allclose(J @ full, r, rtol=1e-6, atol=1e-8 * max(np.abs(r).max(), 1.0))
    sysm.set_imu(None)
    sysm.gn_iteration(1)
    assert np.isfinite(sysm.lastX()).all()
    sysm.close()
