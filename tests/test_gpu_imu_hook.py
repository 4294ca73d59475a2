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


@pytest.mark.parametrize("name,trapped", [("T6", 1), ("T6", 0), ("W7", 1)])
def test_imu_branch_step_equals_oracle_solve_on_device_system(name, trapped):
    """N1 on the device's own numbers: the facade's IMU branch (OB/EnergyFunctional.cpp:1053-1171) runs on the H / b the
    kernels delivered; the oracle's restatement of that branch (orc_imu_solve) is given exactly those matrices, the
    records as the facade filled them and the same delta -- the pose increment, the scale step and the 21 IMU steps per
    keyframe must agree to solver round-off."""
    from sos_slam_amd import host
    win = synth.make_window(name)
    n = win.n
    d0, dI = 4 + 8 * n, imu_dim(n)
    idx = np.array([k if k < 4 else 5 + 29 * ((k - 4) // 8) + (k - 4) % 8 for k in range(d0)])
    HMi, bMi = np.zeros((dI, dI)), np.zeros(dI)
    HMi[np.ix_(idx, idx)] = win.HM
    bMi[idx] = win.bM
    HMi += np.eye(dI) * 1e-3
    sysm = host.System.from_window(win)
    sysm.prepare()
    sysm.keep_last_system(True)
    S, cal, frames, keep = _records(win)
    cal.scale_trapped = trapped
    sysm.set_imu(S, cal, frames, HMi, bMi)
    st0 = np.array([list(f.state_imu) for f in frames])
    scale0 = cal.scale
    delta = np.zeros(d0)
    for f in range(n):
        fr = sysm.frame(f)
        delta[4 + 8 * f:12 + 8 * f] = fr["state"][:8] - fr["state_zero"][:8]
    sysm.gn_iteration(0)
    H, b, Hsc, bsc = sysm.last_system()
    assert np.abs(H - H.T).max() == 0 and np.abs(H).max() > 0 and np.abs(Hsc).max() > 0
    x_g = sysm.lastX()
    ss_g, st_g, _, _ = sysm.imu_state()
    # the oracle's branch on the same inputs: records with the poses of that solve and the pre-step IMU states / scale
    S2, cal2, frames2, keep2 = _records(win)
    cal2.scale_trapped = trapped
    cal2.scale = scale0
    arr = sysm._imu[2]
    for i in range(n):
        frames2[i].camToWorld[:] = list(arr[i].camToWorld)
        frames2[i].evalPT_R[:] = list(arr[i].evalPT_R)
        frames2[i].state_imu[:] = list(st0[i])
        frames2[i].state_imu_zero[:] = list(st0[i])
    x_o, ss_o, st_o = orc.imu().solve(S2, cal2, frames2, H, b, Hsc, bsc, HMi, bMi, delta, lam=1e-5)
    sx = max(np.abs(x_o).max(), 1e-12)
    assert np.abs(x_g - x_o).max() <= 1e-7 * sx, (np.abs(x_g - x_o).max(), sx)
    assert abs(ss_g - ss_o) <= 1e-7 * max(abs(ss_o), 1e-9)
    assert np.abs(st_g - st_o).max() <= 1e-7 * max(np.abs(st_o).max(), 1e-12)
    assert np.abs(st_o).max() > 0 and ss_o != 0
    sysm.close()


@pytest.mark.parametrize("name,trapped", [("T6", 1), ("T6", 0), ("W7", 1)])
def test_optimize_with_imu_matches_oracle_loop(name, trapped):
    """Six Gauss-Newton iterations with the IMU branch on both sides: the facade's loop on the device's H / b against the
    oracle's host loop (orc_optimize with orc_host_set_imu).  The IMU factors couple poses, scale and the 21 IMU states of
    every keyframe, so this exercises the assembly at a moving linearisation point (poses of every solve, stepped states)."""
    from sos_slam_amd import host
    win = synth.make_window(name)
    n = win.n
    d0, dI = 4 + 8 * n, imu_dim(n)
    idx = np.array([k if k < 4 else 5 + 29 * ((k - 4) // 8) + (k - 4) % 8 for k in range(d0)])
    HMi, bMi = np.zeros((dI, dI)), np.zeros(dI)
    HMi[np.ix_(idx, idx)] = win.HM
    bMi[idx] = win.bM
    HMi += np.eye(dI) * 1e-3
    from tests import helpers as hp
    out = {}
    for side in ("device", "oracle", "truth"):
        S, cal, frames, keep = _records(win)
        cal.scale_trapped = trapped
        if side == "device":
            sysm = host.System.from_window(win)
            sysm.set_imu(S, cal, frames, HMi, bMi)
            rm, it = sysm.optimize(6)
            _, _, st, scale = sysm.imu_state()
            poses = np.array([sysm.frame(f)["camToWorld"] for f in range(n)])
            sysm.close()
        else:
            ow = hp.oracle_window(win)
            ow.set_truth_mode(side == "truth")
            ow.set_imu(S, cal, frames, HMi, bMi)
            rm, it = ow.optimize(6, nthreads=1)
            scale, st = ow.imu_state()
            poses = np.array([ow.frame(f)["camToWorld"] for f in range(n)])
            ow.close()
        out[side] = (rm, it, poses, scale, st)
    (rg, ig, pg, sg, stg), (ro, io, po, so, sto), (rt, itt, pt, stt_, stt) = out["device"], out["oracle"], out["truth"]
    e_go, e_gt, e_ot = np.abs(pg - po).max(), np.abs(pg - pt).max(), np.abs(po - pt).max()
    print(f"{name} trapped={trapped}: iterations {ig}/{io}, pose device-oracle {e_go:.3g} device-truth {e_gt:.3g} oracle-truth {e_ot:.3g}; "
          f"scale {sg:.9g} / {so:.9g}; max |state_imu| {np.abs(stg).max():.3g}")
    assert ig == io
    assert abs(rg - ro) <= 1e-5 * abs(ro)
    assert e_go < max(1e-5, 3 * e_ot)
    assert e_gt < max(1e-5, 2 * e_ot)
    assert abs(sg - so) <= max(1e-7 * abs(so), 3 * abs(so - stt_))
    sc = max(np.abs(sto).max(), 1e-12)
    assert np.abs(stg - sto).max() <= max(1e-5 * sc, 3 * np.abs(sto - stt).max())
    assert sg != 1.0 / 200 and np.abs(stg).max() > 2e-4            # the IMU states did move
