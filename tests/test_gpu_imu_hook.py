"""The IMU branch of the facade's solveSystemF on the device-accumulated H / b (sosf_set_imu): with IMU terms switched
off (zero weights, no valid spline) and the expanded prior it must reproduce the plain solve of the same window; with
real IMU records the step it takes satisfies the spline constraints and moves the IMU states / scale."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import ImuCalib, ImuFrame, ImuSettings, imu_dim
from tests.test_imu_assembly import _rot

pytestmark = pytest.mark.gpu


# 3x only under tests/emu (whose roundings differ from the hardware); the round-3 bar of 2x stands on a GPU (advisor, round 4)
_EMU_FAC = 3.0 if (os.environ.get("SOS_EMU") == "1" and not __import__("torch").cuda.is_available()) else 2.0

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
            assert sysm.loop_mode() == 1     # the device-side step: the IMU block only changes what the host solves
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
    # the scale: with scale_trapped = 0 it is the weakly observable direction of the window (it moves from 1 / 200 to ~ -1e-4 here), so the
    # fp32 summation-order noise of the first solve (5e-6 of |x|: the top-Hessian tile sums come from the matrix cores, the oracle's from
    # its three-tier accumulators) shows in it amplified; the yardstick is 2e-4 of the distance the scale travelled, or 3 x the oracle's
    # own fp32-vs-fp64 distance, whichever is larger
    assert abs(sg - so) <= max(1e-7 * abs(so), 3 * abs(so - stt_), 2e-4 * abs(sg - 1.0 / 200)), (sg, so, stt_)
    sc = max(np.abs(sto).max(), 1e-12)
    assert np.abs(stg - sto).max() <= max(1e-5 * sc, 3 * np.abs(sto - stt).max())
    assert sg != 1.0 / 200 and np.abs(stg).max() > 2e-4            # the IMU states did move


def _refresh_records(ow, frames):
    """poses the IMU factors are linearised at: PRE_camToWorld and the rotation of the evaluation point (as the facade does)"""
    for i, f in enumerate(frames):
        f.camToWorld[:] = list(ow.frame(i)["camToWorld"])
        f.evalPT_R[:] = list(ow.evalpt(i)[:9])


def _stitched_delta(ow, n):
    v, vz = ow.calib_value()
    d = np.zeros(4 + 8 * n)
    d[:4] = (v - vz).astype(np.float32)            # cDeltaF is float (OB/EnergyFunctional.cpp:176)
    for f in range(n):
        fr = ow.frame(f)
        d[4 + 8 * f:12 + 8 * f] = fr["state"][:8] - fr["state_zero"][:8]
    return d


@pytest.mark.parametrize("name", ["T6", "W7"])
def test_imu_prior_lifecycle(name):
    """setting_enable_imu with the prior kept by the facade (sosf_set_imu with NULL priors): expansion at the start, six IMU
    iterations, marginalizePointsF into the expanded prior, the IMU form of marginalizeFrame, then the reduced window and a
    new keyframe -- against the same sequence assembled from the oracle's pieces (host loop with orc_host_set_imu,
    marginalize_points, expandHbtoFitImu, orc_imu_marginalize_frame) with the prior kept in NumPy."""
    from sos_slam_amd import host
    from tests import helpers as hp
    from tests.test_gpu_marginalize import _yardstick
    win = synth.make_window(name)
    n, w = win.n, float(win.params["margWeightFac"])
    api = orc.imu()
    # ---- device
    S, cal, frames, keep = _records(win)
    sysm = host.System.from_window(win)
    sysm.set_imu(S, cal, frames)
    H0, b0 = sysm.imu_prior()
    He, be = api.expand(n, win.HM, win.bM)
    assert np.array_equal(H0, He) and np.array_equal(b0, be)
    sysm.optimize(6)
    # ---- oracle (fp32 restatement and fp64-accumulated yardstick), prior in NumPy
    sides = {}
    for truth in (False, True):
        S2, cal2, frames2, keep2 = _records(win)
        ow = hp.oracle_window(win)
        ow.set_truth_mode(truth)
        HMo, bMo = api.expand(n, win.HM, win.bM)
        ow.set_imu(S2, cal2, frames2, HMo, bMo)
        ow.optimize(6)
        sides[truth] = dict(ow=ow, S=S2, cal=cal2, HM=HMo, bM=bMo)
    # ---- marginalizePointsF: points of keyframe 0 that still have residuals
    ow = sides[False]["ow"]
    res = ow.res()
    live = (res["flags"] & 0x100) == 0
    has = np.zeros(win.P, bool)
    has[res["point"][live]] = True
    alive0 = np.flatnonzero(has & (win.points["host"] == 0)).astype(np.int32)
    empty0 = np.flatnonzero(~has & (win.points["host"] == 0)).astype(np.int32)
    sel, rest = alive0[::2], alive0[1::2]
    for truth, sd in sides.items():
        o = sd["ow"]
        Hb, bb = o.get_prior()
        o.marginalize_points(sel)
        Ha, ba_ = o.get_prior()
        dH, db = api.expand(n, Ha - Hb, ba_ - bb)          # = margWeightFac * expandHbtoFitImu(M - Msc)
        sd["HM"] = sd["HM"] + dH
        sd["bM"] = sd["bM"] + db
    sysm.marginalize_points(sel)
    Hg, bg = sysm.imu_prior()
    # (W7 is larger than the windows of tests/test_gpu_marginalize.py, where the bar of 2 holds on the MI355X: under tests/emu the device's
    # tile sums land 2.5x as far from the fp64-accumulated prior as the reference's scalar sums do -- 5.3e-5 against 2.1e-5 of the largest
    # entry; this check has not run on a GPU yet)
    _yardstick(Hg, sides[False]["HM"], sides[True]["HM"], "expanded HM after marginalizePointsF", fac=_EMU_FAC)
    _yardstick(bg, sides[False]["bM"], sides[True]["bM"], "expanded bM after marginalizePointsF", fac=_EMU_FAC)
    assert np.abs(Hg - H0).max() > 0
    # ---- marginalizeFrame(0), IMU form
    ids2 = sysm.point_ids()
    sysm.drop_points(np.concatenate([rest, empty0[np.isin(empty0, ids2)]]))
    for truth, sd in sides.items():
        o = sd["ow"]
        o.drop_points(rest)
        fr = o._imu[2]
        _refresh_records(o, fr)
        pr, dp = o.frame_prior(0)
        sd["HM"], sd["bM"] = api.marginalize_frame(sd["S"], sd["cal"], list(fr), 0, _stitched_delta(o, n), pr, dp, sd["HM"], sd["bM"],
                                                   marg_weight=w)
    sysm.marginalize_frame(0)
    arr = sysm._imu[2]
    kept = [arr[i] for i in range(n - 1)]                # the facade erased record 0 in place; the prior stays with it
    assert [f.timestamp for f in kept] == [frames[i].timestamp for i in range(1, n)]
    sysm.set_imu(S, cal, kept)
    Hg, bg = sysm.imu_prior()
    assert Hg.shape == sides[False]["HM"].shape == (imu_dim(n - 1),) * 2
    _yardstick(Hg, sides[False]["HM"], sides[True]["HM"], "expanded HM after marginalizeFrame", fac=_EMU_FAC)   # (carries the prior of the step above)
    _yardstick(bg, sides[False]["bM"], sides[True]["bM"], "expanded bM after marginalizeFrame", fac=_EMU_FAC)
    assert np.abs(Hg - Hg.T).max() <= 1e-9 * np.abs(Hg).max()
    # ---- the reduced window optimises with the carried prior; the states keep moving, nothing blows up
    sc0 = cal.scale
    rm, it = sysm.optimize(3)
    assert np.isfinite(rm) and rm > 0
    _, st, st_new, scale = sysm.imu_state()
    assert st.shape == (n - 1, 21) and np.isfinite(st_new).all() and np.isfinite(scale) and scale != sc0
    # ---- insertFrame: 29 new states, zero rows / columns (OB/EnergyFunctional.cpp:666-677)
    Hb, bb = sysm.imu_prior()
    fr = win.frames[0].copy()
    fr["frameID"] = 1000
    sysm.add_frame(fr, win.images[0])
    Ha, ba_ = sysm.imu_prior()
    d = imu_dim(n - 1)
    assert Ha.shape == (imu_dim(n),) * 2
    assert np.array_equal(Ha[:d, :d], Hb) and np.array_equal(ba_[:d], bb)
    assert not Ha[d:, :].any() and not Ha[:, d:].any() and not ba_[d:].any()
    sysm.close()
    for sd in sides.values():
        sd["ow"].close()
