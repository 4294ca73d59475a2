"""optimize() against the oracle at the BASELINE.json sizes: W7 (the reference preset, 7 KF x 2000 points), W12 (headline,
12 x 4096) and W16 (config 5's window, 16 x 8192), all at 752 x 480.

Bars (north_star): pose RMSE vs the CPU path < 1e-5, active index sets bit-exact, same iteration count.  Yardstick as in
tests/test_gpu_optimize.py: the oracle run with fp64 H/b accumulation ("truth") -- the device may be no further from it
than twice the fp32 restatement of the reference is.  Also compared here, because FS/FullSystemOptimize.cpp:55-71 writes
them and SURVEY.md 8(b) lists them as outputs the caller reads: PointHessian::maxRelBaseline / numGoodResiduals."""
import numpy as np
import pytest

from sos_slam_amd import synth
from tests import helpers as hp

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-5


def _pose_rmse(get_a, get_b, n):
    err = [get_a(f)["camToWorld"] - get_b(f)["camToWorld"] for f in range(n)]
    return float(np.sqrt(np.mean(np.square(np.concatenate(err)))))


@pytest.mark.parametrize("name", ["W7", "W12", "W16"])
def test_optimize_pose_rmse_and_index_sets(name):
    from sos_slam_amd import host
    win = synth.make_window(name)
    ow, ot = hp.oracle_window(win), hp.oracle_window(win)
    ot.set_truth_mode(True)
    rm_o, it_o = ow.optimize(6, nthreads=6)
    ot.optimize(6, nthreads=6)
    sysm = host.System.from_window(win)
    rm_g, it_g = sysm.optimize(6)
    assert it_g == it_o
    assert abs(rm_g - rm_o) <= 1e-5 * abs(rm_o)
    noise = _pose_rmse(ow.frame, ot.frame, win.n)      # reference fp32 restatement vs fp64-accumulated run
    e_ref = _pose_rmse(sysm.frame, ow.frame, win.n)    # device vs reference restatement
    e_tru = _pose_rmse(sysm.frame, ot.frame, win.n)    # device vs fp64-accumulated run
    print(f"{name}: pose RMSE device-oracle {e_ref:.3g}, device-truth {e_tru:.3g}, oracle-truth {noise:.3g}")
    assert e_ref < POSE_TOL, (e_ref, noise)
    assert e_tru < max(POSE_TOL, 2 * noise), (e_tru, noise)
    for f in range(win.n):
        a, b = sysm.frame(f), ow.frame(f)
        assert np.abs(a["state"] - b["state"]).max() < POSE_TOL
        assert abs(a["frameEnergyTH"] - b["frameEnergyTH"]) <= 1e-4 * b["frameEnergyTH"]
    # ---- active index set after the final linearizeAll(true): identity of every surviving residual
    ro = ow.res()
    alive = (ro["flags"] & 0x100) == 0
    set_o = set(zip(ro["point"][alive].tolist(), ro["target"][alive].tolist()))
    pi, tf = sysm.residual_ids()                       # frameID == frame idx in these windows
    set_g = set(zip(pi.tolist(), tf.tolist()))
    assert len(set_g) == len(pi)
    assert set_g == set_o, (len(set_g - set_o), len(set_o - set_g))
    # ... and their states, residual by residual
    rg = sysm.residuals()
    st_o = {(int(p), int(t)): int(s) for p, t, s in zip(ro["point"][alive], ro["target"][alive], ro["state_state"][alive])}
    assert all(st_o[(int(p), int(t))] == int(s) for p, t, s in zip(pi, tf, rg["state_state"]))
    assert np.all(rg["isActive"] == 1)
    # ---- per-point outputs of linearizeAll(true), FS/FullSystemOptimize.cpp:55-71
    pg = sysm.points()
    ids = sysm.point_ids()
    assert np.array_equal(ids, np.arange(win.P))       # optimize() removes residuals, never points
    assert np.array_equal(pg["numGoodResiduals"], ow.num_good_residuals())
    mrb_o = ow.point_field("maxRelBaseline")
    assert np.abs(pg["maxRelBaseline"] - mrb_o).max() <= 1e-4 * np.abs(mrb_o).max()
    assert np.abs(mrb_o).max() > 0
    idh_o = ow.point_field("idepth_hessian")
    e_idh = np.abs(pg["idepth_hessian"] - idh_o).max() / np.abs(idh_o).max()
    assert e_idh <= 2e-3
    # the yardstick form of the same check (verdict of round 2: 2e-3 is loose for a per-point fp32 sum).  On the CPU the oracle's own
    # fp32-vs-fp64 distance is 1e-5 / 4e-6 / 7e-5 at W7 / W12 / W16, so the bar would be max(1e-4, 3 x that).  Written while the GPU
    # was unreachable: reported as a warning until the device's number has been seen once, then to become the assertion.
    y_idh = np.abs(idh_o - ot.point_field("idepth_hessian")).max() / np.abs(idh_o).max()
    import warnings
    warnings.warn(f"idepth_hessian {name}: device-oracle {e_idh:.3g}, oracle-truth {y_idh:.3g}, candidate bar {max(1e-4, 3 * y_idh):.3g} "
                  f"({'within' if e_idh <= max(1e-4, 3 * y_idh) else 'OUTSIDE'})")
    po = ow.pts()
    assert np.abs(pg["idepth"] - po["idepth_scaled"]).max() <= 1e-4
    sysm.close()
    ow.close()
    ot.close()


REJECT_CASES = [("T4", {}), ("T4", dict(state_noise=1e-2, idepth_noise=0.1, seed=synth.SEED + 1)),
                ("T4", dict(state_noise=2e-2, idepth_noise=0.05, seed=synth.SEED + 2)),
                ("T4", dict(state_noise=3e-2, idepth_noise=0.1)), ("T6", {})]


@pytest.mark.parametrize("name,kw", REJECT_CASES)
def test_step_rejection_matches_oracle(name, kw):
    """setting_forceAceptStep off (FS/FullSystemOptimize.cpp:387-413).  The default T4 window accepts two steps and then
    rejects the rest (the energy rises by 1.7e-4 relative near convergence); the perturbed windows reject one to three
    steps in the middle of the loop -- after a rejection the window is re-linearised at the backup state, where the
    re-estimated frame threshold can make the same step acceptable.  Same accept / reject sequence, iteration count and
    index sets as the oracle; poses within the yardstick (the last case starts centimetres off: its fp32 and
    fp64-accumulated oracle runs already differ by 3e-4, so it only pins the decisions)."""
    from sos_slam_amd import host
    win = synth.make_window(name, **kw)
    ow, ot = hp.oracle_window(win), hp.oracle_window(win)
    ot.set_truth_mode(True)
    rm_o, it_o, rej_o = ow.optimize_ex(6, force_accept=False)
    rm_t, it_t, rej_t = ot.optimize_ex(6, force_accept=False)
    sysm = host.System.from_window(win)
    sysm.set_force_accept_step(False)
    rm_g, it_g = sysm.optimize(6)
    rej_g = sysm.rejected_steps()
    noise = _pose_rmse(ow.frame, ot.frame, win.n)
    e_ref = _pose_rmse(sysm.frame, ow.frame, win.n)
    e_tru = _pose_rmse(sysm.frame, ot.frame, win.n)
    print(f"{name} {kw}: iterations {it_g}/{it_o}, rejected {rej_g}/{rej_o}, rmse {rm_g}/{rm_o}/{rm_t}, pose device-oracle {e_ref:.3g} "
          f"device-truth {e_tru:.3g} oracle-truth {noise:.3g}")
    assert (it_g, rej_g) == (it_o, rej_o)
    if name == "T4":
        assert rej_o >= 1
    assert abs(rm_g - rm_o) <= max(1e-5 * abs(rm_o), 3 * abs(rm_o - rm_t))
    assert e_ref < max(POSE_TOL, 3 * noise), (e_ref, noise)
    assert e_tru < max(POSE_TOL, 2 * noise), (e_tru, noise)
    if noise < 1e-5:                      # the index sets are only comparable while the runs have not drifted apart
        ro = ow.res()
        alive = (ro["flags"] & 0x100) == 0
        pi, tf = sysm.residual_ids()
        assert set(zip(pi.tolist(), tf.tolist())) == set(zip(ro["point"][alive].tolist(), ro["target"][alive].tolist()))
    # the default mode on the same object afterwards equals a fresh forced run (no state leaks out of the checked loop)
    sysm.set_force_accept_step(True)
    fresh = host.System.from_window(win)
    fresh.optimize(6)
    sysm2 = host.System.from_window(win)
    sysm2.set_force_accept_step(False)
    sysm2.set_force_accept_step(True)
    sysm2.optimize(6)
    for f in range(win.n):
        assert np.array_equal(fresh.frame(f)["camToWorld"], sysm2.frame(f)["camToWorld"])
    for x in (sysm, fresh, sysm2, ow, ot):
        x.close()


def test_two_buffer_protocol_keeps_applied_jacobians():
    """sos_ba_linearize fills PointFrameResidual::J only; EFResidual::J (and JpJdF) change at sos_ba_apply_res
    (FS/Residuals.cpp:304-321).  A linearisation at a perturbed state that is NOT applied must leave the applied
    Jacobians, JpJdF rows and the accumulated system untouched."""
    win = synth.make_window("T4")
    ow = hp.oracle_window(win)
    ctx, ba = hp.gpu_backend(win, ow)
    th = np.full(win.n, 512.0, np.float32)
    ow.reset_oob(); ba.reset_oob()
    ow.linearize(th); ba.linearize(th)
    ow.apply_res(); ba.apply_res()
    act = np.flatnonzero((ow.res()["flags"] & 1) != 0)
    J0 = ow.J().copy()
    jp0 = ba.JpJdF().copy()
    a0 = ba.accumulate()
    # perturb the point depths on both sides and linearise WITHOUT applying
    pts = ow.pts()
    idp = (pts["idepth_scaled"] * np.float32(1.05)).astype(np.float32)
    ba.set_state(idepth=idp.copy(), idepth_zero=idp.copy())
    ow.set_state(idepth=idp.copy(), idepth_zero=idp.copy())
    ow.linearize(th)
    g = ba.linearize(th)
    assert np.array_equal(g["newState"].astype(np.int32), ow.new_state())
    Jn = ow.Jnew()
    for r in act[::17]:
        assert hp.jac_equal(ba.jacobian(int(r), which=0), J0[r]), r          # applied side unchanged
        assert hp.jac_equal(ba.jacobian(int(r), which=1), Jn[r]), r          # scratch side = the new linearisation
    assert np.array_equal(ba.JpJdF(), jp0)
    a1 = ba.accumulate()
    for k in ("H_A", "b_A"):
        assert np.array_equal(a0[k], a1[k]), k
    # now apply: the scratch side becomes the applied side
    ow.apply_res(); ba.apply_res()
    act2 = np.flatnonzero((ow.res()["flags"] & 1) != 0)
    J1 = ow.J()
    for r in act2[::17]:
        assert hp.jac_equal(ba.jacobian(int(r), which=0), J1[r]), r
    assert np.array_equal(ba.JpJdF()[act2], ow.JpJdF()[act2])
    ba.close(); ctx.close(); ow.close()
