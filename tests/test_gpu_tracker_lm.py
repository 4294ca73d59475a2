"""The device-resident LM loop of trackNewestCoarse / PoseEstimator::estimate (sos_tracker_track: one launch, K hypotheses side by
side) and the hypothesis loop of FullSystem::trackNewCoarse (FS/FullSystem.cpp:150-283).

Three comparisons:
  * device loop vs the host loop of the facade on the same device residual passes: same decisions (evaluation counts,
    levels, acceptance), poses equal up to the rounding of the 8x8 solve and of sin / cos;
  * device loop vs the oracle's trackNewestCoarse with the fp64-accumulated oracle as the yardstick (as
    tests/test_gpu_tracker_fullsize.py does for the host loop -- that file now runs the device loop too, it is the default);
  * batched hypothesis loop vs the oracle's one-by-one loop: same winner, same number of tries consumed, achievedRes and pose
    within the yardstick; batch sizes 1 / 5 / 16 give bit-identical outcomes (the sequential decisions are replayed)."""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import Calib
from sos_slam_amd.synth import se3_exp12 as se3_exp, se3_mul12 as se3_mul, se3_inv12 as se3_inv
from tests import helpers as hp

pytestmark = pytest.mark.gpu


def _rel_pose(win, k=0):
    ref, new = win.frames[win.n - 1]["camToWorld"], win.extra_poses[k]
    Rr, tr, Rn, tn = ref[:9].reshape(3, 3), ref[9:], new[:9].reshape(3, 3), new[9:]
    return np.concatenate([(Rn.T @ Rr).reshape(-1), Rn.T @ (tr - tn)])


def make_rig(name, **window_args):
    """The tracker rig of this file (device facade + oracle trackers on the optimised window `name`); a generator, also used by
    tools/tracker_sweep.py with other seeds."""
    from sos_slam_amd import host
    win = synth.make_window(name, extra_frames=2, **window_args)
    ow = hp.oracle_window(win)
    ow.optimize(6, nthreads=6)
    sysm = host.System.from_window(win)
    sysm.optimize(6)
    res = ow.res()
    sel = (res["target"] == win.n - 1) & ((res["flags"] & 0x101) == 1) & (res["state_state"] == synth.RES_IN)
    c = ow.center()[sel]
    hdi = ow.point_field("HdiF")[res["point"][sel]]
    calib = Calib.from_K(ow.calib_value_scaled())
    trackers = []
    for truth in (False, True):
        t = orc.OracleTracker(win.params, win.w, win.h)
        t.set_truth_mode(truth)
        t.set_ref(calib, ow.dI[win.n - 1], c[:, 0], c[:, 1], c[:, 2], hdi)
        trackers.append(t)
    ht = host.HostTracker(sysm)
    ht.set_ref_raw(c[:, 0], c[:, 1], c[:, 2], hdi)
    new_dI, _ = orc.make_images(win.extra_images[0])
    st = ow.frame(win.n - 1)["state"]
    yield dict(key=name, win=win, ow=ow, sysm=sysm, ot=trackers[0], ott=trackers[1], ht=ht, new_dI=new_dI,
               new_slot=sysm.upload_image(win.extra_images[0]), ref_aff=np.array([st[6] * 10.0, st[7] * 1000.0]),
               levels=len(trackers[0].pc_n))
    ht.close()
    sysm.close()
    ow.close()


@pytest.fixture(scope="module", params=["W7", "T6"])
def rig(request):
    yield from make_rig(request.param)


PERTURB = [np.zeros(6), np.array([0.004, -0.003, 0.002, 0.002, -0.002, 0.001]), np.array([-0.01, 0.008, 0.005, -0.004, 0.003, 0.004]),
           np.array([0.03, 0.02, -0.02, 0.01, -0.012, 0.008])]


@pytest.mark.parametrize("pi", range(len(PERTURB)))
def test_device_loop_equals_host_loop(rig, pi):
    ht, win, levels = rig["ht"], rig["win"], rig["levels"]
    Tinit = se3_mul(se3_exp(PERTURB[pi]), _rel_pose(win))
    out = {}
    for mode in (True, False):
        ht.set_device_lm(mode)
        out[mode] = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1) + (ht.last_evals(),)
    ht.set_device_lm(True)
    (ok_d, Td, ad, ld, fd, ev_d), (ok_h, Th, ah, lh, fh, ev_h) = out[True], out[False]
    print(f"{rig['key']} perturbation {pi}: evaluations device {ev_d} host {ev_h}; |Td-Th| {np.abs(Td - Th).max():.3g}; res {ld[:levels]}")
    assert ok_d == ok_h
    assert ev_d == ev_h and ev_d >= levels
    assert np.abs(Td - Th).max() < 1e-9
    assert np.abs(ad - ah).max() < 1e-9
    assert np.allclose(ld[:levels], lh[:levels], rtol=1e-7, atol=0, equal_nan=True)
    assert np.allclose(fd, fh, rtol=1e-6)


def test_device_loop_against_truth_yardstick(rig):
    ht, win, ot, ott, levels = rig["ht"], rig["win"], rig["ot"], rig["ott"], rig["levels"]
    T0 = _rel_pose(win)
    Tinit = se3_mul(se3_exp(PERTURB[1]), T0)
    ok_o, To, ao, lo, fo = ot.track(rig["new_dI"], 1.0, 1.0, rig["ref_aff"], Tinit, np.zeros(2), levels - 1)
    ok_t, Tt, at, lt, ft = ott.track(rig["new_dI"], 1.0, 1.0, rig["ref_aff"], Tinit, np.zeros(2), levels - 1)
    ok_g, Tg, ag, lg, fg = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1)
    assert ok_o and ok_t and ok_g
    e_go, e_gt, e_ot = np.abs(Tg - To).max(), np.abs(Tg - Tt).max(), np.abs(To - Tt).max()
    print(f"{rig['key']}: |Tg-To| {e_go:.3g} |Tg-Tt| {e_gt:.3g} |To-Tt| {e_ot:.3g}")
    assert e_go < 5e-5
    assert e_gt <= max(2 * e_ot, 1e-5)
    assert np.allclose(lg[:levels], lo[:levels], rtol=1e-4)


def test_abort_threshold_and_cut(rig):
    """minResForAbort: the device stops a hypothesis at the level where lastResiduals > 1.5 * minRes, as the host loop does."""
    ht, win, levels = rig["ht"], rig["win"], rig["levels"]
    Tinit = se3_mul(se3_exp(PERTURB[2]), _rel_pose(win))
    ok, T, a, lr, fl = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1)
    assert ok
    for cut_lvl in range(levels - 1, -1, -1):
        mr = np.full(5, np.nan)
        mr[cut_lvl] = lr[cut_lvl] / 1.5 * 0.999      # just below: this level aborts
        res = {}
        for mode in (True, False):
            ht.set_device_lm(mode)
            res[mode] = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1, minRes=mr)
        ht.set_device_lm(True)
        okd, Td, _, ld, _ = res[True]
        okh, Th, _, lh, _ = res[False]
        assert not okd and not okh
        assert np.array_equal(np.isnan(ld), np.isnan(lh))
        assert np.all(np.isnan(ld[:cut_lvl])) and np.all(np.isfinite(ld[cut_lvl:levels]))
        assert np.allclose(ld[cut_lvl:levels], lh[cut_lvl:levels], rtol=1e-7)
        assert np.array_equal(Td, Tinit) and np.array_equal(Th, Tinit)   # an aborted try leaves lastToNew_out untouched


def _history(win):
    """slast_2_sprelast and lastF_2_slast for the frame being tracked (extra frame 1), from the rendered trajectory:
    sprelast = newest keyframe, slast = extra frame 0, lastF (tracking reference) = newest keyframe."""
    kf, slast = win.frames[win.n - 1]["camToWorld"], win.extra_poses[0]
    slast_2_sprelast = se3_mul(se3_inv(kf), slast)
    lastF_2_slast = se3_mul(se3_inv(slast), kf)
    return slast_2_sprelast, lastF_2_slast


def test_try_list_matches_oracle(rig):
    a, b = _history(rig["win"])
    tg = rig["ht"].make_tries(a, b)
    to = orc.make_track_tries(a, b)
    assert tg.shape == to.shape == (5 + 78, 12)
    assert np.abs(tg - to).max() < 1e-13
    imu = se3_mul(se3_exp(np.array([1e-3, 0, 0, 0, 1e-3, 0])), tg[0])
    assert np.abs(rig["ht"].make_tries(a, b, imu12=imu) - orc.make_track_tries(a, b, imu=imu)).max() < 1e-13
    assert len(rig["ht"].make_tries(a, b, poses_valid=False)) == 1


@pytest.mark.parametrize("case", ["first_wins", "exhaustive", "bad_start"])
def test_hypothesis_loop_matches_oracle(rig, case):
    """first_wins: the constant-motion try is below lastCoarseRMSE * 1.5 -> one try.  exhaustive: lastCoarseRMSE so small that
    nothing stops the loop -> all 83 tries are consumed, most aborted on the coarsest level by the achievedRes test.
    bad_start: the frame history is wrong by 0.5 rad (as after a tracking hiccup), the loop has to find a working try."""
    from sos_slam_amd import host
    win, sysm, ht, ot, levels = rig["win"], rig["sysm"], rig["ht"], rig["ot"], rig["levels"]
    new_dI, _ = orc.make_images(win.extra_images[1])
    slot = sysm.upload_image(win.extra_images[1])
    a, b = _history(win)
    last_rmse = np.full(5, 100.0)
    if case == "exhaustive":
        last_rmse = np.full(5, 1e-6)
    if case == "bad_start":
        a = se3_mul(se3_exp(np.array([0.3, -0.2, 0.1, 0.05, 0.5, -0.05])), a)
        last_rmse = np.full(5, 5.0)
    tries = ht.make_tries(a, b)
    ref = orc.track_new_coarse(ot, new_dI, 1.0, 1.0, rig["ref_aff"], tries, np.zeros(2), levels - 1, last_rmse)
    got = {bs: ht.track_hypotheses(slot, 1.0, tries, np.zeros(2), levels - 1, last_rmse, batch=bs) for bs in (1, 5, 16)}
    g = got[16]
    print(f"{rig['key']} {case}: chosen {g['chosen']}/{ref['chosen']}, tries consumed {g['tryIterations']}/{ref['tryIterations']} "
          f"(evaluated {g['evaluated']}), achievedRes {g['achievedRes'][:levels]} vs {ref['achievedRes'][:levels]}")
    for bs in (1, 5):
        for k in ("lastF_2_fh", "aff", "achievedRes", "flow"):
            assert np.array_equal(got[bs][k], g[k], equal_nan=True), (bs, k)
        assert (got[bs]["chosen"], got[bs]["tryIterations"], got[bs]["haveOneGood"]) == (g["chosen"], g["tryIterations"], g["haveOneGood"])
    assert got[1]["evaluated"] == g["tryIterations"]
    assert g["haveOneGood"] == ref["haveOneGood"]
    assert g["tryIterations"] == ref["tryIterations"]
    if case != "exhaustive":
        assert g["chosen"] == ref["chosen"]
    # (exhaustive: dozens of tries converge to the same optimum and the winner is the one whose fp32 residual sum is lowest by
    # a few 1e-7 -- a knife edge between two summation orders; what must agree is the optimum itself)
    if case == "first_wins":
        assert g["tryIterations"] == 1 and g["chosen"] == 0 and g["evaluated"] == 1
    if case == "exhaustive":
        assert g["tryIterations"] == len(tries)
    assert np.allclose(g["achievedRes"][:levels], ref["achievedRes"][:levels], rtol=2e-4, equal_nan=True)
    assert np.abs(g["lastF_2_fh"] - ref["lastF_2_fh"]).max() < 5e-5
    assert np.abs(g["aff"] - ref["aff"]).max() < 2e-3
    truth = _rel_pose(win, 1)
    assert np.abs(g["lastF_2_fh"] - truth).max() < 5e-3
    # the host-loop version of the same function (one try at a time through trackNewestCoarse) agrees with the batched one
    ht.set_device_lm(False)
    hl = ht.track_hypotheses(slot, 1.0, tries, np.zeros(2), levels - 1, last_rmse, batch=1)
    ht.set_device_lm(True)
    assert hl["tryIterations"] == g["tryIterations"]
    if case != "exhaustive":
        assert hl["chosen"] == g["chosen"]
        assert np.abs(hl["lastF_2_fh"] - g["lastF_2_fh"]).max() < 1e-8
    assert np.allclose(hl["achievedRes"][:levels], g["achievedRes"][:levels], rtol=1e-5, equal_nan=True)
    sysm.release_image(slot)


@pytest.mark.parametrize("s0", [1.0, 1.3, 0.8, 2.5])
def test_scale_loop_on_device_equals_host_loop(rig, s0):
    """ScaleOptimizer::optimizeScale as one launch (MODE 1 of the same kernel) against the host loop around device passes: all
    arithmetic of the loop is fp32 on both sides, the sums are the same sums -> identical scale and evaluation count; and
    against the oracle with its fp64-accumulated run as the yardstick."""
    win, sysm, ht, levels = rig["win"], rig["sysm"], rig["ht"], rig["levels"]
    st_dI, _ = orc.make_images(win.extra_images[1])
    slot = sysm.upload_image(win.extra_images[1])
    K1 = rig["ow"].calib_value_scaled().astype(np.float32)
    out = {}
    for mode in (True, False):
        ht.set_device_lm(mode)
        out[mode] = ht.optimize_scale(slot, win.stereo_tfm, K1, s0, levels - 1) + (ht.last_evals(),)
    ht.set_device_lm(True)
    (rd, sd, ed), (rh, sh, eh) = out[True], out[False]
    ro, so = rig["ot"].optimize_scale(st_dI, win.stereo_tfm, K1, s0, levels - 1)
    rt, stt = rig["ott"].optimize_scale(st_dI, win.stereo_tfm, K1, s0, levels - 1)
    print(f"{rig['key']} s0={s0}: scale device {sd:.7f} host loop {sh:.7f} oracle {so:.7f} truth {stt:.7f}; evaluations {ed}/{eh}")
    assert (sd, ed) == (sh, eh)
    assert rd == rh
    assert abs(sd - so) <= max(2 * abs(so - stt), 2e-5 * so)
    assert rd == pytest.approx(ro, rel=1e-4)
    sysm.release_image(slot)


def test_optimize_scale_keyframe_function(rig):
    """FullSystem::optimizeScale (FS/FullSystem.cpp:1117-1177): the seven scale guesses side by side in one launch until the
    scale is trapped, then single runs from the tracking reference's scale; rejection above setting_scale_opt_thres and the
    re-initialisation after six failures -- against the oracle's one-by-one restatement, and against the facade's own host loop."""
    win, sysm, ht, levels = rig["win"], rig["sysm"], rig["ht"], rig["levels"]
    st_dI, _ = orc.make_images(win.extra_images[1])
    slot = sysm.upload_image(win.extra_images[1])
    K1 = rig["ow"].calib_value_scaled().astype(np.float32)
    sg, so, sh = [0, 0], [0, 0], [0, 0]
    # not trapped: seven guesses
    ng, eg = ht.optimize_scale_kf(slot, win.stereo_tfm, K1, 1.0, levels - 1, 12.0, sg)
    no, eo = orc.optimize_scale_kf(rig["ot"], st_dI, win.stereo_tfm, K1, 1.0, levels - 1, 12.0, so)
    ht.set_device_lm(False)
    nh, eh = ht.optimize_scale_kf(slot, win.stereo_tfm, K1, 1.0, levels - 1, 12.0, sh)
    ht.set_device_lm(True)
    print(f"{rig['key']}: seven guesses -> scale {ng:.6f} / oracle {no:.6f} / host loop {nh:.6f}; error {eg:.5f} / {eo:.5f}; state {sg} {so}")
    assert sg == so == sh == [1, 0]
    assert (ng, eg) == (nh, eh)                                   # all-fp32 loop: device == host loop bit for bit
    assert abs(ng - no) <= 1e-4 * no and eg == pytest.approx(eo, rel=1e-4)
    assert abs(ng - 1.0) < 0.05
    # trapped: one run from the reference's scale
    ng2, eg2 = ht.optimize_scale_kf(slot, win.stereo_tfm, K1, ng, levels - 1, 12.0, sg)
    no2, eo2 = orc.optimize_scale_kf(rig["ot"], st_dI, win.stereo_tfm, K1, no, levels - 1, 12.0, so)
    assert sg == so == [1, 0] and abs(ng2 - no2) <= 1e-4 * no2 and eg2 == pytest.approx(eo2, rel=1e-4)
    # an impossible threshold: rejected six times -> the scale is released and the guesses come back
    for k in range(6):
        ng3, _ = ht.optimize_scale_kf(slot, win.stereo_tfm, K1, ng, levels - 1, 1e-3, sg)
        no3, _ = orc.optimize_scale_kf(rig["ot"], st_dI, win.stereo_tfm, K1, no, levels - 1, 1e-3, so)
        assert ng3 == no3 == -1.0 and sg == so
    assert sg == [0, 6]
    assert ht.optimize_scale_kf(slot, win.stereo_tfm, K1, 1.0, levels - 1, 0.0, sg)[0] == 1.0      # setting_scale_opt_thres <= 0
    sysm.release_image(slot)


@pytest.mark.parametrize("name", ["T3", "T4", "T6"])
def test_degenerate_templates_and_initial_poses(name):
    """Small templates (96 x 64: two pyramid levels, a handful of 256-pixel chunks) and initial poses from exact to absurd -- 0.3 rad off,
    the camera turned around, 50 units away: with nothing left to warp every sum is zero and the level residuals are NaN.  The device loop,
    the host loop around device passes and the oracle port take the same number of evaluations, report the same `ok`, the same NaN pattern,
    and the same pose where there is one to find."""
    gen = make_rig(name)
    rig = next(gen)
    ht, win, ot, levels = rig["ht"], rig["win"], rig["ot"], rig["levels"]
    T0 = _rel_pose(win)
    perts = {"exact": np.zeros(6), "large": np.array([0.3, 0.2, -0.2, 0.1, -0.12, 0.08]), "behind": np.array([0, 0, 0, 0, 3.1, 0.0]),
             "far": np.array([50.0, 0, 0, 0, 0, 0])}
    for pn, p in perts.items():
        Tinit = se3_mul(se3_exp(p), T0)
        out = {}
        for mode in (True, False):
            ht.set_device_lm(mode)
            out[mode] = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1) + (ht.last_evals(),)
        ht.set_device_lm(True)
        ok_o, To, ao, lo, fo = ot.track(rig["new_dI"], 1.0, 1.0, rig["ref_aff"], Tinit, np.zeros(2), levels - 1)
        (ok_d, Td, ad, ld, fd, ev_d), (ok_h, Th, ah, lh, fh, ev_h) = out[True], out[False]
        print(f"{name} ({levels} levels) {pn}: ok {ok_d}/{ok_h}/{bool(ok_o)}, evaluations {ev_d}/{ev_h}, residuals {ld[:levels]}")
        assert ok_d == ok_h == bool(ok_o) and ev_d == ev_h
        assert np.allclose(Td, Th, rtol=0, atol=1e-9) and np.allclose(ld[:levels], lh[:levels], rtol=1e-6, equal_nan=True)
        assert np.array_equal(np.isnan(ld[:levels]), np.isnan(lo[:levels]))
        if pn in ("exact", "large"):
            assert np.isfinite(ld[:levels]).all() and np.abs(Td - To).max() < 1e-4
        else:
            assert np.isnan(ld[:levels]).all()
    gen.close()
