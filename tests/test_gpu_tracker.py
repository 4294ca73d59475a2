"""GPU parity of the CoarseTracker / ScaleOptimizer path (G1-G6 of SURVEY.md 8(a)) against the oracle.

Template point clouds (set_ref) and the integer statistics of calc_res are bit-exact; float sums (energies,
flow indicators, the 8x8 Gauss-Newton system) use fixed-shape trees on the device instead of the reference's
4-lane SSE order: relative tolerance 1e-5.  The LM loops (host) must land on the same pose.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import Calib
from tests import helpers as hp

pytestmark = pytest.mark.gpu
SUM_TOL = 1e-5


@pytest.fixture(scope="module")
def rig():
    from sos_slam_amd import host
    win = synth.make_window("T6", extra_frames=2)
    ow = hp.oracle_window(win)
    ow.optimize(6)
    sysm = host.System.from_window(win)
    sysm.optimize(6)
    # reference points of the oracle: residuals to the newest keyframe that are still IN
    res, pts = ow.res(), ow.pts()
    sel = (res["target"] == win.n - 1) & ((res["flags"] & 0x101) == 1) & (res["state_state"] == synth.RES_IN)
    c = ow.center()[sel]
    hdi = ow.point_field("HdiF")[res["point"][sel]]
    calib = Calib.from_K(ow.calib_value_scaled())
    ot = orc.OracleTracker(win.params, win.w, win.h)
    ref_dI = ow.dI[win.n - 1]
    pc_n = ot.set_ref(calib, ref_dI, c[:, 0], c[:, 1], c[:, 2], hdi)
    ht = host.HostTracker(sysm)
    ht.set_ref_raw(c[:, 0], c[:, 1], c[:, 2], hdi)   # (every test below may run on its own: xdist, -k)
    new_dI, _ = orc.make_images(win.extra_images[0])
    st_dI, _ = orc.make_images(win.extra_images[1])
    new_slot = sysm.upload_image(win.extra_images[0])
    st_slot = sysm.upload_image(win.extra_images[1])
    yield dict(win=win, ow=ow, sysm=sysm, ot=ot, ht=ht, c=c, hdi=hdi, pc_n=pc_n, calib=calib, new_dI=new_dI, st_dI=st_dI,
               new_slot=new_slot, st_slot=st_slot)
    ht.close()
    sysm.close()


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def test_set_ref_point_clouds_bit_exact(rig):
    c, hdi = rig["c"], rig["hdi"]
    pc_g = rig["ht"].set_ref_raw(c[:, 0], c[:, 1], c[:, 2], hdi)
    levels = len(rig["pc_n"])
    assert np.array_equal(pc_g[:levels], rig["pc_n"])
    assert rig["pc_n"][0] > 50
    dev = rig["ht"].device()
    for lvl in range(levels):
        g, o = dev.get_pc(lvl), rig["ot"].get_pc(lvl)
        for a, b in zip(g, o):
            assert np.array_equal(a, b), lvl


def test_set_ref_from_system_state(rig):
    """setCoarseTrackingRef fed from the facade's own optimize() results: same template size up to the fp32
    noise of the two GN runs (a point may round to a neighbouring pixel)."""
    pc_g = rig["ht"].set_ref()
    levels = len(rig["pc_n"])
    assert np.all(np.abs(pc_g[:levels] - rig["pc_n"]) <= np.maximum(3, 0.01 * rig["pc_n"]))
    c, hdi = rig["c"], rig["hdi"]
    rig["ht"].set_ref_raw(c[:, 0], c[:, 1], c[:, 2], hdi)  # back to the shared template for the other tests


def _pose(win):
    """refToNew for extra frame 0: inverse(newPose) * refPose, slightly perturbed."""
    ref = win.frames[win.n - 1]["camToWorld"]
    new = win.extra_poses[0]
    Rr, tr = ref[:9].reshape(3, 3), ref[9:]
    Rn, tn = new[:9].reshape(3, 3), new[9:]
    R = Rn.T @ Rr
    t = Rn.T @ (tr - tn)
    return np.concatenate([R.reshape(-1), t])


def test_calc_res_and_gs(rig):
    win, ot, dev = rig["win"], rig["ot"], rig["ht"].device()
    T = _pose(win)
    K = rig["ow"].calib_value_scaled()
    for lvl in range(len(rig["pc_n"])):
        fx, fy = K[0] / 2 ** lvl, K[1] / 2 ** lvl
        cx, cy = (K[2] + 0.5) / 2 ** lvl - 0.5, (K[3] + 0.5) / 2 ** lvl - 0.5
        Ki = np.array([[1 / np.float32(fx), 0, -np.float32(cx) / np.float32(fx)],
                       [0, 1 / np.float32(fy), -np.float32(cy) / np.float32(fy)], [0, 0, 1]], dtype=np.float32)
        RKi = (T[:9].reshape(3, 3).astype(np.float32) @ Ki).astype(np.float32)
        t = T[9:].astype(np.float32)
        aff = np.array([1.01, -0.5], np.float32)
        for cutoff in (20.0, 3.0):
            ro = ot.calc_res(lvl, rig["new_dI"][lvl], RKi, t, aff, cutoff)
            rg = dev.calc_res(lvl, rig["new_slot"], RKi, t, aff, cutoff)
            assert rg[1] == ro[1] and rg[1] > 0                       # numTermsInE
            assert rg[5] == pytest.approx(ro[5], rel=1e-6, abs=1e-7)  # saturated ratio (integer counts)
            assert _rel(rg[0], ro[0]) < SUM_TOL
            if lvl == 0:
                assert _rel(rg[2], ro[2]) < 1e-4 and _rel(rg[4], ro[4]) < 1e-4
            Ho, bo = ot.calc_gs(lvl, float(aff[0]), 0.25)
            Hg, bg = dev.calc_gs(lvl, float(aff[0]), 0.25)
            assert hp.relerr(Hg, Ho) < SUM_TOL and hp.relerr(bg, bo) < 1e-4
    # saturation path: a very small cutoff saturates almost everything
    ro = ot.calc_res(0, rig["new_dI"][0], RKi0(win, K), T[9:].astype(np.float32), np.array([1, 0], np.float32), 0.01)
    rg = dev.calc_res(0, rig["new_slot"], RKi0(win, K), T[9:].astype(np.float32), np.array([1, 0], np.float32), 0.01)
    assert rg[1] == ro[1] and rg[5] > 0.9 and rg[5] == pytest.approx(ro[5], rel=1e-6)


def RKi0(win, K):
    T = _pose(win)
    Ki = np.array([[1 / np.float32(K[0]), 0, -np.float32(K[2]) / np.float32(K[0])],
                   [0, 1 / np.float32(K[1]), -np.float32(K[3]) / np.float32(K[1])], [0, 0, 1]], dtype=np.float32)
    return (T[:9].reshape(3, 3).astype(np.float32) @ Ki).astype(np.float32)


def test_track_newest_coarse(rig):
    win, ot, ht = rig["win"], rig["ot"], rig["ht"]
    T0 = _pose(win)
    # start from a perturbed pose so the LM loop has work to do
    from tests.test_oracle_math import se3_exp, se3_mul
    Tinit = se3_mul(se3_exp(np.array([0.004, -0.003, 0.002, 0.002, -0.002, 0.001])), T0)
    ref_aff = np.array([rig["ow"].frame(win.n - 1)["state"][6] * 10.0, rig["ow"].frame(win.n - 1)["state"][7] * 1000.0])
    levels = len(rig["pc_n"])
    ok_o, To, ao, lo, fo = ot.track(rig["new_dI"], 1.0, 1.0, ref_aff, Tinit, np.zeros(2), levels - 1)
    ok_g, Tg, ag, lg, fg = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1)
    assert ok_o and ok_g
    assert np.abs(Tg - To).max() < 2e-4
    assert np.abs(Tg - T0).max() < 5e-3            # converged to the true relative pose
    assert np.allclose(lg[:levels], lo[:levels], rtol=2e-3)
    assert np.allclose(ag, ao, atol=2e-3)


def test_calc_res_scale_and_gs_scale(rig):
    """ScaleOptimizer::calcResScale / calcGSSSEScale (FS/ScaleOptimizer.cpp:232-330, 332-430) per call."""
    ot, dev = rig["ot"], rig["ht"].device()
    K = rig["ow"].calib_value_scaled()
    t = rig["win"].stereo_tfm[9:].astype(np.float32)
    for lvl in range(len(rig["pc_n"])):
        fx, fy = np.float32(K[0] / 2 ** lvl), np.float32(K[1] / 2 ** lvl)
        cx, cy = np.float32((K[2] + 0.5) / 2 ** lvl - 0.5), np.float32((K[3] + 0.5) / 2 ** lvl - 0.5)
        Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]], dtype=np.float32)
        K1 = np.array([fx, fy, cx, cy], np.float32)
        for scale in (1.0, 1.3, 0.8):
            ro = ot.calc_res_scale(lvl, rig["st_dI"][lvl], Ki, t, K1, scale, 20.0)
            rg = dev.calc_res_scale(lvl, rig["st_slot"], Ki, t, K1, scale, 20.0)
            assert rg[1] == ro[1] and rg[1] > 0, (lvl, scale, rg, ro)
            assert _rel(rg[0], ro[0]) < SUM_TOL
            assert rg[5] == pytest.approx(ro[5], rel=1e-6, abs=1e-7)
            Ho, bo = ot.calc_gs_scale(lvl, t, K1, scale)
            Hg, bg = dev.calc_gs_scale(lvl, t, K1, scale)
            assert _rel(Hg, Ho) < SUM_TOL and abs(bg - bo) < 1e-4 * max(abs(bo), 1e-3 * abs(Ho)), (lvl, scale, Hg, Ho, bg, bo)


def test_optimize_scale(rig):
    win, ot, ht = rig["win"], rig["ot"], rig["ht"]
    tfm = win.stereo_tfm  # cam1 = cam0 shifted by the baseline  ->  p1 = p0 - offset
    K1 = rig["ow"].calib_value_scaled().astype(np.float32)
    levels = len(rig["pc_n"])
    for s0 in (1.0, 1.3, 0.8):
        ro, so = ot.optimize_scale(rig["st_dI"], tfm, K1, s0, levels - 1)
        rg, sg = ht.optimize_scale(rig["st_slot"], tfm, K1, s0, levels - 1)
        assert abs(sg - so) < 2e-3 * so
        assert rg == pytest.approx(ro, rel=2e-3)
        assert abs(sg - 1.0) < 0.05              # metric depths: the stereo scale is 1
