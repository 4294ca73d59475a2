"""Rolling-window parity (north star: "marginalised poses within a stated float tolerance, active-point index sets
bit-exact"): a 26-frame synthetic sequence, every frame a keyframe, through the whole backend chain of
FullSystem::makeKeyFrame (FS/FullSystem.cpp:783-931) -- raw frame -> undistortion -> pyramid -> trackNewestCoarse ->
traceNewCoarse -> flagFramesForMarginalization -> activatePointsMT -> optimize -> removeOutliers -> flagPointsForRemoval ->
marginalizePointsF -> makeNewTraces -> marginalizeFrame -- once on the device through the C++ facade and once through the
CPU oracle chain (tests/rolling.py), both FREE-RUNNING from the same raw frames: nothing is copied from one chain to the
other after the bootstrap window.

Compared at every keyframe: flagged keyframes, activated points (identity and order), the residual and point index sets,
marginalised / dropped counts, and for every keyframe that LEAVES the window the pose it leaves with; at the end the
marginalisation prior.  Tolerances: index sets identical; poses within POSE_TOL of the oracle chain, or -- yardstick -- no
further from the oracle chain run with fp64 H/b accumulation than 3x the fp32 oracle chain is."""
import numpy as np
import pytest

from tests import rolling

pytestmark = pytest.mark.gpu

POSE_TOL = 2e-5          # stated tolerance on the 12 entries of a marginalised camToWorld (metres / rotation-matrix units)


def _scaled(A, ref):
    s = 1.0 / np.sqrt(np.abs(np.diag(ref)) + 10)
    return A * s[:, None] * s[None, :]


def test_rolling_window_marginalised_poses_and_index_sets():
    sc = rolling.Scenario(n_frames=26)
    dev, orc_, tru = rolling.DeviceChain(sc), rolling.OracleChain(sc), rolling.OracleChain(sc, truth=True)
    boots = [c.bootstrap() for c in (dev, orc_, tru)]
    assert boots[0][1] == boots[1][1] and abs(boots[0][0] - boots[1][0]) <= 1e-5 * boots[1][0]
    assert dev.point_set() == orc_.point_set()
    for f in range(sc.n0):      # the first immature sets: pixel selection + constructors are bit-exact stages
        assert np.array_equal(dev.imm[f]["u"], orc_.imm[f]["u"]) and np.array_equal(dev.imm[f]["energyTH"], orc_.imm[f]["energyTH"])
    worst = dict(pose=0.0, noise=0.0, track=0.0)
    left = 0
    while dev.next_frame < sc.n_frames:
        lg, lo, lt = dev.step(), orc_.step(), tru.step()
        k = lg.frameID
        e_trk = np.abs(lg.tracked_pose - lo.tracked_pose).max()
        worst["track"] = max(worst["track"], e_trk)
        print(f"KF {k}: track |dev-orc| {e_trk:.2e}; flagged {lg.flagged}/{lo.flagged}; activated {len(lg.activated)}/{len(lo.activated)}; "
              f"residuals {len(lg.residual_set)}/{len(lo.residual_set)} (sym.diff {len(lg.residual_set ^ lo.residual_set)}); "
              f"points {len(lg.point_set_after)}/{len(lo.point_set_after)}; marg {lg.marg_points}/{lo.marg_points} drop {lg.dropped_points}/{lo.dropped_points}; "
              f"its {lg.iterations}/{lo.iterations} rmse {lg.rmse:.5f}/{lo.rmse:.5f}")
        assert e_trk < 1e-4
        assert lg.flagged == lo.flagged, k
        assert lg.activated == lo.activated, (k, len(set(lg.activated) ^ set(lo.activated)))
        assert lg.deleted_immature == lo.deleted_immature
        assert lg.iterations == lo.iterations
        assert lg.window_ids == lo.window_ids
        assert lg.residual_set == lo.residual_set, (k, len(lg.residual_set ^ lo.residual_set))
        assert lg.outliers_removed == lo.outliers_removed
        assert (lg.marg_points, lg.dropped_points) == (lo.marg_points, lo.dropped_points)
        assert lg.point_set_after == lo.point_set_after
        assert lg.new_immature == lo.new_immature
        for f in dev.imm:       # the immature containers stay identical in content and order
            assert np.array_equal(dev.imm[f]["u"], orc_.imm[f]["u"]) and np.array_equal(dev.imm[f]["v"], orc_.imm[f]["v"]), (k, f)
            assert np.array_equal(dev.imm[f]["lastTraceStatus"], orc_.imm[f]["lastTraceStatus"]), (k, f)
        assert abs(lg.rmse - lo.rmse) <= 1e-4 * lo.rmse
        # poses of the window after optimize()
        for fid in lg.window_ids:
            e = np.abs(lg.window_poses[fid] - lo.window_poses[fid]).max()
            nz = np.abs(lo.window_poses[fid] - lt.window_poses[fid]).max()
            assert e < max(POSE_TOL, 3 * nz), (k, fid, e, nz)
        # the poses that leave
        assert [f for f, _ in lg.marginalized] == [f for f, _ in lo.marginalized]
        for (fid, pg), (_, po), (_, pt) in zip(lg.marginalized, lo.marginalized, lt.marginalized):
            e_go, e_gt, e_ot = np.abs(pg - po).max(), np.abs(pg - pt).max(), np.abs(po - pt).max()
            worst["pose"], worst["noise"] = max(worst["pose"], e_go), max(worst["noise"], e_ot)
            left += 1
            print(f"   keyframe {fid} leaves: |dev-orc| {e_go:.2e} |dev-truth| {e_gt:.2e} |orc-truth| {e_ot:.2e}")
            assert e_go < max(POSE_TOL, 3 * e_ot), (fid, e_go, e_ot)
            assert e_gt < max(POSE_TOL, 3 * e_ot), (fid, e_gt, e_ot)
        # the prior after this keyframe, in the reference's own Jacobi scaling (OB/EnergyFunctional.cpp:826-832)
        eg = np.abs(_scaled(lg.HM - lt.HM, lt.HM)).max()
        eo = np.abs(_scaled(lo.HM - lt.HM, lt.HM)).max()
        m = np.abs(_scaled(lt.HM, lt.HM)).max()
        assert eg <= 3 * eo + 1e-4 * m, (k, eg, eo, m)
        assert np.abs(lg.bM - lt.bM).max() <= 3 * np.abs(lo.bM - lt.bM).max() + 1e-3 * max(np.abs(lt.bM).max(), 1.0), k
    print(f"{left} keyframes left the window; worst marginalised pose |dev-orc| {worst['pose']:.2e} (oracle fp32 vs fp64-accumulated "
          f"{worst['noise']:.2e}); worst tracked pose difference {worst['track']:.2e}")
    assert left >= 18
    dev.close()
