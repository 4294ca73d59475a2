"""The oracle side of the rolling-window harness on its own (CPU): the Python statement of the keyframe-rate host logic
(tests/rolling.py: OracleChain) around the C restatement keeps a healthy sliding window over a synthetic sequence --
the window stays within setting_maxFrames, keyframes leave in the order the distance score dictates, the poses they
leave with stay close to the rendered trajectory, the prior keeps its dimension."""
import numpy as np

from tests import rolling


def test_oracle_chain_slides_the_window():
    sc = rolling.Scenario(n_frames=14)
    ch = rolling.OracleChain(sc)
    rmse0, _ = ch.bootstrap()
    assert 0 < rmse0 < 5 and ch.n_points() > 300
    left = []
    while ch.next_frame < sc.n_frames:
        lg = ch.step()
        gt = rolling.se3_mul(rolling.se3_inv(sc.poses[lg.frameID]), sc.poses[lg.frameID - 1])
        assert np.abs(lg.tracked_pose - gt).max() < 2e-3                 # trackNewestCoarse found the rendered motion
        assert 0 < lg.rmse < 5 and lg.iterations >= 1
        assert len(lg.window_ids) <= 8 and lg.window_ids == sorted(lg.window_ids)
        assert lg.HM.shape == (4 + 8 * ch.n(), 4 + 8 * ch.n()) and np.allclose(lg.HM, lg.HM.T, rtol=1e-3, atol=1e-3 * np.abs(lg.HM).max())
        assert all(k[3] in lg.window_ids for k in lg.residual_set)
        left += lg.marginalized
    assert len(left) >= 6
    assert [f for f, _ in left] == sorted(set(f for f, _ in left), key=[f for f, _ in left].index)   # each keyframe leaves once
    for fid, pose in left:
        assert np.abs(pose - sc.poses[fid]).max() < 2e-3
    assert ch.n() <= 7 and 0 in ch.window_ids()                          # frame 0 carries the gauge and is never picked by distance


def test_oracle_chain_visual_inertial():
    """The same harness in visual-inertial mode (configs 2-3 in synthetic form): IMU samples generated from the rendered
    trajectory (metric scale 1, a constant gyroscope bias), initializeImu at the fifth keyframe, propagateImuState for every later
    keyframe, the IMU branch of solveSystemF in every optimize(), the expanded prior through marginalizePointsF and the IMU form of
    marginalizeFrame, updateVel / tryTrapScale after every optimisation.  The chain must recover what generated the data."""
    from sos_slam_amd.records import imu_dim
    sc = rolling.Scenario(n_frames=14, vio=True)
    ch = rolling.OracleChain(sc)
    ch.bootstrap()
    left, scales, broken = [], [], 0
    while ch.next_frame < sc.n_frames:
        lg = ch.step()
        v = lg.vio
        assert v["init"] == 1
        scales.append(v["scale"] * 200.0)
        assert abs(scales[-1] - sc.scale_true) < 0.05                   # metric scale from the accelerometer
        bg = np.array([x[3:6] for x in v["states"].values()])          # gyroscope bias (SCALE_BG = 1) of the keyframes that stay
        assert np.abs(bg - sc.bias_g).max() < 2e-3
        assert v["HMi"].shape == (imu_dim(ch.n()),) * 2
        assert np.isfinite(v["HMi"]).all() and np.isfinite(v["bMi"]).all()
        gt = rolling.se3_mul(rolling.se3_inv(sc.poses[lg.frameID]), sc.poses[lg.frameID - 1])
        assert np.abs(lg.tracked_pose - gt).max() < 2e-3
        for fid in lg.window_ids:
            assert np.abs(lg.window_poses[fid] - sc.poses[fid]).max() < 5e-3 + 0.03 * np.abs(sc.poses[fid][9:]).max()   # monocular gauge drift
        left += lg.marginalized
        # shell->trackingRef == the shell of the window predecessor (OB/EnergyFunctional.cpp:318, :350) is what the records must say: the
        # successor of a keyframe that left from the MIDDLE of the window keeps its tracking reference and loses its spline terms
        ids = ch.window_ids()
        recs, _keep = ch.vio_records(ids)
        for i, fid in enumerate(ids):
            want = 1 if (i > 0 and ids[i - 1] == fid - 1) else 0       # every frame is a keyframe here: the tracking reference of k is k - 1
            assert recs[i].trackingRefIsPrev == want, (ids, fid)
        broken += sum(1 for i, fid in enumerate(ids) if i > 0 and ids[i - 1] != fid - 1)
    assert len(left) >= 5 and broken > 0                                # middle keyframes did leave the window
    assert np.abs(np.array(scales[-4:]) - sc.scale_true).max() < 0.05
    vel_true = (sc.poses[sc.n_frames - 1][9:] - sc.poses[sc.n_frames - 2][9:]) / sc.dt
    assert np.abs(ch.shells[sc.n_frames - 1]["vel"] - vel_true).max() < 0.2 * np.abs(vel_true).max() + 0.05


def test_oracle_chain_stereo_inertial():
    """stereo-inertial: FullSystem::optimizeScale on the stereo partner of every keyframe gives the metric scale (trapped at the
    first keyframe from the seven guesses), setting_enable_scale_opt semantics elsewhere"""
    sc = rolling.Scenario(n_frames=10, vio=True, stereo=True)
    ch = rolling.OracleChain(sc)
    ch.bootstrap()
    assert ch.scale_log[0][3] == [1, 0] and abs(ch.scale_log[0][1] - 1.0) < 0.02
    while ch.next_frame < sc.n_frames:
        lg = ch.step()
        k, new_scale, err, st = ch.scale_log[-1]
        assert k == lg.frameID and st == [1, 0] and 0 < err < sc.scale_opt_thres
        assert abs(new_scale - sc.scale_true) < 0.02 and abs(lg.vio["scale"] * 200 - new_scale) < 1e-6
        assert lg.vio["trapped"] == 1 and lg.vio["init"] == 1


def test_ensemble_summary_criteria():
    """tests/rolling_ensemble.summarize (the criteria of the GPU ensemble tests) on hand-made runs: hard decisions always count, the
    geometric mean and the maximum are compared per quantity with their own factors, set differences of zero count as one."""
    from tests import rolling_ensemble as re_

    def run(seed, d, n, hard=()):
        r = dict(seed=seed, vio=False, keyframes=10, left=5, hard=list(hard), its_mismatch=0)
        for k in ("leave", "win"):
            r["d_" + k], r["n_" + k] = d, n
        for k in ("res", "act", "pts"):
            r["d_" + k], r["n_" + k] = 0, 0
        return r

    ok = re_.summarize([run(1, 2e-5, 1e-5), run(2, 1e-6, 4e-5), run(3, 3e-5, 2e-5)])
    assert not ok["violations"] and ok["left_total"] == 15 and ok["keyframes_total"] == 30
    assert abs(ok["leave"]["max_ratio"] - 0.75) < 1e-12 and ok["res"]["geo_ratio"] == 1.0
    geo = re_.summarize([run(1, 4e-5, 1e-5), run(2, 4e-5, 1e-5), run(3, 4e-5, 5e-5)])          # mean 2.3x above... max within 3x
    assert not geo["violations"]
    bad_geo = re_.summarize([run(1, 4e-5, 1e-5), run(2, 4e-5, 1e-5), run(3, 4e-5, 1e-5)])
    assert {v[:2] for v in bad_geo["violations"]} == {("leave", "geometric mean"), ("leave", "maximum"), ("win", "geometric mean"), ("win", "maximum")}
    bad_max = re_.summarize([run(1, 1e-6, 1e-5), run(2, 1e-6, 1e-5), run(3, 4e-5, 1e-5)])      # one outlier: the tail criterion
    assert {v[:2] for v in bad_max["violations"]} == {("leave", "maximum"), ("win", "maximum")}
    hard = re_.summarize([run(1, 1e-6, 1e-5, hard=[(7, "flagged", [1], [2])])])
    assert hard["violations"] and hard["violations"][0][:2] == (1, "hard decision")
