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
