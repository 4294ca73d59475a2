"""Synthetic window generator: determinism, structure the C-ABI requires, realism knobs."""
import numpy as np

from sos_slam_amd import synth


def test_deterministic_and_structured():
    a, b = synth.make_window("T4"), synth.make_window("T4")
    assert np.array_equal(a.images, b.images) and np.array_equal(a.points, b.points) and np.array_equal(a.resid, b.resid)
    assert a.P == 256 and a.n == 4
    # allPoints order: frames -> points; residuals grouped by point
    assert np.all(np.diff(a.points["host"]) >= 0)
    assert np.all(np.diff(a.resid["point"]) >= 0)
    assert np.all(a.resid["host"] == a.points["host"][a.resid["point"]])
    assert np.all(a.resid["host"] != a.resid["target"])
    # integer pixel positions with room for the pattern (padding 2) and the bilinear tap
    assert np.all(a.points["u"] == np.round(a.points["u"])) and a.points["u"].min() >= 4
    assert 0.1 < a.points["idepth_scaled"].min() and a.points["idepth_scaled"].max() < 3.0
    assert np.allclose(a.HM, a.HM.T) and np.all(np.linalg.eigvalsh(a.HM) > 0)
    assert a.frames["frameID"][0] == 0 and np.all(a.frames["state"][0][:6] == 0)


def test_point_seed_changes_only_points():
    a, b = synth.make_window("T3", point_seed=1), synth.make_window("T3", point_seed=2)
    assert np.array_equal(a.images, b.images) and np.array_equal(a.HM, b.HM)
    assert np.array_equal(a.frames, b.frames)
    assert not np.array_equal(a.points["u"], b.points["u"])


def test_gradient_magnitude_in_target_band():
    """mean |grad I| ~ 8-15 grey levels per pixel at every image size."""
    for name in ("T4", "T6"):
        w = synth.make_window(name)
        gx = np.diff(w.images[0], axis=1)[1:-1]
        gy = np.diff(w.images[0], axis=0)[:, 1:-1]
        g = np.sqrt(gx[:, :gy.shape[1]] ** 2 + gy[:gx.shape[0]] ** 2).mean()
        assert 6.0 < g < 20.0, g
