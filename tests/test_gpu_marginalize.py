"""Keyframe-rate marginalisation through the C++ facade (F14 marginalizePointsF, F15 marginalizeFrame) against
the oracle's host-level restatement, after both ran the same 6-iteration optimize()."""
import numpy as np
import pytest

from sos_slam_amd import synth
from tests import helpers as hp

pytestmark = pytest.mark.gpu


def _scaled_err(a, b):
    """max |a-b| in the Jacobi-scaled metric sqrt(|diag|+10) the reference itself uses for HM
    (OB/EnergyFunctional.cpp:826-832)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.ndim == 2:
        s = 1.0 / np.sqrt(np.abs(np.diag(b)) + 10)
        return np.abs((a - b) * s[:, None] * s[None, :]).max(), np.abs(b * s[:, None] * s[None, :]).max()
    return np.abs(a - b).max(), np.abs(b).max()


def _prep_oracle(win, truth):
    ow = hp.oracle_window(win)
    ow.set_truth_mode(truth)
    ow.optimize(6)
    return ow


def _yardstick(g, o, t, what, fac=2.0):
    """The two fp32 runs (device, reference restatement) end optimize() at states ~1e-6 apart, and b = M_b - M_b,sc
    amplifies that; the bar is the one of tests/test_gpu_optimize.py: the device may be at most twice as far from
    the fp64-accumulated run as the fp32 restatement of the reference is (plus fp32 round-off of the sums)."""
    eg, m = _scaled_err(g, t)
    eo, _ = _scaled_err(o, t)
    assert eg <= fac * eo + 1e-5 * m, (what, eg, eo, m)
    assert eg < 2e-2 * m, (what, eg, m)


@pytest.mark.parametrize("name", ["T4", "T6"])
def test_marginalize_points_and_frame(name):
    from sos_slam_amd import host
    win = synth.make_window(name)
    ow, owt = _prep_oracle(win, False), _prep_oracle(win, True)
    sysm = host.System.from_window(win)
    sysm.optimize(6)
    # points hosted in frame 0 that still have residuals after the final linearizeAll(true)
    ids = sysm.point_ids()
    res = ow.res()
    has_res = np.zeros(win.P, bool)
    live = (res["flags"] & 0x100) == 0
    has_res[res["point"][live]] = True
    alive0 = np.flatnonzero(has_res & (win.points["host"] == 0)).astype(np.int32)
    assert np.isin(alive0, ids).all()
    empty0 = np.flatnonzero(~has_res & (win.points["host"] == 0)).astype(np.int32)
    sel, rest = alive0[::2], alive0[1::2]
    HM0, bM0 = ow.get_prior()
    flag = ow.marginalize_points(sel)
    assert np.array_equal(owt.marginalize_points(sel), flag)
    sysm.marginalize_points(sel)
    assert flag.sum() > 10
    HMo, bMo = ow.get_prior()
    HMt, bMt = owt.get_prior()
    HMg, bMg = sysm.get_prior()
    assert np.abs(HMo - HM0).max() > 0
    _yardstick(HMg, HMo, HMt, "HM after marginalizePointsF")
    _yardstick(bMg, bMo, bMt, "bM after marginalizePointsF")
    st = sysm.stats()
    assert st["resInM"] > 0
    # the marginalised / dropped points left the window
    ids2 = sysm.point_ids()
    assert not np.isin(sel, ids2).any() and np.isin(rest, ids2).all()
    # second half of the frame's points is dropped, then the frame is marginalised
    ow.drop_points(rest)
    owt.drop_points(rest)
    sysm.drop_points(np.concatenate([rest, empty0[np.isin(empty0, ids2)]]))
    HM1, bM1 = ow.marginalize_frame_prior(0)
    HM1t, bM1t = owt.marginalize_frame_prior(0)
    sysm.marginalize_frame(0)
    n2, _, _ = sysm.counts()
    assert n2 == win.n - 1
    HMg, bMg = sysm.get_prior()
    assert HMg.shape == HM1.shape
    _yardstick(HMg, HM1, HM1t, "HM after marginalizeFrame")
    _yardstick(bMg, bM1, bM1t, "bM after marginalizeFrame")
    # the reduced window still optimises (indices re-packed, prior dimension consistent)
    rmse, its = sysm.optimize(2)
    assert np.isfinite(rmse) and rmse > 0
    sysm.close()
    ow.close()
    owt.close()
