"""GPU parity of the PixelSelector kernels (makeHists / select / makeMaps / the makeNewTraces list) against the oracle:
thresholds, selection maps and counts bit-identical, also on an image whose gradients are exactly orthogonal to one of
the sixteen directions (every cell's selection then depends on the running count, the serial branch of the scan)."""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import IMMATURE_DTYPE, PixselParams, TraceParams, random_pattern

pytestmark = pytest.mark.gpu


def _check_frame(ctx, sel_g, sel_o, slot, img, pots=(1, 2, 3, 5), density=1500.0):
    ctx.make_pyramid(slot, img)
    dI, absg = orc.make_images(img)
    ths_g, sm_g = sel_g.make_hists(slot)
    ths_o, sm_o = sel_o.make_hists(absg[0])
    assert np.array_equal(ths_g, ths_o) and np.array_equal(sm_g, sm_o)
    for pot in pots:
        for thf in (1.0, 2.0):
            m_g, n_g = sel_g.select(slot, pot, thf)
            m_o, n_o = sel_o.select(dI, absg, pot, thf)
            assert np.array_equal(n_g, n_o), (pot, thf, n_g, n_o)
            assert np.array_equal(m_g, m_o), (pot, thf, int((m_g != m_o).sum()))
    return dI, absg


def test_selection_bit_exact_and_new_traces():
    from sos_slam_amd import lib
    win = synth.make_window("W7")
    w, h = win.w, win.h
    pattern = random_pattern(w * h)
    ctx = lib.Context(w, h)
    prm = PixselParams.default()
    sel_g, sel_o = lib.PixelSelector(ctx, prm, pattern), orc.PixelSelector(prm, pattern, w, h)
    for k in range(3):   # successive keyframes: currentPotential carries over
        dI, absg = _check_frame(ctx, sel_g, sel_o, k, win.images[k], pots=(1, 3) if k else (1, 2, 3, 5))
        for want in (1500.0, 400.0):
            m_g, num_g = sel_g.make_maps(k, want)
            m_o, num_o = sel_o.make_maps(dI, absg, want)
            assert num_g == num_o and sel_g.current_potential == sel_o.current_potential, (k, want, num_g, num_o)
            assert np.array_equal(m_g, m_o)
        # FullSystem::makeNewTraces: the selected pixels inside the pattern padding, row-major, into the constructor
        u, v, t = sel_g.list(pattern_padding=2)
        ys, xs = np.nonzero(m_o)
        keep = (xs >= 3) & (xs < w - 4) & (ys >= 3) & (ys < h - 4)
        assert np.array_equal(u, xs[keep]) and np.array_equal(v, ys[keep]) and np.array_equal(t, m_o[ys[keep], xs[keep]])
        tp = TraceParams.default()
        p_g = ctx.immature_init(tp, k, u, v)
        p_o = orc.immature_init(tp, dI[0], u, v)
        for f in IMMATURE_DTYPE.names:
            assert np.array_equal(p_g[f], p_o[f], equal_nan=True), f
    # without the direction distribution
    prm2 = PixselParams.default(selectDirectionDistribution=0, minGradHistAdd=4.0)
    s2g, s2o = lib.PixelSelector(ctx, prm2, pattern), orc.PixelSelector(prm2, pattern, w, h)
    _check_frame(ctx, s2g, s2o, 0, win.images[0], pots=(2, 4))
    for s in (sel_g, s2g):
        s.close()
    ctx.close()


def test_direction_dependent_cells_take_the_serial_scan():
    from sos_slam_amd import lib
    w, h = 320, 240
    rng = np.random.default_rng(2)
    # columns only: dy == 0 exactly, so direction 0 = (0, 1) sees no gradient anywhere and whether a cell selects
    # depends on the running count n2
    cols = np.cumsum(rng.normal(0, 12, w)).astype(np.float32) + 120
    img = np.repeat(cols[None, :], h, axis=0).astype(np.float32)
    pattern = random_pattern(w * h)
    ctx = lib.Context(w, h)
    prm = PixselParams.default()
    sel_g, sel_o = lib.PixelSelector(ctx, prm, pattern), orc.PixelSelector(prm, pattern, w, h)
    dI, absg = _check_frame(ctx, sel_g, sel_o, 0, img, pots=(1, 2, 3))
    assert np.all(dI[0][..., 2] == 0)
    # mixed image: a few dependent cells among independent ones
    img2 = img.copy()
    img2[60:180, 80:240] += rng.normal(0, 15, (120, 160)).astype(np.float32)
    _check_frame(ctx, sel_g, sel_o, 1, img2, pots=(1, 3))
    m_g, num_g = sel_g.make_maps(1, 800.0)
    m_o, num_o = sel_o.make_maps(*orc.make_images(img2), 800.0)
    assert num_g == num_o and np.array_equal(m_g, m_o)
    sel_g.close()
    ctx.close()
