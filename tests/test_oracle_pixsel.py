"""CPU known-answer tests of the PixelSelector restatement (oracle/orc_pixsel.c)."""
import numpy as np

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import PixselParams, random_pattern


def _frame(name="T6", k=0):
    win = synth.make_window(name)
    dI, absg = orc.make_images(win.images[k])
    return win, dI, absg


def test_histogram_thresholds_and_selection_structure():
    win, dI, absg = _frame()
    w, h = win.w, win.h
    ps = orc.PixelSelector(PixselParams.default(), random_pattern(w * h), w, h)
    ths, sm = ps.make_hists(absg[0])
    # independent numpy restatement of the per-cell median-of-gradient-magnitude threshold
    w32, h32 = w // 32, h // 32
    for (x, y) in ((0, 0), (w32 - 1, h32 - 1), (w32 // 2, h32 // 2)):
        blk = absg[0][32 * y:32 * y + 32, 32 * x:32 * x + 32]
        jj, ii = np.mgrid[32 * y:32 * y + 32, 32 * x:32 * x + 32]
        ok = ~((ii > w - 2) | (jj > h - 2) | (ii < 1) | (jj < 1))
        g = np.minimum(np.sqrt(blk[ok]).astype(np.int32), 48)
        hist = np.bincount(g, minlength=49)
        th = int(np.float32(ok.sum()) * np.float32(0.5) + np.float32(0.5))
        q = 90
        for i in range(49):
            th -= hist[i]
            if th < 0:
                q = i
                break
        assert ths[x + y * w32] == q + 7.0
    assert np.all(sm > 0) and sm.shape == ths.shape
    for pot in (1, 3, 5):
        m, n = ps.select(dI, absg, pot)
        assert set(np.unique(m)).issubset({0.0, 1.0, 2.0, 4.0})
        assert (m == 1).sum() == n[0] and (m == 2).sum() == n[1] and (m == 4).sum() == n[2]
        ys, xs = np.nonzero(m)
        assert xs.min() >= 4 and xs.max() < w - 5 and ys.min() >= 4 and ys.max() <= h - 4
        # at most one pixel per pot x pot cell of the (4 pot)-aligned grid
        cells = (ys // pot) * 100000 + xs // pot if (4 * pot) % pot == 0 else None
        assert len(np.unique(cells)) == len(cells)
        # level-0 picks exceed their cell threshold
        y0, x0 = np.nonzero(m == 1)
        thp = sm[np.minimum((x0 >> 5) + (y0 >> 5) * w32, len(sm) - 1)]
        inside = (x0 >> 5) + (y0 >> 5) * w32 < len(sm)
        assert np.all(absg[0][y0, x0][inside] > thp[inside])
    n1 = ps.select(dI, absg, 1)[1].sum()
    n5 = ps.select(dI, absg, 5)[1].sum()
    assert n1 > 4 * n5     # ~ K / (pot + 1)^2


def test_make_maps_density_control():
    win, dI, absg = _frame()
    w, h = win.w, win.h
    ps = orc.PixelSelector(PixselParams.default(), random_pattern(w * h), w, h)
    ps.make_hists(absg[0])
    want = 600.0
    counts = []
    for _ in range(3):   # the potential adapts from call to call
        m, num = ps.make_maps(dI, absg, want)
        assert num == int((m != 0).sum())
        counts.append(num)
    assert 0.6 * want <= counts[-1] <= 1.3 * want, (counts, ps.current_potential)
    assert ps.current_potential >= 1
