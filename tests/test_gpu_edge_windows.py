"""Window shapes at the edges of what the facade's loops are built for, through optimize() against the oracle: two and three keyframes
(the 20- / 15-iteration rules of FS/FullSystemOptimize.cpp:311-312), 17 keyframes (the last size the device-side step takes), 18 and 20
(host-side step, no device-resident loop), one and three points, 40 points over nine keyframes, and each of them again with a host
keyframe that owns no point at all.  Same iteration counts; poses within 4x the oracle's own fp32-vs-fp64-accumulation distance."""
import dataclasses

import numpy as np
import pytest

from sos_slam_amd import synth
from tests import helpers as hp

pytestmark = pytest.mark.gpu

CASES = [("T4", dict(n=2)), ("T4", dict(n=3)), ("T6", dict(n=17)), ("T6", dict(n=18)), ("T6", dict(n=20, P=1500)), ("T4", dict(P=1)),
         ("T4", dict(P=3)), ("T6", dict(n=9, P=40))]


def _rmse(a, b, n):
    return float(np.sqrt(np.mean(np.square(np.concatenate([a(f)["camToWorld"] - b(f)["camToWorld"] for f in range(n)])))))


def _without_host(win, h0):
    keepp = win.points["host"] != h0
    idx = np.cumsum(keepp) - 1
    keepr = keepp[win.resid["point"]]
    r2 = win.resid[keepr].copy()
    r2["point"] = idx[r2["point"]]
    return dataclasses.replace(win, points=win.points[keepp].copy(), resid=r2)


@pytest.mark.parametrize("name,ov", CASES, ids=[f"{n}-{'-'.join(f'{k}{v}' for k, v in o.items())}" for n, o in CASES])
def test_edge_window_optimize_matches_oracle(name, ov):
    from sos_slam_amd import host
    win = synth.make_window(name, **ov)
    variants = [win] + ([_without_host(win, int(win.points["host"][0]))] if win.P > 10 else [])
    for w in variants:
        ow, ot = hp.oracle_window(w), hp.oracle_window(w)
        ot.set_truth_mode(True)
        rm_o, it_o = ow.optimize(6)
        ot.optimize(6)
        sysm = host.System.from_window(w)
        rm_g, it_g = sysm.optimize(6)
        e, nz = _rmse(sysm.frame, ow.frame, w.n), _rmse(ow.frame, ot.frame, w.n)
        print(f"{name} {ov}: n={w.n} P={w.P} R={w.R}: iterations {it_g}/{it_o}, rmse {rm_g:.5f}/{rm_o:.5f}, pose device-oracle {e:.2e} oracle-truth {nz:.2e}")
        assert it_g == it_o
        assert abs(rm_g - rm_o) <= 2e-3 * rm_o
        assert e <= max(1e-5, 4 * nz), (e, nz)
        sysm.close(); ow.close(); ot.close()
