#!/usr/bin/env python
"""Developer tool: error of the FIRST solve of optimize() against the fp64-accumulating oracle, over windows and seeds.
The top-Hessian block sums of that solve come from the matrix cores (tile sums formed inside prepare()'s linearisation);
SOS_NO_PREPARE_PREFETCH=1 selects the stored-Jacobian accumulate instead, for comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sos_slam_amd import host, synth  # noqa: E402
from tests import helpers as hp  # noqa: E402


def prepared(win, truth):
    ow = hp.oracle_window(win)
    ow.set_truth_mode(truth)
    ow.reset_oob()
    th = np.array([ow.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
    ow.linearize(th)
    ow.apply_res()
    return ow


rows = []
CASES = [("T4", {}), ("T6", {}), ("W7", {})] + [("T6", dict(n=n, P=100 * n)) for n in (5, 8, 9, 10, 11)]
for name, over in CASES:
    for seed in range(1, 9 if not over else 4):
        win = synth.make_window(name, seed=seed, **over)
        label = name if not over else "T6n%d" % over["n"]
        o_ref, o_tru = prepared(win, False), prepared(win, True)
        dev = host.System.from_window(win)
        dev.prepare()
        for o in (o_ref, o_tru, dev):
            o.gn_iteration(0)
        t = o_tru.lastX()
        e = [np.abs(x.lastX() - t).max() for x in (o_ref, dev)]
        rows.append(e)
        print(label, seed, "ref %.2e dev %.2e  dev/ref %.2f" % (*e, e[1] / max(e[0], 1e-30)))
        dev.close(); o_ref.close(); o_tru.close()
r = np.array(rows)
print("variant", "stored-J" if os.environ.get("SOS_NO_PREPARE_PREFETCH") else "tile sums", "geometric mean dev/ref %.2f  max %.2f" % (np.exp(np.mean(np.log(r[:, 1] / r[:, 0]))), (r[:, 1] / r[:, 0]).max()))
