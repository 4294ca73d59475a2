#!/usr/bin/env python
"""Developer tool: error of the solved increment against the fp64-accumulating oracle for the stored-tile loop, the
pipelined (fused, MFMA block sums) loop and the op-ordered fp32 oracle, over several windows and seeds."""
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
for name in ("T4", "T6", "W7"):
    for seed in (1, 2, 3, 4):
        win = synth.make_window(name, seed=seed)
        o_ref, o_tru = prepared(win, False), prepared(win, True)
        plain, piped = host.System.from_window(win), host.System.from_window(win)
        plain.prepare(); piped.prepare(); piped.set_pipeline(True)
        for it in range(3):
            for o in (o_ref, o_tru, plain, piped):
                o.gn_iteration(it)
            t = o_tru.lastX()
            e = [np.abs(x.lastX() - t).max() for x in (o_ref, plain, piped)]
            rows.append((name, seed, it, *e))
            print(name, seed, it, "ref %.2e plain %.2e piped %.2e  piped/max(ref,plain) %.2f" % (*e, e[2] / max(e[0], e[1], 1e-30)))
        plain.close(); piped.close(); o_ref.close(); o_tru.close()
r = np.array([x[3:] for x in rows if x[2] > 0])
print("geometric mean ratio piped/ref %.2f  plain/ref %.2f" % (np.exp(np.mean(np.log(r[:, 2] / r[:, 0]))), np.exp(np.mean(np.log(r[:, 1] / r[:, 0])))))
