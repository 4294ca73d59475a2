#!/usr/bin/env python
"""Wall time of FullSystem::optimize (FS/FullSystemOptimize.cpp:305-489) on a freshly packed window: pack + first
linearisation + the Gauss-Newton iterations + the final linearizeAll(true), i.e. what one keyframe pays for the
backend (FS/FullSystem.cpp:853), as opposed to bench.py's steady-state iteration.

  python tools/keyframe_cost.py [W12] [reps]          SOS_TIMING=1 prints the phases of every call to stderr
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sos_slam_amd import host, synth  # noqa: E402


def measure(window="W12", reps=5, iters=6, min_iters=1):
    """optimize() on ONE system whose pack is invalidated before every call, as in a running system where every keyframe changes
    the graph and the handles keep their buffers.  The first call (allocations) is not counted; every call continues from the
    state the previous one converged to, so the number of iterations is reported with the time."""
    win = synth.make_window(window)
    sysm = host.System.from_window(win)
    sysm.set_min_opt_iterations(min_iters)
    out, its = [], []
    for _ in range(reps + 1):
        sysm.invalidate_pack()
        t0 = time.perf_counter()
        _, it = sysm.optimize(iters)
        out.append((time.perf_counter() - t0) * 1e3)
        its.append(it)
    first = out[0]
    out = out[1:]
    sysm.close()
    return {"window": window, "optimize_ms": float(np.median(out)), "optimize_ms_min": float(min(out)), "optimize_ms_first_call": float(first),
            "iterations": [int(i) for i in its], "residuals": int(win.R)}


if __name__ == "__main__":
    w = sys.argv[1] if len(sys.argv) > 1 else "W12"
    r = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    print(json.dumps({"converging": measure(w, r), "six_iterations": measure(w, r, min_iters=6)}))
