#!/usr/bin/env python
"""Developer tool: phase timestamps of k_linearize (build with SOS_HIPCC_EXTRA=-DSOS_LIN_PROFILE)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sos_slam_amd import host, lib, synth  # noqa: E402

win = synth.make_window(sys.argv[1] if len(sys.argv) > 1 else "W12")
sysm = host.System.from_window(win)
sysm.prepare()
for i in range(3):
    sysm.gn_iteration(i)
L = lib.load()
ba = C.c_void_p(host.load().sosf_ba(sysm.h_))
th = np.array([sysm.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
ms = C.c_float(0)
for name in (sys.argv[2:] or ["linearize_fused"]):
    L.sos_ba_time_kernel(ba, name.encode(), th.ctypes.data_as(C.c_void_p), 1, C.byref(ms))
    nb = min(8192, (win.R + 31) // 32 + 200)
    out = np.zeros((nb, 8), dtype=np.uint64)
    L.sos_debug_lin_prof(out.ctypes.data_as(C.c_void_p), nb)
    out = out[out[:, 0] > 0].astype(np.int64)
    t0 = out[:, 0].min()
    ncol = 8
    rel = (out[:, :ncol] - t0) * 1.0  # s_memtime ticks (shader clock cycles); clocks of different XCDs are not aligned
    print(name, "blocks", len(out), "avg launch us", round(ms.value * 1e3, 2))
    names = ["start", "L1 loads", "taps", "phase1 end", "barrier1", "sums+barrier2", "phase2+barrier3", "end"]
    for k, nm in enumerate(names):
        col = rel[:, k]
        print(f"  {nm:10s} min {col.min():7.2f} p50 {np.median(col):7.2f} p90 {np.percentile(col, 90):7.2f} max {col.max():7.2f}")
    d = np.diff(rel, axis=1)
    print("  per-phase median cycles:", [int(np.median(d[:, k])) for k in range(ncol - 1)], "block total", int(np.median(rel[:, ncol - 1] - rel[:, 0])))
sysm.close()
