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
    if os.environ.get("SOS_LIN_PROFILE_HWID"):  # built with -DSOS_LIN_PROFILE_HWID: column 1 = HW_ID | XCC_ID << 32
        o = out[out[:, 0] > 0]
        hw = o[:, 1]
        cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(int)
        sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(int)
        se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
        xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
        key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        cnt = np.bincount(np.unique(key, return_inverse=True)[1])
        nd = int(os.environ.get("SOS_LIN_ND_INFO", "0"))
        print(name, "blocks", len(o), "distinct CUs", len(cnt), "blocks per CU histogram", np.bincount(cnt).tolist())
        if nd:  # tiles per CU when the first nd blocks carry two tiles
            tiles = np.where(np.arange(len(o)) < nd, 2, 1)
            tcu = np.bincount(np.unique(key, return_inverse=True)[1], weights=tiles)
            print("  tiles per CU histogram", np.bincount(tcu.astype(int)).tolist())
        print("  xcc of the first 24 blocks", xcc[:24].tolist())
        print("  (se, sh, cu) of the first 12 blocks of xcc", xcc[0], [(int(a), int(b), int(c)) for a, b, c in zip(se[xcc == xcc[0]][:12], sh[xcc == xcc[0]][:12], cu[xcc == xcc[0]][:12])])
        dur = (o[:, 7] - o[:, 0]).astype(float) / 100
        for c in sorted(set(cnt.tolist())):
            sel = cnt[np.unique(key, return_inverse=True)[1]] == c
            print("  blocks on CUs with", c, "blocks: n", int(sel.sum()), "median duration us %.2f max %.2f" % (np.median(dur[sel]), dur[sel].max()))
        continue
    out = out[out[:, 0] > 0].astype(np.int64)
    t0 = out[:, 0].min()
    ncol = 8
    rel = (out[:, :ncol] - t0) * 1.0  # s_memtime ticks (shader clock cycles); clocks of different XCDs are not aligned
    print(name, "blocks", len(out), "avg launch us", round(ms.value * 1e3, 2))
    names = ["start", "L1 loads", "taps", "phase1 end", "barrier1", "sums+barrier2", "phase2+barrier3", "end"]
    for k, nm in enumerate(names):
        col = rel[:, k]
        print(f"  {nm:10s} min {col.min():7.2f} p50 {np.median(col):7.2f} p90 {np.percentile(col, 90):7.2f} max {col.max():7.2f}")
    if os.environ.get("SOS_LIN_PROFILE_WALL"):  # ticks of 10 ns on a clock shared by the XCDs
        dur = rel[:, ncol - 1] - rel[:, 0]
        order = np.argsort(rel[:, 0])
        print("  wall clock (us): start spread", rel[:, 0].max() / 100, "end spread", (rel[:, 7].max() - rel[:, 7].min()) / 100,
              "kernel span", rel[:, 7].max() / 100)
        print("  block duration us: min %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(dur, [0, 50, 90, 100]) / 100))
        print("  start of the last-finishing 10 blocks (us):", sorted((rel[np.argsort(rel[:, 7])[-10:], 0] / 100).tolist()))
        print("  duration of the last-finishing 10 blocks (us):", (dur[np.argsort(rel[:, 7])[-10:]] / 100).tolist())
        print("  start time percentiles us:", (np.percentile(rel[:, 0], [10, 50, 90, 99]) / 100).tolist())
        print("  end time percentiles us:", (np.percentile(rel[:, 7], [10, 50, 90, 99]) / 100).tolist())
        bid = np.flatnonzero(np.ones(len(rel)))
        print("  corr(block index, start):", float(np.corrcoef(bid, rel[:, 0])[0, 1]))
    d = np.diff(rel, axis=1)
    print("  per-phase median cycles:", [int(np.median(d[:, k])) for k in range(ncol - 1)], "block total", int(np.median(rel[:, ncol - 1] - rel[:, 0])))
sysm.close()
