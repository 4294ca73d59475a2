#!/usr/bin/env python
"""Workload for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): the W12 window, a few launches of the
calibration kernels (known streamed bytes) and of the hot-path kernels, each name repeated so that the per-kernel
averages of the counter database are launch averages.  Prints the byte counts the calibration kernels move.

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o fetch -- python tools/pmc_probe.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o write -- python tools/pmc_probe.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sos_slam_amd import host, lib, synth  # noqa: E402
import ctypes as C  # noqa: E402

win = synth.make_window(sys.argv[1] if len(sys.argv) > 1 else "W12")
sysm = host.System.from_window(win)
sysm.prepare()
for i in range(3):
    sysm.gn_iteration(i)
L = lib.load()
ba = C.c_void_p(host.load().sosf_ba(sysm.h_))
th = np.array([sysm.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
ms = C.c_float(0)
for name in ("calib_read", "calib_write", "linearize_fused", "linearize_apply", "top_accumulate", "sc_gram_prep", "reduce", "stitch"):
    L.sos_ba_time_kernel(ba, name.encode(), th.ctypes.data_as(C.c_void_p), 20, C.byref(ms))
print(json.dumps({"window": win.name, "residuals": int(win.R), "points": int(win.P)}))
sysm.close()
