import ctypes as C, sys, numpy as np
sys.path.insert(0, '/root/repo')
from sos_slam_amd import host, lib, synth
for name in ("W12", "W16"):
    win = synth.make_window(name)
    sysm = host.System.from_window(win)
    sysm.prepare()
    for i in range(2):
        sysm.gn_iteration(i)
    L = lib.load()
    ba = C.c_void_p(host.load().sosf_ba(sysm.h_))
    th = np.array([sysm.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
    ms = C.c_float(0)
    for k in ("lin_floor", "linearize_fused", "linearize", "calib_read", "calib_write", "sc_gram_prep", "stitch", "reduce", "resub_fused"):
        rc = L.sos_ba_time_kernel(ba, k.encode(), th.ctypes.data_as(C.c_void_p), 200, C.byref(ms))
        print(name, k, rc, round(ms.value * 1e3, 2), "us")
    sysm.close()
