import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from sos_slam_amd import synth, host, lib
win = synth.make_window(sys.argv[1] if len(sys.argv) > 1 else "W12")
s = host.System.from_window(win); s.prepare()
L = lib.load(); ba = C.c_void_p(host.load().sosf_ba(s.h_))
th = np.array([s.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
ms = C.c_float(0)
L.sos_ba_time_kernel(ba, b"linearize", th.ctypes.data_as(C.c_void_p), 20, C.byref(ms))
print("R", win.R, "linearize us", ms.value*1e3)
