import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from sos_slam_amd import synth, host, lib
win = synth.make_window(sys.argv[1] if len(sys.argv) > 1 else "W12")
s = host.System.from_window(win); s.prepare()
for i in range(3): s.gn_iteration(i)
L = lib.load(); ba = C.c_void_p(host.load().sosf_ba(s.h_))
th = np.array([s.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
ms = C.c_float(0)
for name in ["linearize", "linearize_apply", "top_accumulate", "sc_accumulate", "reduce", "stitch", "resubstitute"]:
    best = 1e9
    for rep in range(3):
        L.sos_ba_time_kernel(ba, name.encode(), th.ctypes.data_as(C.c_void_p), 300, C.byref(ms)); best = min(best, ms.value)
    extra = ""
    if name.startswith("linearize"): extra = "  %.0f GB/s algorithmic = %.1f%% of 8 TB/s" % (win.R*776/best/1e6, win.R*776/best/1e6/80)
    print("%-18s %7.2f us%s" % (name, best*1e3, extra))
