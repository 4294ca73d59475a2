import sys, time, ctypes as C, numpy as np
sys.path.insert(0, '.')
from sos_slam_amd import synth, host
win = synth.make_window(sys.argv[1] if len(sys.argv) > 1 else "W12")
s = host.System.from_window(win); s.prepare()
L = host.load(); ph = (C.c_double*8)()
for i in range(5): s.gn_iteration(i)
L.sosf_get_timing(ph, 1)
N=50; t=time.perf_counter()
for i in range(N): s.gn_iteration(i)
dt=(time.perf_counter()-t)/N
L.sosf_get_timing(ph, 1)
names = ["accumulate+stitch","assemble+solve","resubstitute","step+precalc","pushState","linearize","applyRes"]
print("iter %.1f us" % (dt*1e6))
for k,nm in enumerate(names): print("  %-18s %.1f us" % (nm, ph[k]/N*1e6))
