import time, numpy as np, sys
sys.path.insert(0, '.')
from sos_slam_amd import synth, lib
from tests import helpers as hp
t=time.time(); win = synth.make_window("W12"); print("gen W12", time.time()-t, win.P, win.R, flush=True)
t=time.time(); ow = hp.oracle_window(win); print("oracle win", time.time()-t, flush=True)
ctx, ba = hp.gpu_backend(win, ow)
th = np.full(win.n, 512.0, np.float32)
ow.reset_oob(); ba.reset_oob()
t=time.time(); E_o = ow.linearize(th); t_lin=time.time()-t
g = ba.linearize(th)
print("oracle linearize 1 thread: %.2f ms" % (t_lin*1e3), "states", np.bincount(ow.new_state(), minlength=3))
print("state equal", np.array_equal(g["newState"].astype(np.int32), ow.new_state()), "energy equal", np.array_equal(g["newEnergy"], ow.new_energy()))
ow.apply_res(); ba.apply_res()
t=time.time(); a_t = ow.accumulate(fp64_truth=True); print("oracle accumulate(fp64) %.1f ms" % ((time.time()-t)*1e3))
t=time.time(); a_r = ow.accumulate(fp64_truth=False); print("oracle accumulate(ref fp32) %.1f ms" % ((time.time()-t)*1e3))
t=time.time(); a_6 = ow.accumulate(fp64_truth=False, nthreads=6); print("oracle accumulate(ref fp32, 6 thr) %.1f ms" % ((time.time()-t)*1e3))
a_g = ba.accumulate()
for k in ("H_A","b_A","H_sc","b_sc"): print(k, "gpu-vs-truth", hp.relerr(a_g[k], a_t[k]), "ref-vs-truth", hp.relerr(a_r[k], a_t[k]))
for name in ["linearize","apply_res","top_accumulate","sc_accumulate","reduce","accumulate_local"]:
    ms = ba.time_kernel(name, th, 200)
    print("%-18s %.2f us/launch" % (name, ms*1e3))
R = win.R
ms = ba.time_kernel("linearize", th, 500)
print("linearize: %.2f us, %.1f GB/s algorithmic (776 B/res), %.2f Gres/s" % (ms*1e3, R*776/ms/1e6, R/ms/1e6))
t=time.time()
for i in range(20): a_g = ba.accumulate()
print("accumulate+stitch+D2H: %.1f us" % ((time.time()-t)/20*1e6))
