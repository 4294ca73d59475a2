#!/usr/bin/env python
"""Create / use / destroy cycle of every handle type (System with a window and optimize, tracker with set_ref + one track, context images,
immature set) 150 times; prints host RSS and free device memory at four points -- they must not move (GPU box)."""
import os, sys, resource, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from sos_slam_amd import host, lib, synth
from sos_slam_amd.records import TraceParams
hip = C.CDLL("libamdhip64.so")
def devfree():
    f, t = C.c_size_t(0), C.c_size_t(0); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 2**20
def rss(): return int(open("/proc/self/statm").read().split()[1]) * 4096 / 2**20
win = synth.make_window("W7", extra_frames=1)
prm = TraceParams.default()
for i in range(151):
    sysm = host.System.from_window(win)
    sysm.optimize(2)
    ht = host.HostTracker(sysm); ht.set_ref()
    slot = sysm.upload_image(win.extra_images[0])
    T0 = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
    ht.track(slot, 1.0, T0, np.zeros(2), sysm.context().levels - 1)
    ctx = sysm.context()
    st = lib.ImmatureSet(ctx)
    u = np.arange(20, 220, dtype=np.int32); v = np.full(200, 100, np.int32)
    st.put(0, ctx.immature_init(prm, 0, u, v)); st.close()
    ht.close(); sysm.close()
    if i in (10, 50, 100, 150):
        print(f"iteration {i}: host RSS {rss():.1f} MiB, device free {devfree():.1f} MiB", flush=True)
