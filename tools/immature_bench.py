#!/usr/bin/env python
"""Timing of FullSystem::traceNewCoarse's inner loops (ImmaturePoint::traceOn over the immature points of every
keyframe against a new frame) on the device against the oracle port on one host core (the reference traces
single-threaded, FS/FullSystem.cpp:311-350)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from sos_slam_amd import lib, synth  # noqa: E402
from sos_slam_amd.records import TraceParams  # noqa: E402
from tests import immature_helpers as ih  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "W7"
per_host = int(sys.argv[2]) if len(sys.argv) > 2 else 1500   # setting_desiredImmatureDensity
win = synth.make_window(name, extra_frames=1)
prm = TraceParams.default()
ctx = lib.Context(win.w, win.h)
for i in range(win.n):
    ctx.make_pyramid(i, win.images[i])
ctx.make_pyramid(win.n, win.extra_images[0])
new_dI, _ = orc.make_images(win.extra_images[0])
hosts = []
for h in range(win.n):
    u, v, _ = ih.candidates(win, h, per_host, seed=h)
    pts = ctx.immature_init(prm, h, u, v)
    # give the points a finite interval the way earlier traces would have: one trace against a neighbour keyframe
    nb = h + 1 if h + 1 < win.n else h - 1
    KRKi, Kt, aff = ih.host_to_frame(win.K, win.frames[h]["camToWorld"], win.frames[nb]["camToWorld"])
    pts = ctx.immature_trace(prm, nb, pts, KRKi, Kt, aff)
    hosts.append((pts, ih.host_to_frame(win.K, win.frames[h]["camToWorld"], win.extra_poses[0])))


all_pts = np.concatenate([p for p, _ in hosts])
host_of = np.concatenate([np.full(len(p), k, np.int32) for k, (p, _) in enumerate(hosts)])
KR = np.stack([kk[0].reshape(-1) for _, kk in hosts])
KT = np.stack([kk[1] for _, kk in hosts])
AF = np.stack([kk[2] for _, kk in hosts])


def run_gpu():
    out = ctx.immature_trace_all(prm, win.n, all_pts, host_of, KR, KT, AF)
    o, r = 0, []
    for p, _ in hosts:
        r.append(out[o:o + len(p)])
        o += len(p)
    return r


def run_cpu():
    return [orc.immature_trace(prm, new_dI[0], pts, *kk) for pts, kk in hosts]


g = run_gpu()
t0 = time.perf_counter()
for _ in range(10):
    run_gpu()
t_gpu = (time.perf_counter() - t0) / 10
# the same trace on device-resident records (sos_immset): what a frame between two keyframes costs.  Every timed trace starts from
# the same records (put + synchronise outside the timed region), like the array call above
iset = lib.ImmatureSet(ctx)
t_res, reps = 0.0, 10
for _ in range(reps + 1):
    for k, (p, _) in enumerate(hosts):
        iset.put(k, p)
    ctx.synchronize()
    t0 = time.perf_counter()
    iset.trace(prm, win.n, np.arange(len(hosts)), KR, KT, AF)
    ctx.synchronize()
    if _:
        t_res += time.perf_counter() - t0
t_res /= reps
res_same = all(np.array_equal(iset.get(k).tobytes(), g[k].tobytes()) for k in range(len(hosts)))
iset.close()
c = run_cpu()
t0 = time.perf_counter()
for _ in range(3):
    run_cpu()
t_cpu = (time.perf_counter() - t0) / 3
same = all(np.array_equal(a["lastTraceStatus"], b["lastTraceStatus"]) and np.array_equal(a["idepth_min"], b["idepth_min"], equal_nan=True)
           for a, b in zip(g, c))
st = np.concatenate([a["lastTraceStatus"] for a in g])

# activation (FullSystem::optimizeImmaturePoint over the candidates activatePointsMT picks: ~2000 a keyframe at start-up,
# a few hundred in steady state); the reference spreads them over its worker threads, the port runs on one
from sos_slam_amd.records import ActivateParams, Calib  # noqa: E402
aprm, calib = ActivateParams.default(), Calib.from_K(win.K)
pairs = ih.pair_tfms(win)
traced = np.concatenate(g)
cand = np.flatnonzero(np.isfinite(traced["idepth_max"]) & (traced["lastTraceStatus"] != 2))[:2000]
cpts, chost = traced[cand], host_of[cand]
dI0 = [orc.make_images(win.images[f])[0][0] for f in range(win.n)]
slots = np.arange(win.n)
a_g = ctx.immature_activate(aprm, calib, slots, pairs, cpts, chost)
t0 = time.perf_counter()
for _ in range(20):
    ctx.immature_activate(aprm, calib, slots, pairs, cpts, chost)
t_act_gpu = (time.perf_counter() - t0) / 20
a_c = orc.immature_activate(aprm, calib, dI0, pairs, cpts, chost)
t0 = time.perf_counter()
for _ in range(3):
    orc.immature_activate(aprm, calib, dI0, pairs, cpts, chost)
t_act_cpu = (time.perf_counter() - t0) / 3
act = {"candidates": int(len(cand)), "gpu_ms": t_act_gpu * 1e3, "cpu_port_ms_1thread": t_act_cpu * 1e3,
       "identical_to_oracle": bool(all(np.array_equal(a_g[f], a_c[f], equal_nan=True) for f in a_g.dtype.names if f != "pad")),
       "status_histogram_skip_delete_activated": [int((a_g["status"] == k).sum()) for k in (0, -1, 1)]}
# pixel selection in front of the constructor (PixelSelector::makeMaps, once per keyframe)
from sos_slam_amd.records import PixselParams, random_pattern  # noqa: E402
pat = random_pattern(win.w * win.h)
pprm = PixselParams.default()
sel_g, sel_o = lib.PixelSelector(ctx, pprm, pat), orc.PixelSelector(pprm, pat, win.w, win.h)
dIk, absgk = orc.make_images(win.images[0])
sel_o.make_hists(absgk[0])
sel_g.make_hists(0)
m_g, n_g = sel_g.make_maps(0, 1500.0)
m_o, n_o = sel_o.make_maps(dIk, absgk, 1500.0)
t0 = time.perf_counter()
for _ in range(20):
    sel_g.current_potential = 3
    sel_g.make_maps(0, 1500.0, want_map=False)
t_sel_gpu = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(5):
    sel_o.current_potential = 3
    sel_o.make_maps(dIk, absgk, 1500.0)
t_sel_cpu = (time.perf_counter() - t0) / 5
pixsel = {"make_maps_gpu_ms": t_sel_gpu * 1e3, "make_maps_cpu_port_ms": t_sel_cpu * 1e3, "selected": int(n_g),
          "identical_to_oracle": bool(n_g == n_o and np.array_equal(m_g, m_o))}
sel_g.close()
print(json.dumps({"window": name, "keyframes": win.n, "immature_points": int(len(st)), "gpu_ms": t_gpu * 1e3, "gpu_resident_ms": t_res * 1e3, "resident_identical_to_array_call": bool(res_same), "cpu_port_ms_1thread": t_cpu * 1e3,
                  "points_per_s_gpu": len(st) / t_gpu, "identical_to_oracle": bool(same),
                  "status_histogram": np.bincount(st, minlength=6).tolist(), "activation": act, "pixel_selection": pixsel}))
ctx.close()
