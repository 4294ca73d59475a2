#!/usr/bin/env python
"""Run-to-run determinism of the device path (GPU box): the free-running device chain of tests/rolling.py twice (visual with frames between
keyframes, and visual-inertial) and a W12 optimize() three times -- every logged quantity must come out bit-identical.  The kernels sum in fixed
trees and exchange through tagged granules; a missing barrier or an order that depends on timing would show up here as a difference."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from sos_slam_amd import host, synth  # noqa: E402
from tests import rolling  # noqa: E402


def digest_chain(**kw):
    sc = rolling.Scenario(**kw)
    c = rolling.DeviceChain(sc)
    c.bootstrap()
    h = hashlib.sha256()
    while True:
        lg = c.step()
        if lg is None:
            break
        for a in (lg.tracked_pose, lg.tracked_aff, lg.HM, lg.bM):
            h.update(np.ascontiguousarray(a).tobytes())
        for fid in lg.window_ids:
            h.update(np.ascontiguousarray(lg.window_poses[fid]).tobytes())
        h.update(repr((lg.flagged, lg.activated, sorted(lg.residual_set), sorted(lg.point_set_after), lg.iterations, lg.marg_points, lg.dropped_points)).encode())
        if lg.vio:
            h.update(repr((lg.vio["scale"], lg.vio["trapped"])).encode())
            for fid in sorted(lg.vio["states"]):
                h.update(np.ascontiguousarray(lg.vio["states"][fid]).tobytes())
    c.close()
    return h.hexdigest()


def digest_optimize():
    win = synth.make_window("W12")
    sysm = host.System.from_window(win)
    sysm.optimize(6)
    h = hashlib.sha256()
    for f in range(win.n):
        h.update(np.ascontiguousarray(sysm.frame(f)["camToWorld"]).tobytes())
    h.update(np.ascontiguousarray(sysm.points()["idepth"]).tobytes())
    sysm.close()
    return h.hexdigest()


bad = 0
for name, fn in (("visual, keyframe every 3rd frame", lambda: digest_chain(n_frames=4 + 3 * 8, kf_every=3, step=0.07 / 3, rot=0.008 / 3)),
                 ("visual-inertial", lambda: digest_chain(vio=True, n_frames=14)),
                 ("optimize W12", digest_optimize)):
    d = [fn() for _ in range(3)]
    same = len(set(d)) == 1
    bad += not same
    print(f"{name}: {'identical' if same else 'DIFFERENT'} over {len(d)} runs ({d[0][:16]}...)", flush=True)
sys.exit(1 if bad else 0)
