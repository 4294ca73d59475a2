#!/usr/bin/env python
"""Timing of the CoarseTracker / ScaleOptimizer path on the device against the oracle port on the host cores
(trackNewestCoarse and optimizeScale on a W12-sized keyframe window, 752x480, 5 pyramid levels)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from sos_slam_amd import host, synth  # noqa: E402
from sos_slam_amd.records import Calib  # noqa: E402
from sos_slam_amd.synth import se3_exp12 as se3_exp, se3_mul12 as se3_mul  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "W7"
win = synth.make_window(name, extra_frames=2)
sysm = host.System.from_window(win)
sysm.optimize(3)
ht = host.HostTracker(sysm)
pc_n = ht.set_ref()
new_slot = sysm.upload_image(win.extra_images[0])
st_slot = sysm.upload_image(win.extra_images[1])
levels = int(np.count_nonzero(pc_n)) if np.count_nonzero(pc_n) else len(pc_n)
ref = win.frames[win.n - 1]["camToWorld"]
new = win.extra_poses[0]
Rr, tr, Rn, tn = ref[:9].reshape(3, 3), ref[9:], new[:9].reshape(3, 3), new[9:]
T0 = np.concatenate([(Rn.T @ Rr).reshape(-1), Rn.T @ (tr - tn)])
Tinit = se3_mul(se3_exp(np.array([0.004, -0.003, 0.002, 0.002, -0.002, 0.001])), T0)


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


t_set = timeit(lambda: ht.set_ref(), 20)
t_trk = timeit(lambda: ht.track(new_slot, 1.0, Tinit, np.zeros(2), levels - 1), 50)   # device-resident LM loop (default)
evals = ht.last_evals()
lm_profile = ht.lm_profile(0)   # the kernel's own phase stamps of the last launch
ht.set_device_lm(False)
t_trk_host = timeit(lambda: ht.track(new_slot, 1.0, Tinit, np.zeros(2), levels - 1), 20)  # LM loop on the host, one round trip per evaluation
ht.set_device_lm(True)
# hypothesis loop of trackNewCoarse: 83 tries that all have to be looked at (lastCoarseRMSE tiny -> no early exit), batched vs one by one
kf, sl = win.frames[win.n - 1]["camToWorld"], win.extra_poses[0]
from sos_slam_amd.synth import se3_inv12 as se3_inv  # noqa: E402
tries = ht.make_tries(se3_mul(se3_inv(kf), sl), se3_mul(se3_inv(sl), kf))
tiny = np.full(5, 1e-6)
t_hyp16 = timeit(lambda: ht.track_hypotheses(st_slot, 1.0, tries, np.zeros(2), levels - 1, tiny, batch=16), 5)
t_hyp1 = timeit(lambda: ht.track_hypotheses(st_slot, 1.0, tries, np.zeros(2), levels - 1, tiny, batch=1), 5)
ht.set_device_lm(False)
t_hyp_host = timeit(lambda: ht.track_hypotheses(st_slot, 1.0, tries, np.zeros(2), levels - 1, tiny, batch=1), 3)
ht.set_device_lm(True)
K1 = np.array(sysm.calib_value_scaled(), np.float32)
t_scl = timeit(lambda: ht.optimize_scale(st_slot, win.stereo_tfm, K1, 1.2, levels - 1), 20)
# FullSystem::optimizeScale before the scale is trapped: seven guesses (one launch on the device, one by one with the host loop)
t_sc7 = timeit(lambda: ht.optimize_scale_kf(st_slot, win.stereo_tfm, K1, 1.0, levels - 1, 12.0, [0, 0]), 20)
ht.set_device_lm(False)
t_sc7_host = timeit(lambda: ht.optimize_scale_kf(st_slot, win.stereo_tfm, K1, 1.0, levels - 1, 12.0, [0, 0]), 5)
ht.set_device_lm(True)
out = {"window": name, "template_pixels_per_level": [int(x) for x in pc_n[:levels]],
       "gpu_ms": {"set_ref": t_set * 1e3, "track": t_trk * 1e3, "track_host_loop": t_trk_host * 1e3, "optimize_scale": t_scl * 1e3,
                  "track_83_hypotheses_batch16": t_hyp16 * 1e3, "track_83_hypotheses_batch1": t_hyp1 * 1e3,
                  "track_83_hypotheses_host_loop": t_hyp_host * 1e3,
                  "optimize_scale_7_guesses": t_sc7 * 1e3, "optimize_scale_7_guesses_host_loop": t_sc7_host * 1e3},
       "residual_evaluations_per_track": evals, "us_per_evaluation": t_trk * 1e6 / max(evals, 1), "track_lm_profile": lm_profile}

# oracle port on the host (single thread, as the reference's tracker is)
ow = orc.window_from_synth(win)
ow.optimize(3)
res = ow.res()
sel = (res["target"] == win.n - 1) & ((res["flags"] & 0x101) == 1) & (res["state_state"] == 0)
c = ow.center()[sel]
hdi = ow.point_field("HdiF")[res["point"][sel]]
calib = Calib.from_K(ow.calib_value_scaled())
ot = orc.OracleTracker(win.params, win.w, win.h)
new_dI, _ = orc.make_images(win.extra_images[0])
st_dI, _ = orc.make_images(win.extra_images[1])
ref_aff = np.array([ow.frame(win.n - 1)["state"][6] * 10.0, ow.frame(win.n - 1)["state"][7] * 1000.0])
c_set = timeit(lambda: ot.set_ref(calib, ow.dI[win.n - 1], c[:, 0], c[:, 1], c[:, 2], hdi), 5)
c_trk = timeit(lambda: ot.track(new_dI, 1.0, 1.0, ref_aff, Tinit, np.zeros(2), levels - 1), 5)
c_scl = timeit(lambda: ot.optimize_scale(st_dI, win.stereo_tfm, K1, 1.2, levels - 1), 5)
out["cpu_port_ms"] = {"set_ref": c_set * 1e3, "track": c_trk * 1e3, "optimize_scale": c_scl * 1e3}
print(json.dumps(out))
