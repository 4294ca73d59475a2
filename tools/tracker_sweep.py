#!/usr/bin/env python
"""Seed sweep of the tracker parity checks (GPU box): tests/test_gpu_tracker_lm.py pins two templates (W7, T6); this builds the same
rig on windows drawn from other seeds and measures, per (seed, initial-pose perturbation): the device loop against the host loop
around the device passes (same evaluation counts, poses), and against the oracle port with its fp64-accumulation variant as the
yardstick (|T_dev - T_orc|, |T_dev - T_truth|, |T_orc - T_truth|).

    python tools/tracker_sweep.py [--seeds 8] [--window T6] [--out gpurun_out/tracker_sweep.json]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from sos_slam_amd import synth  # noqa: E402
from sos_slam_amd.synth import se3_exp12 as se3_exp, se3_mul12 as se3_mul  # noqa: E402
from tests import test_gpu_tracker_lm as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--window", default="T6")
    ap.add_argument("--out", default="gpurun_out/tracker_sweep.json")
    a = ap.parse_args()
    rows = []
    for s in range(a.seeds):
        seed = synth.SEED + 500 + 29 * s
        gen = T.make_rig(a.window, seed=seed)
        rig = next(gen)
        ht, win, ot, ott, levels = rig["ht"], rig["win"], rig["ot"], rig["ott"], rig["levels"]
        for pi, pert in enumerate(T.PERTURB):
            Tinit = se3_mul(se3_exp(pert), T._rel_pose(win))
            ok_o, To, ao, lo, fo = ot.track(rig["new_dI"], 1.0, 1.0, rig["ref_aff"], Tinit, np.zeros(2), levels - 1)
            ok_t, Tt, at, lt, ft = ott.track(rig["new_dI"], 1.0, 1.0, rig["ref_aff"], Tinit, np.zeros(2), levels - 1)
            ht.set_device_lm(True)
            ok_g, Tg, ag, lg, fg = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1)
            ev_d = ht.last_evals()
            ht.set_device_lm(False)
            ok_h, Th, ah, lh, fh = ht.track(rig["new_slot"], 1.0, Tinit, np.zeros(2), levels - 1)
            ev_h = ht.last_evals()
            ht.set_device_lm(True)
            rows.append(dict(seed=seed, perturbation=pi, ok=[bool(ok_g), bool(ok_h), bool(ok_o), bool(ok_t)], evals_device=int(ev_d), evals_host_loop=int(ev_h),
                             dev_hostloop=float(np.abs(Tg - Th).max()), dev_orc=float(np.abs(Tg - To).max()), dev_truth=float(np.abs(Tg - Tt).max()),
                             orc_truth=float(np.abs(To - Tt).max()), aff_dev_orc=float(np.abs(ag - ao).max()),
                             res_rel=float(np.nanmax(np.abs(lg[:levels] - lo[:levels]) / np.maximum(np.abs(lo[:levels]), 1e-12)))))
            print(rows[-1], flush=True)
        gen.close()

    def geo(k):
        return float(np.exp(np.mean(np.log(np.maximum([r[k] for r in rows], 1e-16)))))

    summ = dict(window=a.window, cases=len(rows), evals_equal=int(sum(r["evals_device"] == r["evals_host_loop"] for r in rows)),
                ok_equal=int(sum(len(set(r["ok"])) == 1 for r in rows)),
                max_dev_hostloop=max(r["dev_hostloop"] for r in rows),
                geo=dict(dev_orc=geo("dev_orc"), dev_truth=geo("dev_truth"), orc_truth=geo("orc_truth")),
                max=dict(dev_orc=max(r["dev_orc"] for r in rows), dev_truth=max(r["dev_truth"] for r in rows), orc_truth=max(r["orc_truth"] for r in rows)),
                max_res_rel=max(r["res_rel"] for r in rows))
    print(json.dumps(summ, indent=1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(summary=summ, rows=rows), f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
