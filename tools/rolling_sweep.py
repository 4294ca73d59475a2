#!/usr/bin/env python
"""Seed sweep of the free-running rolling-window parity chains (GPU box): device chain vs oracle chain vs the oracle chain with
fp64-accumulated H/b over many seeds, per-seed worst distances and the ensemble summary of tests/rolling_ensemble.py.

    python tools/rolling_sweep.py [--seeds 12] [--frames 20] [--kf-every 3] [--out gpurun_out/rolling_sweep.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from sos_slam_amd import synth  # noqa: E402
from tests import rolling_ensemble as re_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--out", default="gpurun_out/rolling_sweep.json")
    ap.add_argument("--kf-every", type=int, default=1, help="> 1: only every k-th frame is a keyframe (visual only; --frames counts keyframes then)")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    res = {}
    rc = 0
    for vio in ((False,) if a.kf_every > 1 else (False, True)):
        runs = []
        kw = dict(n_frames=a.frames) if a.kf_every == 1 else dict(n_frames=4 + a.kf_every * a.frames, kf_every=a.kf_every, step=0.07 / a.kf_every, rot=0.008 / a.kf_every)
        for s in range(a.seeds):
            t0 = time.time()
            r = re_.run_seed(synth.SEED + 1000 + 37 * s, vio=vio, **kw)
            r["seconds"] = round(time.time() - t0, 1)
            runs.append(r)
            print({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
        summ = re_.summarize(runs)
        res["visual_inertial" if vio else "visual"] = dict(summary=summ, runs=runs)
        print(json.dumps(summ, indent=1, default=str), flush=True)
        rc |= 1 if summ["violations"] else 0
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1, default=str)
    return rc


if __name__ == "__main__":
    sys.exit(main())
