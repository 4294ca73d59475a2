"""Seed / noise sweep of the backend parity checks (GPU box): the tests pin a handful of windows, this runs the same
comparisons over many randomly drawn ones and writes a summary.

    python tools/parity_sweep.py [--seeds 12] [--out gpurun_out/parity_sweep.json]

Per window: state sets, energies, JpJdF bit-exact after linearize + applyRes (device vs oracle); then optimize(4) on
both: same iteration count, pose RMSE, equal active index sets.  The oracle is the checker here, nothing else.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from sos_slam_amd import host, synth  # noqa: E402
from tests import helpers as hp  # noqa: E402


def one(name, seed, noise_sigma, state_noise):
    win = synth.make_window(name, seed=seed, noise_sigma=noise_sigma, state_noise=state_noise)
    ow = hp.oracle_window(win)
    ctx, ba = hp.gpu_backend(win, ow)
    th = np.array([ow.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
    ow.reset_oob()
    ba.reset_oob()
    ow.linearize(th, nthreads=6)
    g = ba.linearize(th)
    r = dict(window=name, seed=seed, noise_sigma=noise_sigma, state_noise=state_noise, residuals=int(win.R))
    r["state_sets_equal"] = bool(np.array_equal(g["newState"].astype(np.int32), ow.new_state()))
    r["energies_equal"] = bool(np.array_equal(g["newEnergy"], ow.new_energy()) and
                               np.array_equal(g["newEnergyWithOutlier"], ow.new_energy_wo()))
    r["counts_in_oob_outlier"] = [int(x) for x in np.bincount(ow.new_state(), minlength=3)]
    ow.apply_res()
    ba.apply_res()
    act = (ow.res()["flags"] & 1) != 0
    r["JpJdF_equal"] = bool(np.array_equal(ba.JpJdF()[act], ow.JpJdF()[act]))
    ba.close()
    ctx.close()
    ow.close()
    # the whole loop
    ow = hp.oracle_window(win)
    ot = hp.oracle_window(win)
    ot.set_truth_mode(True)
    rm_o, it_o = ow.optimize(4)
    ot.optimize(4)
    sysm = host.System.from_window(win)
    rm_g, it_g = sysm.optimize(4)

    def rmse(a, b):
        e = [a(f)["camToWorld"] - b(f)["camToWorld"] for f in range(win.n)]
        return float(np.sqrt(np.mean(np.square(np.concatenate(e)))))

    r["iterations"] = [int(it_g), int(it_o)]
    r["rmse_vs_oracle"] = rmse(sysm.frame, ow.frame)
    r["oracle_fp32_noise"] = rmse(ow.frame, ot.frame)
    r["rmse_vs_fp64_accumulation"] = rmse(sysm.frame, ot.frame)
    ro = ow.res()
    alive = (ro["flags"] & 0x100) == 0
    rg = sysm.residuals()
    r["active_sets_equal"] = bool(len(rg["state_state"]) == int(alive.sum()) and
                                  np.array_equal(np.sort(rg["state_state"]), np.sort(ro["state_state"][alive])))
    r["rms_energy_rel_diff"] = float(abs(rm_g - rm_o) / max(abs(rm_o), 1e-30))
    sysm.close()
    ow.close()
    ot.close()
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--windows", default="T4,T6,W7")
    ap.add_argument("--out", default="gpurun_out/parity_sweep.json")
    a = ap.parse_args()
    rows = []
    t0 = time.time()
    for name in a.windows.split(","):
        for k in range(a.seeds):
            seed = synth.SEED + 7919 * (k + 1)
            noise = (1.0, 2.0, 4.0)[k % 3]
            sn = (3e-4, 1e-3)[k % 2]
            rows.append(one(name, seed, noise, sn))
            print(json.dumps(rows[-1]), flush=True)
    bit = [r for r in rows if not (r["state_sets_equal"] and r["energies_equal"] and r["JpJdF_equal"])]
    act = [r for r in rows if not r["active_sets_equal"]]
    summary = dict(cases=len(rows), bit_exact_failures=len(bit), active_set_failures=len(act),
                   iteration_count_mismatches=sum(r["iterations"][0] != r["iterations"][1] for r in rows),
                   max_rmse_vs_oracle=max(r["rmse_vs_oracle"] for r in rows),
                   max_rmse_over_noise=max(r["rmse_vs_oracle"] / max(r["oracle_fp32_noise"], 1e-12) for r in rows),
                   worst_rmse_vs_fp64=max(r["rmse_vs_fp64_accumulation"] for r in rows),
                   worst_oracle_noise=max(r["oracle_fp32_noise"] for r in rows), seconds=round(time.time() - t0, 1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(summary=summary, rows=rows), f, indent=1)
    print("SUMMARY", json.dumps(summary))


if __name__ == "__main__":
    main()
