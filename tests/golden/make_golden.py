"""Generates tests/golden/t3_golden.npz: inputs and oracle outputs of the small T3 window.

The reference ships no golden vectors for this path, so this fixture pins the ORACLE (the C
restatement in oracle/) against accidental change and gives the GPU tests a reference that travels with the
repository.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from sos_slam_amd import synth  # noqa: E402


def main():
    win = synth.make_window("T3")
    ow = orc.window_from_synth(win)
    th = np.full(win.n, 512.0, np.float32)
    ow.reset_oob()
    E = ow.linearize(th)
    out = dict(
        images=win.images, frames=win.frames, points=win.points, resid=win.resid, K=win.K, HM=win.HM, bM=win.bM,
        dI0=ow.dI[0][0], dI0_l1=ow.dI[0][1],
        precalc=ow.precalc().copy(), adHTdeltaF=ow.adHTdeltaF().copy(), adHost=ow.adHost().copy(),
        adTarget=ow.adTarget().copy(),
        lin_energy=np.float64(E), new_state=ow.new_state().copy(), new_energy=ow.new_energy().copy(),
        new_energy_wo=ow.new_energy_wo().copy(), center=ow.center().copy(), Jnew=ow.Jnew().copy())
    ow.apply_res()
    out["JpJdF"] = ow.JpJdF().copy()
    a32 = ow.accumulate(fp64_truth=False)
    a64 = ow.accumulate(fp64_truth=True)
    for k in ("H_A", "b_A", "H_sc", "b_sc"):
        out["acc32_" + k] = a32[k]
        out["acc64_" + k] = a64[k]
    out["resInA"] = np.int32(a32["resInA"])
    out["idepth_hessian"] = ow.point_field("idepth_hessian").copy()
    x = np.linspace(-1e-3, 1e-3, 4 + 8 * win.n)
    out["resub_x"] = x
    out["resub_step"] = ow.resubstitute(x).copy()
    ow2 = orc.window_from_synth(win)
    rmse, its = ow2.optimize(6)
    out["opt_rmse"] = np.float32(rmse)
    out["opt_iters"] = np.int32(its)
    out["opt_camToWorld"] = np.stack([ow2.frame(f)["camToWorld"] for f in range(win.n)])
    out["opt_state"] = np.stack([ow2.frame(f)["state"] for f in range(win.n)])
    out["opt_frameEnergyTH"] = np.array([ow2.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
    out["opt_idepth"] = ow2.pts()["idepth_scaled"].copy()
    out["opt_res_flags"] = ow2.res()["flags"].copy()
    out["opt_res_state"] = ow2.res()["state_state"].copy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "t3_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
