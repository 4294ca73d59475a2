"""CPU study (no GPU): how far from the fp64 truth does the ABSOLUTE-coordinate Schur path of the device land, compared with the
relative-coordinate arithmetic of the reference (the oracle's fp32 accumulators)?  The device tests judge the Gauss-Newton step
against the oracle's with the oracle's own fp32-vs-fp64 distance as the yardstick, so the quantity of interest is

    |x_abs32 - x_truth|  /  |x_oracle32 - x_truth|        (x = solution of (H_A - H_sc + damping) x = b_A - b_sc)

The absolute path is emulated in NumPy with the device's roundings: rows w_p formed in fp32 from the fp32 copies of the adjoints
(8-term sums, left to right), Gram products of 32-point chunks accumulated in fp32, chunk sums in fp64.

    python tools/abs_path_accuracy.py [T6 W7 ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sos_slam_amd import synth  # noqa: E402
from tests import helpers as hp  # noqa: E402


def study(name):
    win = synth.make_window(name)
    ow = hp.oracle_window(win)
    ow.reset_oob()
    th = np.array([ow.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
    ow.linearize(th)
    ow.apply_res()
    truth = ow.accumulate(fp64_truth=True)
    orc32 = ow.accumulate(fp64_truth=False)
    n, P = win.n, win.P
    JpJd = np.array(ow.JpJdF(), dtype=np.float32)
    res = ow.res()
    active = (res["flags"] & synth.RF_ACTIVE) != 0
    adH = np.array(ow.adHost()).reshape(n * n, 8, 8).astype(np.float32)
    adT = np.array(ow.adTarget()).reshape(n * n, 8, 8).astype(np.float32)
    hdi, bds = np.array(ow.point_field("HdiF"), np.float32), np.array(ow.point_field("bdSumF"), np.float32)
    hcd = (np.array(ow.point_field("Hcd_accAF"), np.float32) + np.array(ow.point_field("Hcd_accLF"), np.float32)).astype(np.float32)
    cols = 8 * n + 5
    W = np.zeros((P, cols), np.float32)
    WH = {}      # (point, target) -> adHost row, summed into the host block in target order like the kernel
    host_of = np.full(P, -1)
    for r in np.flatnonzero(active):
        p, h, t = int(res["point"][r]), int(res["host"][r]), int(res["target"][r])
        host_of[p] = h
        vt, vh = np.zeros(8, np.float32), np.zeros(8, np.float32)
        for j in range(8):     # left-to-right fp32 sums, as the kernel's unrolled loops
            vt = (vt + adT[h + n * t][:, j] * JpJd[r][j]).astype(np.float32)
            vh = (vh + adH[h + n * t][:, j] * JpJd[r][j]).astype(np.float32)
        W[p, 8 * t:8 * t + 8] = vt
        WH[(p, t)] = vh
    for p in range(P):
        if host_of[p] < 0:
            continue
        sv = np.zeros(8, np.float32)
        for t in range(n):
            if (p, t) in WH:
                sv = (sv + WH[(p, t)]).astype(np.float32)
        W[p, 8 * host_of[p]:8 * host_of[p] + 8] = sv
    W[:, 8 * n:8 * n + 4] = hcd
    W[:, 8 * n + 4] = bds
    G = np.zeros((cols, cols))
    order_pts = np.argsort(host_of, kind="stable")      # chunks hold points of one host
    for c0 in range(0, P, 32):
        idx = order_pts[c0:c0 + 32]
        A = W[idx]
        G += ((A * hdi[idx][:, None]).astype(np.float32).T @ A).astype(np.float64)      # fp32 products and sums inside a chunk
    order = np.array([8 * n + k for k in range(4)] + list(range(8 * n)))
    Hsc_abs, bsc_abs = G[np.ix_(order, order)], G[order, 8 * n + 4]

    def step(H_A, b_A, H_sc, b_sc):
        H = H_A - H_sc
        H = H + np.diag(np.diag(H_A) * 1e-4 + 1e-3)       # the loop's damping plus a little prior so that the gauge is fixed
        return np.linalg.solve(H, b_A - b_sc)

    x_t = step(truth["H_A"], truth["b_A"], truth["H_sc"], truth["b_sc"])
    x_o = step(orc32["H_A"], orc32["b_A"], orc32["H_sc"], orc32["b_sc"])
    x_a = step(orc32["H_A"], orc32["b_A"], Hsc_abs, bsc_abs)       # the top half is the same on both device paths
    sc = np.abs(truth["H_sc"]).max()
    out = dict(window=name, n=n, P=P, residuals=int(active.sum()),
               Hsc_err_oracle32=float(np.abs(orc32["H_sc"] - truth["H_sc"]).max() / sc),
               Hsc_err_abs32=float(np.abs(Hsc_abs - truth["H_sc"]).max() / sc),
               step_err_oracle32=float(np.abs(x_o - x_t).max()), step_err_abs32=float(np.abs(x_a - x_t).max()),
               step_norm=float(np.abs(x_t).max()))
    out["ratio"] = out["step_err_abs32"] / max(out["step_err_oracle32"], 1e-300)
    ow.close()
    return out


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["T6", "W7"]):
        print(study(nm))
