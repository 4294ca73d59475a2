"""The identity behind the absolute-coordinate Schur path of the device (csrc/sos_ba.hip, k_sc_gram_abs): the reference accumulates
accD / accE / accEB per (host, target1, target2) in relative coordinates and stitches them with the adjoints
(OB/AccumulatedSCHessian.cpp:63-77, 105-139); applying the (linear) stitch to the per-residual rows first,

    w_p = [ sum_t adHost[h,t] JpJd_t  (block of the host) ;  adTarget[h,t] JpJd_t  (block of target t) ;  Hcd ;  bdSum ],

gives  H_sc | b_sc = sum_p Hdi_p w_p w_p^T  directly.  Checked here on the CPU with the oracle's own per-residual / per-point
quantities and adjoints against its stitched H_sc / b_sc -- the index conventions ([host + n target] adjoint blocks, row i of the
adjoint times JpJd, Gram order [frames | calib | b] -> system order [calib | frames]) are the ones the kernels use."""
import numpy as np
import pytest

from sos_slam_amd import synth
from tests import helpers as hp


@pytest.mark.parametrize("name", ["T3", "T4", "T6"])
def test_gram_of_stitched_rows_is_the_stitched_schur_complement(name):
    win = synth.make_window(name)
    ow = hp.oracle_window(win)
    ow.reset_oob()
    th = np.array([ow.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
    ow.linearize(th)
    ow.apply_res()
    acc = ow.accumulate(fp64_truth=True)
    n, P = win.n, win.P
    JpJd = np.array(ow.JpJdF(), dtype=np.float64)
    res = ow.res()
    active = (res["flags"] & synth.RF_ACTIVE) != 0
    adH, adT = np.array(ow.adHost()).reshape(n * n, 8, 8), np.array(ow.adTarget()).reshape(n * n, 8, 8)
    adH, adT = adH.astype(np.float32).astype(np.float64), adT.astype(np.float32).astype(np.float64)   # the fp32 copies (adHostF / adTargetF)
    hdi, bds = np.array(ow.point_field("HdiF"), np.float64), np.array(ow.point_field("bdSumF"), np.float64)
    hcd = np.array(ow.point_field("Hcd_accAF"), np.float64) + np.array(ow.point_field("Hcd_accLF"), np.float64)
    cols = 8 * n + 5
    W = np.zeros((P, cols))
    for r in np.flatnonzero(active):
        p, h, t = int(res["point"][r]), int(res["host"][r]), int(res["target"][r])
        W[p, 8 * t:8 * t + 8] += adT[h + n * t] @ JpJd[r]
        W[p, 8 * h:8 * h + 8] += adH[h + n * t] @ JpJd[r]
    W[:, 8 * n:8 * n + 4] = hcd
    W[:, 8 * n + 4] = bds
    G = (W * hdi[:, None]).T @ W
    dim = 4 + 8 * n
    order = np.array([8 * n + k for k in range(4)] + list(range(8 * n)))      # system order [calib | frames] in Gram columns
    H_abs, b_abs = G[np.ix_(order, order)], G[order, 8 * n + 4]
    H_ref, b_ref = acc["H_sc"], acc["b_sc"]
    assert np.abs(H_ref).max() > 0
    assert np.abs(H_abs - H_ref).max() <= 2e-5 * np.abs(H_ref).max(), np.abs(H_abs - H_ref).max() / np.abs(H_ref).max()
    assert np.abs(b_abs - b_ref).max() <= 2e-5 * np.abs(b_ref).max()
    ow.close()
