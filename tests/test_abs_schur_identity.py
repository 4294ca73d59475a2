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


def _tile_decode(ut, T):
    mt, rem = 0, ut
    while rem >= T - mt:
        rem -= T - mt
        mt += 1
    return mt, rem


@pytest.mark.parametrize("n,nchunks", [(3, 5), (6, 19), (12, 70), (17, 9)])
def test_index_arithmetic_of_the_gram_store_and_the_chunk_sum(n, nchunks):
    """k_sc_gram_abs stores only the 16 x 16 tiles on and above the diagonal of each chunk's Gram matrix (lane (kq, col) of the wave
    owning tile `ut` writes rows m0 + 4 kq + 0..3, column n0 + col); k_abs_reduce_stitch1 sums them over the chunks (block = one row of
    one tile, thread = (column, one of eight chunk classes)) and scatters Gram order [frames | calib | b] into the system order
    [calib | frames].  The two index schemes transliterated from the kernels, run on random symmetric per-chunk matrices: every
    entry of the upper triangle of H_sc and of b_sc must be written, from positions the Gram kernel has written, with the right sum."""
    rng = np.random.default_rng(n * 100 + nchunks)
    cols = 8 * n + 5
    Dm = (cols + 15) // 16 * 16
    T = Dm // 16
    full = rng.normal(size=(nchunks, Dm, Dm))
    full = full + full.transpose(0, 2, 1)
    full[:, cols:, :] = 0
    full[:, :, cols:] = 0                      # the padding columns of A are zero in the kernel
    gram_part = np.full((nchunks, Dm, Dm), np.nan)
    for blk in range(nchunks):                 # ---- the store of k_sc_gram_abs
        for wave in range(4):
            for ut in range(wave, T * (T + 1) // 2, 4):
                mt, rem = _tile_decode(ut, T)
                m0, n0 = mt << 4, (mt + rem) << 4
                for lane in range(64):
                    kq, col = lane >> 4, lane & 15
                    for rgi in range(4):
                        gram_part[blk, m0 + kq * 4 + rgi, n0 + col] = full[blk, m0 + kq * 4 + rgi, n0 + col]
    dim = 4 + 8 * n
    Hs, bs = np.full((dim, dim), np.nan), np.full(dim, np.nan)
    writes = np.zeros((dim, dim), int)
    for b in range(T * (T + 1) // 2 * 16):     # ---- the H_sc | b_sc half of k_abs_reduce_stitch1
        ut, rr = b >> 4, b & 15
        mt, rem = _tile_decode(ut, T)
        sPart = np.zeros((8, 16))
        for tid in range(128):
            r, c, sub = (mt << 4) + rr, ((mt + rem) << 4) + (tid & 15), tid >> 4
            sv = 0.0
            if r < cols - 1 and c < cols and c >= r:
                k0 = sub
                while k0 < nchunks:
                    for u in range(8):
                        if k0 + 8 * u < nchunks:
                            sv += gram_part[k0 + 8 * u, r, c]
                    k0 += 64
            sPart[sub][tid & 15] = sv
        for tid in range(16):
            r, c = (mt << 4) + rr, ((mt + rem) << 4) + tid
            if r < cols - 1 and c < cols and c >= r:
                tot = sum(sPart[u][tid] for u in range(8))
                R = 4 + r if r < 8 * n else r - 8 * n
                if c == cols - 1:
                    assert np.isnan(bs[R])
                    bs[R] = tot
                else:
                    Cc = 4 + c if c < 8 * n else c - 8 * n
                    i, j = (R, Cc) if R <= Cc else (Cc, R)
                    Hs[i, j] = tot
                    writes[i, j] += 1
    G = full.sum(axis=0)
    order = np.array([8 * n + k for k in range(4)] + list(range(8 * n)))
    H_ref, b_ref = G[np.ix_(order, order)], G[order, 8 * n + 4]
    iu = np.triu_indices(dim)
    assert not np.isnan(Hs[iu]).any() and not np.isnan(bs).any()
    assert (writes[iu] == 1).all() and writes[np.tril_indices(dim, -1)].sum() == 0      # each entry once, nothing below the diagonal
    assert np.allclose(Hs[iu], H_ref[iu], rtol=1e-12, atol=1e-12) and np.allclose(bs, b_ref, rtol=1e-12, atol=1e-12)
