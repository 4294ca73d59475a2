"""oracle/mirror_np.py -- TEST INFRASTRUCTURE (only tests/ may import it).

An INDEPENDENT second reading of the reference's linearisation and Hessian accumulation, written from the reference
sources in vectorised NumPy **fp64** (SURVEY.md 8(c)(1)), so that a misreading shared by the C restatement
(oracle/orc_backend.c) and the HIP kernels -- both written by the same hand -- cannot hide:

  FrameFramePrecalc::set             FS/HessianBlocks.cpp:431-461, AffLight::fromToVecExposure util/NumType.h:156-168
  projectPoint (both overloads)      FS/ResidualProjections.h:43-76
  getInterpolatedElement33           util/globalFuncs.h:68-82
  PointFrameResidual::linearize      FS/Residuals.cpp:77-271   (pattern: util/settings.cpp:307-317, patternNum 8)
  EFResidual::takeDataF              OB/EnergyFunctionalStructs.cpp:36-45
  AccumulatedTopHessianSSE::addPoint<0> + stitchDoubleInternal   OB/AccumulatedTopHessian.cpp:35-147, 231-301
  AccumulatedSCHessianSSE::addPoint + stitchDoubleInternal       OB/AccumulatedSCHessian.cpp:32-79, 80-158

It does NOT follow the reference's accumulator structure.  The accumulation is restated as what it computes: every
active residual contributes 8 rows of a dense Jacobian over (calibration 4 | 8 per keyframe | one inverse depth per
point); H_top / b_top are the pose-calibration block of J^T J / J^T r, and H_sc / b_sc the Schur complement of the
inverse-depth block (with the point priors of the reference).  fp64 throughout: it is compared with the fp32 restatement
at fp32 tolerances, never bit for bit.
"""
from __future__ import annotations

import numpy as np

# staticPattern[8] (util/settings.cpp:307-317): the 8-pixel DSO pattern
PATTERN_P = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], dtype=np.float64)
SCALE_IDEPTH, SCALE_F, SCALE_C = 1.0, 50.0, 50.0   # FS/HessianBlocks.h:53-60
RES_IN, RES_OOB, RES_OUTLIER = 0, 1, 2


def _se3_inv(T):
    R, t = T[:9].reshape(3, 3), T[9:]
    return np.concatenate([R.T.reshape(-1), -R.T @ t])


def _se3_mul(A, B):
    Ra, ta, Rb, tb = A[:9].reshape(3, 3), A[9:], B[:9].reshape(3, 3), B[9:]
    return np.concatenate([(Ra @ Rb).reshape(-1), Ra @ tb + ta])


def precalc(evalPT, pre, K, ab_exposure, aff_g2l, aff_g2l_0_b):
    """FrameFramePrecalc::set for all ordered pairs.  evalPT / pre: (n, 12) camToWorld at the linearisation point / at the
    current state; K = fx, fy, cx, cy; aff_g2l (n, 2) current (a, b); aff_g2l_0_b (n,) = b at the linearisation point.
    Returns dict of (n*n, ...) arrays indexed [h + n*t]."""
    n = len(evalPT)
    fx, fy, cx, cy = [float(np.float32(v)) for v in K]
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    Ki = np.linalg.inv(Km)
    out = dict(KRKi=np.zeros((n * n, 3, 3)), Kt=np.zeros((n * n, 3)), R0=np.zeros((n * n, 3, 3)), t0=np.zeros((n * n, 3)),
               aff=np.zeros((n * n, 2)), b0=np.zeros(n * n))
    for h in range(n):
        for t in range(n):
            l0 = _se3_mul(_se3_inv(evalPT[t]), evalPT[h])
            l = _se3_mul(_se3_inv(pre[t]), pre[h])
            k = h + n * t
            out["R0"][k], out["t0"][k] = l0[:9].reshape(3, 3), l0[9:]
            out["KRKi"][k] = Km @ l[:9].reshape(3, 3) @ Ki
            out["Kt"][k] = Km @ l[9:]
            eF, eT = ab_exposure[h], ab_exposure[t]
            if eF == 0 or eT == 0:
                eF = eT = 1.0
            a = np.exp(aff_g2l[t, 0] - aff_g2l[h, 0]) * eT / eF
            out["aff"][k] = (a, aff_g2l[t, 1] - a * aff_g2l[h, 1])
            out["b0"][k] = aff_g2l_0_b[h]
    return out


def _interp33(dI, x, y):
    """getInterpolatedElement33 on an (h, w, 3) image at float positions (vectorised); positions must be inside."""
    ix, iy = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)   # (int)x for x > 0
    dx, dy = x - ix, y - iy
    dxdy = dx * dy
    return (dxdy[..., None] * dI[iy + 1, ix + 1] + (dy - dxdy)[..., None] * dI[iy + 1, ix] + (dx - dxdy)[..., None] * dI[iy, ix + 1]
            + (1 - dx - dy + dxdy)[..., None] * dI[iy, ix])


def linearize(points, resid, pc, K, images, frameEnergyTH, params):
    """PointFrameResidual::linearize over all residuals.  points / resid: the structured arrays of sos_slam_amd.synth;
    pc: dict from `precalc` (or float arrays of the same shapes); images: list of (h, w, 3) level-0 dI per frame.
    Returns new_state, energy (returned value), energy_wo, center (R, 3), J (dict), JpJdF (R, 8)."""
    R = len(resid)
    n = len(images)
    h_img, w_img = images[0].shape[:2]
    wM3G, hM3G = w_img - 3, h_img - 3
    fx, fy, cx, cy = [float(np.float32(v)) for v in K]
    fxi, fyi = float(np.float32(1.0) / np.float32(fx)), float(np.float32(1.0) / np.float32(fy))
    pt = points[resid["point"]]
    k = resid["host"] + n * resid["target"]
    u_pt, v_pt = pt["u"].astype(np.float64), pt["v"].astype(np.float64)
    idz, idp = pt["idepth_zero_scaled"].astype(np.float64), pt["idepth_scaled"].astype(np.float64)
    R0, t0, KRKi, Kt = [np.asarray(pc[q], dtype=np.float64)[k] for q in ("R0", "t0", "KRKi", "Kt")]
    aff, b0 = np.asarray(pc["aff"], dtype=np.float64)[k], np.asarray(pc["b0"], dtype=np.float64)[k]
    state = resid["state_state"].astype(np.int64)
    new_state = np.full(R, RES_IN, np.int64)
    oob = state == RES_OOB
    # ---- centre projection at the linearisation point (projectPoint with R, t, FS/ResidualProjections.h:53-76)
    KliP = np.stack([(u_pt - cx) * fxi, (v_pt - cy) * fyi, np.ones(R)], axis=1)
    ptp = np.einsum("rij,rj->ri", R0, KliP) + t0 * idz[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        drescale = 1.0 / ptp[:, 2]
    new_idepth = idz * drescale
    u, v = ptp[:, 0] * drescale, ptp[:, 1] * drescale
    Ku, Kv = u * fx + cx, v * fy + cy
    ok_c = (drescale > 0) & (Ku > 1.1) & (Kv > 1.1) & (Ku < wM3G) & (Kv < hM3G)
    oob |= ~ok_c
    center = np.stack([Ku, Kv, new_idepth], axis=1)
    # geometric Jacobians, FS/Residuals.cpp:125-170
    Jpdd = np.stack([drescale * (t0[:, 0] - t0[:, 2] * u) * SCALE_IDEPTH * fx, drescale * (t0[:, 1] - t0[:, 2] * v) * SCALE_IDEPTH * fy], axis=1)
    dCx = np.zeros((R, 4)); dCy = np.zeros((R, 4))
    dCx[:, 2] = drescale * (R0[:, 2, 0] * u - R0[:, 0, 0])
    dCx[:, 3] = fx * drescale * (R0[:, 2, 1] * u - R0[:, 0, 1]) * fyi
    dCx[:, 0] = KliP[:, 0] * dCx[:, 2]
    dCx[:, 1] = KliP[:, 1] * dCx[:, 3]
    dCy[:, 2] = fy * drescale * (R0[:, 2, 0] * v - R0[:, 1, 0]) * fxi
    dCy[:, 3] = drescale * (R0[:, 2, 1] * v - R0[:, 1, 1])
    dCy[:, 0] = KliP[:, 0] * dCy[:, 2]
    dCy[:, 1] = KliP[:, 1] * dCy[:, 3]
    dCx[:, 0] = (dCx[:, 0] + u) * SCALE_F
    dCx[:, 1] *= SCALE_F
    dCx[:, 2] = (dCx[:, 2] + 1) * SCALE_C
    dCx[:, 3] *= SCALE_C
    dCy[:, 0] *= SCALE_F
    dCy[:, 1] = (dCy[:, 1] + v) * SCALE_F
    dCy[:, 2] *= SCALE_C
    dCy[:, 3] = (dCy[:, 3] + 1) * SCALE_C
    z = np.zeros(R)
    dxi_x = np.stack([new_idepth * fx, z, -new_idepth * u * fx, -u * v * fx, (1 + u * u) * fx, -v * fx], axis=1)
    dxi_y = np.stack([z, new_idepth * fy, -new_idepth * v * fy, -(1 + v * v) * fy, u * v * fy, u * fy], axis=1)
    # ---- the 8 pattern pixels at the current state (projectPoint with KRKi, Kt)
    pu = u_pt[:, None] + PATTERN_P[None, :, 0]
    pv = v_pt[:, None] + PATTERN_P[None, :, 1]
    hom = np.stack([pu, pv, np.ones_like(pu)], axis=2)                       # (R, 8, 3)
    q = np.einsum("rij,rpj->rpi", KRKi, hom) + Kt[:, None, :] * idp[:, None, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        pKu, pKv = q[..., 0] / q[..., 2], q[..., 1] / q[..., 2]
    ok_p = (pKu > 1.1) & (pKv > 1.1) & (pKu < wM3G) & (pKv < hM3G)
    oob |= ~ok_p.all(axis=1)
    pKu_s, pKv_s = np.where(ok_p, pKu, 2.0), np.where(ok_p, pKv, 2.0)       # safe positions for the gather
    hit = np.zeros((R, 8, 3))
    for f in range(n):
        m = resid["target"] == f
        if m.any():
            hit[m] = _interp33(np.asarray(images[f], dtype=np.float64), pKu_s[m], pKv_s[m])
    oob |= ~np.isfinite(hit[..., 0]).all(axis=1)
    color, weights = pt["color"].astype(np.float64), pt["weights"].astype(np.float64)
    residual = hit[..., 0] - (aff[:, 0:1] * color + aff[:, 1:2])
    drdA = color - b0[:, None]
    cth = float(params["outlierTHSumComponent"])
    huber = float(params["huberTH"])
    w = np.sqrt(cth / (cth + hit[..., 1] ** 2 + hit[..., 2] ** 2))
    w = 0.5 * (w + weights)
    with np.errstate(divide="ignore", invalid="ignore"):
        hw = np.where(np.abs(residual) < huber, 1.0, huber / np.abs(residual))
    energy_wo = np.sum(w * w * hw * residual * residual * (2 - hw), axis=1)
    hw2 = np.where(hw < 1, np.sqrt(hw), hw) * w
    JIdx = np.stack([hit[..., 1] * hw2, hit[..., 2] * hw2], axis=1)          # (R, 2, 8)
    JabF = np.stack([drdA * hw2, hw2], axis=1)
    if params["affineOptModeA"] < 0:
        JabF[:, 0] = 0
    if params["affineOptModeB"] < 0:
        JabF[:, 1] = 0
    resF = residual * hw2
    # the 2 x 2 sums are formed from the UNMASKED photometric rows (the masking of JabF happens after the += lines)
    ab_rows = np.stack([drdA * hw2, hw2], axis=1)
    JIdx2 = np.einsum("rap,rbp->rab", JIdx, JIdx)
    JabJIdx = np.einsum("rap,rbp->rab", ab_rows, JIdx)
    Jab2 = np.einsum("rap,rbp->rab", ab_rows, ab_rows)
    wJI2_sum = np.sum(hw2 * hw2 * (hit[..., 1] ** 2 + hit[..., 2] ** 2), axis=1)
    th = np.asarray(frameEnergyTH, dtype=np.float64)
    thmax = np.maximum(th[resid["host"]], th[resid["target"]])
    outlier = (energy_wo > thmax) | (wJI2_sum < 2)
    energy = np.where(outlier, thmax, energy_wo)
    new_state[outlier] = RES_OUTLIER
    new_state[oob] = RES_OOB
    energy = np.where(oob, resid["state_energy"].astype(np.float64), energy)
    energy_wo = np.where(oob, -1.0, energy_wo)
    J = dict(resF=resF, Jpdxi=np.stack([dxi_x, dxi_y], axis=1), Jpdc=np.stack([dCx, dCy], axis=1), Jpdd=Jpdd, JIdx=JIdx, JabF=JabF,
             JIdx2=JIdx2, JabJIdx=JabJIdx, Jab2=Jab2)
    # EFResidual::takeDataF
    JI_JI_Jd = np.einsum("rab,rb->ra", JIdx2, Jpdd)
    JpJdF = np.zeros((R, 8))
    JpJdF[:, :6] = J["Jpdxi"][:, 0] * JI_JI_Jd[:, 0:1] + J["Jpdxi"][:, 1] * JI_JI_Jd[:, 1:2]
    JpJdF[:, 6:] = np.einsum("rab,rb->ra", JabJIdx, Jpdd)
    return dict(new_state=new_state, energy=energy, energy_wo=energy_wo, center=center, center_ok=ok_c & (state != RES_OOB), J=J, JpJdF=JpJdF)


def dense_system(J, active, resid, points, n, adHost, adTarget, cDeltaF):
    """The accumulated system of the ACTIVE (non-linearised) residuals, restated as dense normal equations.
    adHost / adTarget: (n*n, 8, 8) fp64 as EnergyFunctional::setAdjointsF leaves them.  Returns H_A, b_A (pose /
    calibration block of J^T J, J^T r), H_sc, b_sc (Schur complement of the inverse-depth block with the point priors,
    shiftPriorToZero = true as accumulateSCF_MT calls it) and the per-point idepth_hessian / HdiF / bdSumF."""
    act = np.flatnonzero(active)
    P = len(points)
    dim = 4 + 8 * n
    rows = 8 * len(act)
    A = np.zeros((rows, dim + P))
    rvec = np.zeros(rows)
    for j, r in enumerate(act):
        h, t, p = int(resid["host"][r]), int(resid["target"][r]), int(resid["point"][r])
        k = h + n * t
        JI = J["JIdx"][r]                                          # (2, 8): d residual_i / d (x, y)
        geo_c = JI.T @ J["Jpdc"][r]                                # (8, 4)
        geo_xi = JI.T @ J["Jpdxi"][r]                              # (8, 6)
        rel8 = np.concatenate([geo_xi, J["JabF"][r].T], axis=1)    # (8 pixels, 8 relative parameters)
        sl = slice(8 * j, 8 * j + 8)
        A[sl, 0:4] += geo_c
        A[sl, 4 + 8 * h:12 + 8 * h] += rel8 @ adHost[k].T
        A[sl, 4 + 8 * t:12 + 8 * t] += rel8 @ adTarget[k].T
        A[sl, dim + p] += JI.T @ J["Jpdd"][r]
        rvec[sl] = J["resF"][r]
    H = A.T @ A
    b = A.T @ rvec
    H_A, b_A = H[:dim, :dim].copy(), b[:dim].copy()
    Hpd = H[:dim, dim:]
    prior = points["priorF"].astype(np.float64)
    delta = points["deltaF"].astype(np.float64)
    has = np.zeros(P, bool)
    has[resid["point"][act]] = True
    Hdd = np.diag(H)[dim:] + prior
    Hdd = np.where(Hdd < 1e-10, 1e-10, Hdd)
    bd = b[dim:] + prior * delta
    Hdi = np.where(has, 1.0 / Hdd, 0.0)
    H_sc = (Hpd * Hdi[None, :]) @ Hpd.T
    b_sc = Hpd @ (Hdi * bd)
    (void := cDeltaF)   # the calibration delta only enters through the L (linearised) residuals and the prior terms
    return dict(H_A=H_A, b_A=b_A, H_sc=H_sc, b_sc=b_sc, idepth_hessian=np.where(has, Hdd, 0.0), HdiF=Hdi,
                bdSumF=np.where(has, bd, 0.0), Hpd=Hpd, has=has)


def resubstitute(dense, x):
    """EnergyFunctional::resubstituteF_MT / resubstituteFPt (OB/EnergyFunctional.cpp:496-551) restated on the dense normal equations in
    ABSOLUTE coordinates: the second block row  Hpd^T x + Hdd d = bd  gives d, and the point step is -d (the frame and calibration
    steps are -x).  The reference gets there through relative coordinates (xAd = x_h^T adHost + x_t^T adTarget per frame pair, then
    xAd . JpJdF per residual); no adjoint appears here.  `dense` = the result of dense_system."""
    d = dense["HdiF"] * (dense["bdSumF"] - dense["Hpd"].T @ np.asarray(x, dtype=np.float64))
    return np.where(dense["has"], -d, 0.0)


# ------------------------------------------------------------------------------------------------
# CoarseTracker::calcResPose + calcGSSSEPose (FS/CoarseTracker.cpp:612-764, 554-610) as ONE vectorised fp64 function: the energy /
# counts of a pose, and the 8 x 8 system of the accepted pixels.  No warp buffers, no SSE accumulator: every template pixel
# contributes w * j j^T with j = (8 Jacobian entries, residual), FS/CoarseTracker.cpp:571-591 + OB/MatrixAccumulators.h
# (Accumulator9::updateSSE_eighted).
# ------------------------------------------------------------------------------------------------
def tracker_res_gs(pc_u, pc_v, pc_idepth, pc_color, dI, K4, R, t, affLL, b0, huber, cutoff, lvl0):
    """pc_*: template pixels of the level; dI: (h, w, 3) new frame level; K4 = (fx, fy, cx, cy) of the level; (R, t) = refToNew;
    affLL = (a, b) of fromToVecExposure; b0 = lastRef_aff_g2l.b.  Returns dict(E, numTermsInE, numWarped, numSaturated, flowT,
    flowRT, H (8, 8), b (8)) -- H, b before the SCALE_* factors and with the reference's 1 / n (n = numWarped rounded up to 4)."""
    x, y, idp, col = (np.asarray(a, dtype=np.float64) for a in (pc_u, pc_v, pc_idepth, pc_color))
    fx, fy, cx, cy = (float(v) for v in K4)
    hl, wl = dI.shape[:2]
    Ki = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1]])
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = np.asarray(t, dtype=np.float64)
    ray = Ki @ np.stack([x, y, np.ones_like(x)])               # Ki * (x, y, 1)
    pt = R @ ray + t[:, None] * idp
    u, v = pt[0] / pt[2], pt[1] / pt[2]
    Ku, Kv = fx * u + cx, fy * v + cy
    new_id = idp / pt[2]
    out = {}
    if lvl0:       # flow indicators over every 32nd template pixel, :666-696
        s = np.arange(len(x)) % 32 == 0
        def proj(p):
            return fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy
        KuT, KvT = proj(ray[:, s] + t[:, None] * idp[s])
        KuT2, KvT2 = proj(ray[:, s] - t[:, None] * idp[s])
        Ku3, Kv3 = proj(R @ ray[:, s] - t[:, None] * idp[s])
        sT = np.sum((KuT - x[s]) ** 2 + (KvT - y[s]) ** 2 + (KuT2 - x[s]) ** 2 + (KvT2 - y[s]) ** 2)
        sRT = np.sum((Ku[s] - x[s]) ** 2 + (Kv[s] - y[s]) ** 2 + (Ku3 - x[s]) ** 2 + (Kv3 - y[s]) ** 2)
        num = 2.0 * np.count_nonzero(s)
        out["flowT"], out["flowRT"] = sT / (num + 0.1), sRT / (num + 0.1)
    ok = (Ku > 2) & (Kv > 2) & (Ku < wl - 3) & (Kv < hl - 3) & (new_id > 0)
    Ku, Kv, u, v, new_id, col = Ku[ok], Kv[ok], u[ok], v[ok], new_id[ok], col[ok]
    hit = _interp33(np.asarray(dI, dtype=np.float64), Ku, Kv)
    fin = np.isfinite(hit[:, 0])
    hit, u, v, new_id, col = hit[fin], u[fin], v[fin], new_id[fin], col[fin]
    r = hit[:, 0] - (affLL[0] * col + affLL[1])
    hw = np.where(np.abs(r) < huber, 1.0, huber / np.maximum(np.abs(r), 1e-300))
    sat = np.abs(r) > cutoff
    maxE = 2 * huber * cutoff - huber * huber
    out["E"] = float(np.sum(np.where(sat, maxE, hw * r * r * (2 - hw))))
    out["numTermsInE"], out["numSaturated"] = int(len(r)), int(np.count_nonzero(sat))
    g = ~sat
    out["numWarped"] = int(np.count_nonzero(g))
    dx, dy = hit[g, 1] * fx, hit[g, 2] * fy
    u, v, idn, r, hw, col = u[g], v[g], new_id[g], r[g], hw[g], col[g]
    J = np.stack([idn * dx, idn * dy, -idn * (u * dx + v * dy), -(u * v * dx + dy * (1 + v * v)), u * v * dy + dx * (1 + u * u),
                  u * dy - v * dx, affLL[0] * (b0 - col), -np.ones_like(u), r], axis=1)
    H9 = (J * hw[:, None]).T @ J
    n = (out["numWarped"] + 3) // 4 * 4
    out["H"], out["b"] = H9[:8, :8] / max(n, 1), H9[:8, 8] / max(n, 1)
    return out


# ------------------------------------------------------------------------------------------------
# EnergyFunctional::marginalizeFrame, visual part (OB/EnergyFunctional.cpp:788-859): restated as index selection + a scaled Schur
# complement instead of the block moves of the reference
# ------------------------------------------------------------------------------------------------
def marginalize_frame(HM, bM, idx, prior8, delta_prior8):
    HM, bM = np.asarray(HM, dtype=np.float64), np.asarray(bM, dtype=np.float64)
    odim = HM.shape[0]
    io = 4 + 8 * idx
    gone = np.arange(io, io + 8)
    keep = np.array([i for i in range(odim) if i < io or i >= io + 8])
    order = np.concatenate([keep, gone])
    H, b = HM[np.ix_(order, order)].copy(), bM[order].copy()
    nd = odim - 8
    H[nd:, nd:] += np.diag(prior8)
    b[nd:] += np.asarray(prior8) * np.asarray(delta_prior8)
    S = np.sqrt(np.abs(np.diag(H)) + 10.0)
    Hs, bs = H / np.outer(S, S), b / S
    hpi = np.linalg.inv(Hs[nd:, nd:])
    bli = Hs[nd:, :nd].T @ hpi
    Ht = Hs[:nd, :nd] - bli @ Hs[nd:, :nd]
    bt = bs[:nd] - bli @ bs[nd:]
    Ht, bt = Ht * np.outer(S[:nd], S[:nd]), bt * S[:nd]
    return 0.5 * (Ht + Ht.T), bt


def solve_system(H_top, b_top, H_sc, b_sc, HM, bM, delta, lam=1e-5):
    """EnergyFunctional::solveSystemF with the IMU off (OB/EnergyFunctional.cpp:1046-1148) from its pieces: H_top = HL_top + HA_top
    (priors in), the prior around delta (bM + HM delta), (1 + lambda) on the diagonal, H_sc * (1.0f / (1 + lambda)) -- a DOUBLE
    quotient: float literal over a double sum --, Jacobi scaling by (diagonal + 10)^-1/2, solve.  Independent of the C code: NumPy,
    the lower triangles mirrored (what Eigen's ldlt() reads), extended-precision residuals in an iterative refinement."""
    LD = np.longdouble

    def low(A):
        A = np.asarray(A, dtype=np.float64)
        return np.tril(A) + np.tril(A, -1).T

    H = (low(H_top) + low(HM)).astype(LD)
    # (bM + HM * delta is a full matrix-vector product, :1090: it sees BOTH triangles of an HM that carries the fp32-level asymmetry of
    # M - Msc after marginalizePointsF; only the factorisation below reads one triangle)
    b = np.asarray(b_top, LD) + np.asarray(bM, LD) + np.asarray(HM, LD) @ np.asarray(delta, LD)
    H[np.diag_indices(len(b))] *= LD(1 + lam)
    H -= low(H_sc).astype(LD) * LD(1.0 / (1 + lam))
    b = b - np.asarray(b_sc, LD)
    s = 1 / np.sqrt(np.diag(H).astype(np.float64) + 10)
    Hs = H * s[:, None] * s[None, :]
    x = np.linalg.solve(Hs.astype(np.float64), (b * s).astype(np.float64)).astype(LD)
    for _ in range(5):
        x = x + np.linalg.solve(Hs.astype(np.float64), (b * s - Hs @ x).astype(np.float64)).astype(LD)
    return (x * s).astype(np.float64)
