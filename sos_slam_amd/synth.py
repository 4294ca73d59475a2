"""Deterministic synthetic keyframe windows.

A window is n keyframes looking at one smooth textured surface, P active points spread over the
hosts and a PointFrameResidual for every (point, other frame) whose centre projects in bounds.  The
generator only produces *inputs* (images, poses, points, residual graph, marginalisation prior); it
contains none of the path's arithmetic beyond what creates a point in the reference:

  * intrinsics: EuRoC cam0 under DSO's relative-format rule (tests/EuRoC/camera0.txt:1,
    util/Undistort.cpp:749-766), scaled with the image width
  * point colour / weights: ImmaturePoint ctor, FS/ImmaturePoint.cpp:36-53 (integer pixel, forward
    differences of getInterpolatedElement33BiLin, util/globalFuncs.h:161-182)
  * default frameEnergyTH 8*8*patternNum: FS/HessianBlocks.h:269

Named windows: W7 (7 KF x 2000 pts, the reference preset), W12 (12 x 4096, headline), W16 (16 x 8192).
"""
from __future__ import annotations

import dataclasses

import numpy as np

SEED = 20260929
PATTERN = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], dtype=np.int32)

SCALE_XI_TRANS = 0.5
SCALE_XI_ROT = 1.0
SCALE_A = 10.0
SCALE_B = 1000.0

WINDOWS = {
    "W7": dict(n=7, P=2000, w=752, h=480),
    "W12": dict(n=12, P=4096, w=752, h=480),
    "W16": dict(n=16, P=8192, w=752, h=480),
    # small parity cases
    "T3": dict(n=3, P=64, w=96, h=64),
    "T4": dict(n=4, P=256, w=160, h=128),
    "T6": dict(n=6, P=900, w=320, h=240),
}

# numpy mirrors of the C records in include/sos_slam.h
POINT_DTYPE = np.dtype(
    [("u", "f4"), ("v", "f4"), ("idepth_scaled", "f4"), ("idepth_zero_scaled", "f4"),
     ("color", "f4", (8,)), ("weights", "f4", (8,)), ("priorF", "f4"), ("deltaF", "f4"),
     ("host", "i4"), ("pad", "i4")], align=True)
RESID_DTYPE = np.dtype(
    [("point", "i4"), ("host", "i4"), ("target", "i4"), ("flags", "u4"), ("state_state", "i4"),
     ("state_energy", "f4")], align=True)
PRECALC_DTYPE = np.dtype(
    [("PRE_KRKiTll", "f4", (9,)), ("PRE_KtTll", "f4", (3,)), ("PRE_RTll_0", "f4", (9,)),
     ("PRE_tTll_0", "f4", (3,)), ("PRE_aff_mode", "f4", (2,)), ("PRE_b0_mode", "f4"), ("pad", "f4")],
    align=True)
RAWJAC_DTYPE = np.dtype(
    [("resF", "f4", (8,)), ("Jpdxi", "f4", (2, 6)), ("Jpdc", "f4", (2, 4)), ("Jpdd", "f4", (2,)),
     ("JIdx", "f4", (2, 8)), ("JabF", "f4", (2, 8)), ("JIdx2", "f4", (4,)), ("JabJIdx", "f4", (4,)),
     ("Jab2", "f4", (4,))], align=True)
FRAME_INIT_DTYPE = np.dtype(
    [("camToWorld", "f8", (12,)), ("state", "f8", (10,)), ("ab_exposure", "f4"), ("frameID", "i4"),
     ("frameEnergyTH", "f4"), ("pad", "i4")], align=True)

RF_ACTIVE, RF_LINEARIZED, RF_ISNEW = 1, 2, 4
RES_IN, RES_OOB, RES_OUTLIER = 0, 1, 2


def default_params(w: int, h: int) -> dict:
    """util/settings.cpp defaults read by the path."""
    return dict(w=w, h=h, huberTH=9.0, outlierTHSumComponent=50.0 * 50.0, affineOptModeA=1e12,
                affineOptModeB=1e8, idepthFixPrior=50.0 * 50.0, idepthFixPriorMargFac=600.0 * 600.0,
                margWeightFac=0.25, initialCalibHessian=5e9, coarseCutoffTH=20.0, frameEnergyTHN=0.7,
                frameEnergyTHFacMedian=1.5, frameEnergyTHConstWeight=0.5, overallEnergyTHWeight=1.0)


def so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def se3_exp12(tangent6):
    """Sophus convention (thirdparty/Sophus/sophus/se3.hpp:550-585): tangent = (upsilon, omega); returns R row-major | t."""
    a = np.asarray(tangent6, dtype=np.float64)
    ups, w = a[:3], a[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
    R = so3_exp(w)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * K
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    return np.concatenate([R.reshape(-1), V @ ups])


def se3_mul12(A, B):
    Ra, ta, Rb, tb = A[:9].reshape(3, 3), A[9:], B[:9].reshape(3, 3), B[9:]
    return np.concatenate([(Ra @ Rb).reshape(-1), Ra @ tb + ta])


def se3_inv12(A):
    R, t = A[:9].reshape(3, 3), A[9:]
    return np.concatenate([R.T.reshape(-1), -R.T @ t])


@dataclasses.dataclass
class Window:
    name: str
    n: int
    w: int
    h: int
    K: np.ndarray            # fx fy cx cy (float64, value_scaled)
    images: np.ndarray       # (n, h, w) float32 irradiance
    frames: np.ndarray       # FRAME_INIT_DTYPE[n]
    points: np.ndarray       # POINT_DTYPE[P]
    resid: np.ndarray        # RESID_DTYPE[R]
    HM: np.ndarray           # (4+8n, 4+8n) float64
    bM: np.ndarray           # (4+8n,) float64
    params: dict
    extra_images: np.ndarray = None   # (k, h, w): frames that are NOT keyframes of the window (tracker inputs)
    extra_poses: np.ndarray = None    # (k, 12) camToWorld (R row-major | t) used to render them
    extra_aff: np.ndarray = None      # (k, 2) affine brightness (a, b) applied when rendering them
    stereo_tfm: np.ndarray = None     # (12,) tfmF0ToF1 of the stereo partner extra[1]: p1 = R p0 + t

    @property
    def P(self) -> int:
        return len(self.points)

    @property
    def R(self) -> int:
        return len(self.resid)


class _Scene:
    """Smooth surface z = Z(x, y) in world coordinates with a band-limited texture."""

    def __init__(self, rng: np.random.Generator, scale: float = 1.0, nsin: int = 64):
        self.amp = rng.uniform(0.5, 1.0, nsin)
        ang = rng.uniform(0, 2 * np.pi, nsin)
        freq = rng.uniform(15.0, 90.0, nsin) * scale  # rad / m; ~10 grey levels / pixel at any image size
        self.fx = freq * np.cos(ang)
        self.fy = freq * np.sin(ang)
        self.ph = rng.uniform(0, 2 * np.pi, nsin)
        self.amp *= 55.0 / np.sqrt(0.5 * np.sum(self.amp ** 2))  # texture RMS ~55 grey levels

    @staticmethod
    def depth(x, y):
        return 2.2 + 0.5 * np.sin(0.8 * x + 0.3) * np.cos(0.6 * y - 0.2) + 0.15 * x - 0.1 * y

    def texture(self, x, y):
        out = np.full(x.shape, 128.0, dtype=np.float32)
        xf = x.astype(np.float32)
        yf = y.astype(np.float32)
        for a, fx, fy, ph in zip(self.amp, self.fx, self.fy, self.ph):
            out += np.float32(a) * np.sin(np.float32(fx) * xf + np.float32(fy) * yf + np.float32(ph))
        return out

    def render(self, R, t, K, w, h):
        """Ray-cast camToWorld (R, t). Returns (image, camera-frame depth along z)."""
        fx, fy, cx, cy = K
        uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        d = np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], axis=-1)  # camera rays, z = 1
        dw = d @ R.T
        s = np.full(uu.shape, 2.2)
        for _ in range(8):  # fixed-point ray / surface intersection
            px = t[0] + s * dw[..., 0]
            py = t[1] + s * dw[..., 1]
            s = (self.depth(px, py) - t[2]) / dw[..., 2]
        px = t[0] + s * dw[..., 0]
        py = t[1] + s * dw[..., 1]
        return self.texture(px, py), s


def make_window(name: str = "W12", seed: int = SEED, noise_sigma: float = 1.0, state_noise: float = 3e-4,
                idepth_noise: float = 0.002, point_seed: int | None = None, extra_frames: int = 0,
                stereo_baseline: float = 0.11, **override) -> Window:
    cfg = dict(WINDOWS[name]) if name in WINDOWS else {}
    cfg.update(override)
    n, P, w, h = cfg["n"], cfg["P"], cfg["w"], cfg["h"]
    rng = np.random.default_rng(seed)
    s = w / 752.0
    K = np.array([458.654 * s, 457.296 * s, (367.215 + 0.5) * s - 0.5, (248.375 + 0.5) * s - 0.5])
    K = K.astype(np.float32).astype(np.float64)  # value_scaled is set from float globals fxG.. in the reference
    scene = _Scene(rng, s)

    frames = np.zeros(n, dtype=FRAME_INIT_DTYPE)
    images = np.zeros((n, h, w), dtype=np.float32)
    depth = np.zeros((n, h, w), dtype=np.float64)
    aff_true = np.zeros((n, 2))
    for i in range(n):
        R = so3_exp(0.01 * i * np.array([0.3, 1.0, 0.2]))
        t = 0.08 * i * np.array([1.0, 0.1, 0.05])
        a_i, b_i = (0.0, 0.0) if i == 0 else (rng.normal(0, 0.02), rng.normal(0, 2.0))
        aff_true[i] = (a_i, b_i)
        img, dep = scene.render(R, t, K, w, h)
        img = np.exp(a_i) * img + b_i + rng.normal(0, noise_sigma, img.shape)
        images[i] = np.clip(img, 0.0, 255.0).astype(np.float32)
        depth[i] = dep
        frames[i]["camToWorld"][:9] = R.reshape(-1)
        frames[i]["camToWorld"][9:] = t
        st = np.zeros(10)
        if i > 0:  # frame 0 carries the gauge prior (frameID 0) and stays at its evalPT
            st[:6] = rng.normal(0, state_noise, 6)
        st[6] = (a_i + (rng.normal(0, 0.002) if i > 0 else 0.0)) / SCALE_A
        st[7] = (b_i + (rng.normal(0, 0.2) if i > 0 else 0.0)) / SCALE_B
        frames[i]["state"] = st
        frames[i]["ab_exposure"] = 1.0
        frames[i]["frameID"] = i
        frames[i]["frameEnergyTH"] = 8 * 8 * 8

    # frames outside the window, for the coarse tracker: extra[0] continues the trajectory (the "new frame"),
    # extra[1] is a stereo partner of the newest keyframe: cam1 = cam0 shifted by the baseline along x plus the
    # millimetre-level y/z offsets every calibrated rig has (with tz == 0 exactly the reference's calcGSSSEScale
    # turns its zero-padded SSE lanes into inf * 0 = NaN, FS/ScaleOptimizer.cpp:255-259 -- not a case to pin)
    stereo_off = np.array([stereo_baseline, 0.0012, 0.0021])
    ex_img, ex_pose, ex_aff = [], [], []
    for k in range(extra_frames):
        if k == 1:
            Rk = frames[n - 1]["camToWorld"][:9].reshape(3, 3).copy()
            tk = frames[n - 1]["camToWorld"][9:] + Rk @ stereo_off
            a_k, b_k = 0.0, 0.0
        else:
            i = n + k * 0.35
            Rk = so3_exp(0.01 * i * np.array([0.3, 1.0, 0.2]))
            tk = 0.08 * i * np.array([1.0, 0.1, 0.05])
            a_k, b_k = rng.normal(0, 0.02), rng.normal(0, 2.0)
        img, _ = scene.render(Rk, tk, K, w, h)
        img = np.exp(a_k) * img + b_k + rng.normal(0, noise_sigma, img.shape)
        ex_img.append(np.clip(img, 0.0, 255.0).astype(np.float32))
        ex_pose.append(np.concatenate([Rk.reshape(-1), tk]))
        ex_aff.append((a_k, b_k))

    # points: P/n per host at integer pixels (point_seed: a different point set on the same frames, used to
    # give every rank of a multi-GPU run its own shard)
    rng_main = rng
    if point_seed is not None:
        rng = np.random.default_rng(point_seed)
    pts = np.zeros(P, dtype=POINT_DTYPE)
    per = [P // n + (1 if i < P % n else 0) for i in range(n)]
    k = 0
    c = np.float32(50.0 * 50.0)
    for hst in range(n):
        m = per[hst]
        uu = rng.integers(4, w - 4, m)
        vv = rng.integers(4, h - 4, m)
        I = images[hst]
        for j in range(m):
            u, v = int(uu[j]), int(vv[j])
            p = pts[k]
            p["u"], p["v"] = u, v
            idt = 1.0 / depth[hst, v, u]
            idn = np.float32(idt * (1.0 + rng.normal(0, idepth_noise)))
            p["idepth_scaled"] = idn
            p["idepth_zero_scaled"] = idn
            for q in range(8):
                x, y = u + PATTERN[q, 0], v + PATTERN[q, 1]
                tl, tr, bl = I[y, x], I[y, x + 1], I[y + 1, x]
                gx, gy = np.float32(tr - tl), np.float32(bl - tl)
                p["color"][q] = tl
                p["weights"][q] = np.sqrt(c / (c + (gx * gx + gy * gy)), dtype=np.float32)
            p["priorF"] = 0.0
            p["deltaF"] = 0.0
            p["host"] = hst
            k += 1

    # residual graph: every (point, other frame) whose centre projects in bounds
    Rm = [frames[i]["camToWorld"][:9].reshape(3, 3) for i in range(n)]
    tm = [frames[i]["camToWorld"][9:] for i in range(n)]
    res_list = []
    fx, fy, cx, cy = K
    for pi in range(P):
        p = pts[pi]
        hst = int(p["host"])
        X = np.array([(p["u"] - cx) / fx, (p["v"] - cy) / fy, 1.0]) / float(p["idepth_scaled"])
        Xw = Rm[hst] @ X + tm[hst]
        for t_ in range(n):
            if t_ == hst:
                continue
            Xc = Rm[t_].T @ (Xw - tm[t_])
            if Xc[2] <= 0.05:
                continue
            Ku, Kv = fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy
            if Ku > 5.0 and Kv > 5.0 and Ku < w - 7.0 and Kv < h - 7.0:
                res_list.append((pi, hst, t_, RF_ISNEW, RES_IN, 0.0))
    resid = np.array(res_list, dtype=RESID_DTYPE) if res_list else np.zeros(0, dtype=RESID_DTYPE)

    rng = rng_main
    dim = 4 + 8 * n
    A = rng.normal(0, 1.0, (dim, dim))
    HM = 10.0 * (A @ A.T) / dim + 100.0 * np.eye(dim)
    bM = rng.normal(0, 1.0, dim)
    return Window(name=name, n=n, w=w, h=h, K=K, images=images, frames=frames, points=pts, resid=resid,
                  HM=HM, bM=bM, params=default_params(w, h),
                  extra_images=np.stack(ex_img) if ex_img else None,
                  extra_poses=np.stack(ex_pose) if ex_pose else None,
                  extra_aff=np.array(ex_aff) if ex_aff else None,
                  stereo_tfm=np.concatenate([np.eye(3).reshape(-1), -stereo_off]))


def shard_points(win: Window, rank: int, world: int) -> np.ndarray:
    """Contiguous slices of the allPoints order balanced by residual count."""
    counts = np.bincount(win.resid["point"], minlength=win.P).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(counts)])
    total = csum[-1]
    bounds = [int(np.searchsorted(csum, total * r / world, side="left")) for r in range(world + 1)]
    bounds[0], bounds[-1] = 0, win.P
    return np.arange(bounds[rank], bounds[rank + 1])


def take_shard(win: Window, point_idx: np.ndarray) -> Window:
    """Sub-window holding only `point_idx` (and their residuals); frames/images replicated."""
    remap = -np.ones(win.P, dtype=np.int64)
    remap[point_idx] = np.arange(len(point_idx))
    keep = remap[win.resid["point"]] >= 0
    resid = win.resid[keep].copy()
    resid["point"] = remap[resid["point"]]
    return dataclasses.replace(win, points=win.points[point_idx].copy(), resid=resid)


def make_imu_records(win, weights=True, valid=True, seed=0, trapped=1, consistent=False, dt=0.1):
    """IMU settings, calibration part and one sosf_imu_frame record per keyframe of a synthetic window (eight samples before each
    keyframe): the inputs of sosf_set_imu for parity tests and for bench.py --imu.  Returns (settings, calib, records, keep) --
    `keep` holds the sample arrays the records point at.
    consistent = False: arbitrary samples and small random IMU states (exercises the assembly; a long loop on it drifts away).
    consistent = True: the samples an IMU would deliver on make_window's trajectory (keyframe i at rotation exp(0.01 i axis),
    translation 0.08 i (1, 0.1, 0.05): constant velocity and rate, so the accelerometer sees gravity only), zero IMU states: the
    visual-inertial problem then has the visual solution as its fixed point and a long loop converges on it."""
    from .records import ImuCalib, ImuFrame, ImuSettings
    rng = np.random.default_rng(seed)
    S = ImuSettings()
    S.weight_imu[:] = list((np.eye(6) * (4.0 if weights else 0.0)).reshape(-1))
    S.weight_imu_bias[:] = list((np.eye(6) * (10.0 if weights else 0.0)).reshape(-1))
    S.gravity[:] = [0, 9.81, 0]
    S.rot_imu_cam[:] = list(so3_exp(np.array([0.1, -0.2, 0.05])).reshape(-1))
    S.maxImuInterval = 0.5
    S.enable_scale_opt = 0
    cal = ImuCalib(1.0 / 200, 1.0 / 200, int(trapped), 1)
    frames, keep = [], []
    for i in range(win.n):
        f = ImuFrame()
        f.timestamp = 1.0 + 0.1 * i
        st = rng.normal(0, 1e-4, 21)
        f.state_imu[:] = list(st)
        f.state_imu_zero[:] = list(st)
        f.trackingRefIsPrev = 1 if valid else 0
        imu = np.zeros((8, 7))
        imu[:, 0] = f.timestamp - np.linspace(0.07, 0.0, 8)
        imu[:, 1:4] = [0.2, 9.6, -0.1]
        imu[:, 4:7] = rng.normal(0, 0.02, (8, 3))
        if consistent:
            Ric, g, axis = so3_exp(np.array([0.1, -0.2, 0.05])), np.array([0.0, 9.81, 0.0]), np.array([0.3, 1.0, 0.2])
            f.timestamp = 1.0 + dt * i
            f.state_imu[:] = [0.0] * 21
            f.state_imu_zero[:] = [0.0] * 21
            imu = np.zeros((8 if i > 0 else 0, 7))
            for j in range(len(imu)):
                tau = (i - 1) + (j + 1) / 8.0
                imu[j, 0] = 1.0 + dt * tau
                imu[j, 1:4] = Ric @ so3_exp(0.01 * tau * axis).T @ g
                imu[j, 4:7] = Ric @ (0.01 * axis / dt)
            f.trackingRefIsPrev = 1 if (valid and i > 0) else 0
        imu = np.ascontiguousarray(imu)
        keep.append(imu)
        f.n_imu = len(imu)
        f.imu = imu.ctypes.data if len(imu) else None
        frames.append(f)
    return S, cal, frames, keep


def expand_prior_imu(win, eps=1e-3):
    """the window's marginalisation prior in the expanded (IMU) dimension CPARS + 1 + 29 n (expandHbtoFitImu,
    OB/EnergyFunctional.cpp:256-286) plus eps on the diagonal"""
    from .records import imu_dim
    n = win.n
    d0, dI = 4 + 8 * n, imu_dim(n)
    idx = np.array([k if k < 4 else 5 + 29 * ((k - 4) // 8) + (k - 4) % 8 for k in range(d0)])
    HMi, bMi = np.zeros((dI, dI)), np.zeros(dI)
    HMi[np.ix_(idx, idx)] = win.HM
    bMi[idx] = win.bM
    return HMi + np.eye(dI) * eps, bMi
