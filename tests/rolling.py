"""Rolling-window harness: a synthetic keyframe sequence driven through FullSystem::makeKeyFrame's backend chain
(FS/FullSystem.cpp:783-931) twice -- once on the device through the C++ facade, once through the CPU oracle -- so that the
poses that LEAVE the window (the marginalised poses of the north star) and every index set on the way can be compared.

Per new frame, both chains run, in the reference's order:
  undistortion + pyramid -> trackNewestCoarse against the newest keyframe -> traceNewCoarse of all immature points ->
  flagFramesForMarginalization -> insertFrame -> residuals of the old points towards the new keyframe -> activatePointsMT
  (distance-map selection, optimizeImmaturePoint) -> optimize(6) -> removeOutliers -> setCoarseTrackingRef ->
  flagPointsForRemoval + dropPointsF + marginalizePointsF -> makeNewTraces (pixel selection, ImmaturePoint constructors) ->
  marginalizeFrame for the flagged keyframes.

The stage ORDER and the bookkeeping of the immature points (which the facade leaves with the caller) are shared code
(`Chain`); everything else differs: the device chain calls the facade's C++ (flagFramesForMarginalization,
flagPointsForRemoval, removeOutliers, marginalizeFrame ... in csrc/host/sos_host.cpp) and the HIP kernels, the oracle chain
keeps its own Python graph of frames / points / residuals (`OGraph`, a second statement of the same reference logic) and
calls the C restatement for the arithmetic.
"""
from __future__ import annotations

import dataclasses

import os

import numpy as np

from oracle import oracle as orc
from sos_slam_amd import synth
from tests.immature_helpers import kinv_f32, mm3_f32
from sos_slam_amd.records import (ACT_ACTIVATED, ACT_DELETE, IMMATURE_DTYPE, PAIR_TFM_DTYPE, ActivateParams, Calib, PixselParams,
                                  TraceParams, random_pattern)

SCALE_A, SCALE_B = synth.SCALE_A, synth.SCALE_B
IPS_GOOD, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED = 0, 1, 2, 3, 4, 5
FRAME_INIT_EX_DTYPE = np.dtype([("camToWorld", "f8", (12,)), ("state", "f8", (10,)), ("state_zero", "f8", (10,)), ("ab_exposure", "f4"),
                                ("frameID", "i4"), ("frameEnergyTH", "f4"), ("pad", "i4")], align=True)
# reference settings read by the host logic (util/settings.cpp:61-96)
MIN_IDEPTH_H_MARG = 50.0
MIN_POINTS_REMAINING, MAX_LOG_AFF_FAC = 0.05, 0.7
MIN_FRAMES, MAX_FRAMES, MIN_FRAME_AGE = 5, 7, 1
MIN_GOOD_ACTIVE_RES, MIN_GOOD_RES = 3, 4


def se3_inv(T):
    return synth.se3_inv12(T)


def se3_mul(A, B):
    return synth.se3_mul12(A, B)


# ------------------------------------------------------------------------------------------------
# the synthetic sequence
# ------------------------------------------------------------------------------------------------
class Scenario:
    """A camera moving past the textured surface of sos_slam_amd.synth: raw 8-bit frames, ground-truth poses, a bootstrap
    window (what the initialiser would hand over: n0 keyframes at their true poses with noisy inverse depths)."""

    def __init__(self, w=320, h=240, n_frames=26, n0=4, points0=360, seed=synth.SEED + 77, step=0.07, noise_sigma=1.0,
                 desired_points=1000.0, immature_density=450.0, vio=False, stereo=False, kf_every=1, rot=0.008):
        self.w, self.h, self.n_frames, self.n0 = w, h, n_frames, n0
        self.vio, self.stereo = vio, stereo
        self.scale_opt_thres = 12.0          # tests/EuRoC/euroc.launch: scale_opt_thres
        self.raw_right = []
        stereo_off = np.array([0.11, 0.0012, 0.0021])
        self.stereo_tfm = np.concatenate([np.eye(3).reshape(-1), -stereo_off])   # tfmF0ToF1: p1 = R p0 + t
        self.desired_points, self.immature_density = desired_points, immature_density
        # kf_every > 1: only every kf_every-th frame becomes a keyframe; the frames between are tracked and traced (makeNonKeyFrame)
        self.kf_every = int(kf_every)
        assert self.kf_every == 1 or not (vio or stereo)
        rng = np.random.default_rng(seed)
        s = w / 752.0
        K = np.array([458.654 * s, 457.296 * s, (367.215 + 0.5) * s - 0.5, (248.375 + 0.5) * s - 0.5])
        self.K = K.astype(np.float32).astype(np.float64)
        self.scene = synth._Scene(rng, s)
        self.poses, self.raw, self.depth, self.aff_true = [], [], [], []
        if vio:
            self._make_imu(n_frames, step)
        for i in range(n_frames):
            R = synth.so3_exp(rot * i * np.array([0.3, 1.0, 0.2]))
            t = step * i * np.array([1.0, 0.1, 0.05])
            if vio:
                R, t = self._traj(float(i))
            a_i, b_i = (0.0, 0.0) if i == 0 else (rng.normal(0, 0.01), rng.normal(0, 1.0))
            img, dep = self.scene.render(R, t, self.K, w, h)
            img = np.exp(a_i) * img + b_i + rng.normal(0, noise_sigma, img.shape)
            self.raw.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
            if stereo:    # the stereo partner: same intrinsics, shifted by the baseline in the camera frame, same exposure
                img1, _ = self.scene.render(R, t + R @ stereo_off, self.K, w, h)
                img1 = np.exp(a_i) * img1 + b_i + rng.normal(0, noise_sigma, img1.shape)
                self.raw_right.append(np.clip(np.rint(img1), 0, 255).astype(np.uint8))
            self.depth.append(dep)
            self.poses.append(np.concatenate([R.reshape(-1), t]))
            self.aff_true.append((a_i, b_i))
        self.cam_txt = f"Pinhole {self.K[0]:.9g} {self.K[1]:.9g} {self.K[2]:.9g} {self.K[3]:.9g} 0\n{w} {h}\nnone\n{w} {h}\n"
        self.params = synth.default_params(w, h)
        self.pattern = random_pattern(w * h)
        self.seed_boot = seed + 1
        self.points0 = points0

    # ---- visual-inertial variant: a trajectory with acceleration, and the IMU samples that belong to it
    def _traj(self, tau, step=None):
        """camToWorld at frame time tau (frames are dt apart): the straight path plus a small loop"""
        step = self._step if step is None else step
        R = synth.so3_exp(0.008 * tau * self._axis)
        t = step * tau * np.array([1.0, 0.1, 0.05]) + self._amp * (np.sin(self._om * tau) * self._e1 + (1 - np.cos(self._om * tau)) * self._e2)
        return R, t

    def _make_imu(self, n_frames, step, m=10):
        from sos_slam_amd.records import ImuSettings
        self._step, self._axis = step, np.array([0.3, 1.0, 0.2])
        self._amp, self._om = 0.012, 0.8
        self._e1, self._e2 = np.array([0.0, 1.0, 0.0]), np.array([0.0, 0.0, 1.0])
        self.dt = 0.05
        self.ts = 1.0 + self.dt * np.arange(n_frames)
        self.scale_true, self.bias_g = 1.0, np.array([0.004, -0.003, 0.002])
        S = ImuSettings()
        S.weight_imu[:] = list(np.diag([25.0, 25.0, 25.0, 2500.0, 2500.0, 2500.0]).reshape(-1))
        S.weight_imu_bias[:] = list(np.diag([1e3, 1e3, 1e3, 1e5, 1e5, 1e5]).reshape(-1))
        S.gravity[:] = [0.0, 9.81, 0.0]
        Ric = synth.so3_exp(np.array([0.1, -0.2, 0.05]))
        S.rot_imu_cam[:] = list(Ric.reshape(-1))
        S.maxImuInterval, S.enable_scale_opt = 0.5, int(self.stereo)
        self.imu_settings = S
        g = np.array(S.gravity[:])
        self.imu = [np.zeros((0, 7))]
        for k in range(1, n_frames):
            rows = np.zeros((m, 7))
            for j in range(m):
                tau = (k - 1) + (j + 1) / m
                R, _ = self._traj(tau)
                a_w = self._amp * self._om ** 2 * (-np.sin(self._om * tau) * self._e1 + np.cos(self._om * tau) * self._e2) / self.dt ** 2
                rows[j, 0] = 1.0 + self.dt * tau
                rows[j, 1:4] = Ric @ R.T @ (self.scale_true * a_w + g)
                rows[j, 4:7] = Ric @ (0.008 * self._axis / self.dt) + self.bias_g
            self.imu.append(rows)

    def bootstrap_points(self, images):
        """(POINT_DTYPE records, residual (point, target idx) list) of the bootstrap window from its undistorted images"""
        n0, w, h = self.n0, self.w, self.h
        rng = np.random.default_rng(self.seed_boot)     # the same points for every chain
        pts = np.zeros(self.points0, dtype=synth.POINT_DTYPE)
        per = [self.points0 // n0 + (1 if i < self.points0 % n0 else 0) for i in range(n0)]
        c = np.float32(50.0 * 50.0)
        k = 0
        taken = set()
        for hst in range(n0):
            I = images[hst]
            m = 0
            while m < per[hst]:
                u, v = int(rng.integers(6, w - 6)), int(rng.integers(6, h - 6))
                if (hst, u, v) in taken:
                    continue
                taken.add((hst, u, v))
                p = pts[k]
                p["u"], p["v"] = u, v
                idn = np.float32((1.0 / self.depth[hst][v, u]) * (1.0 + rng.normal(0, 0.01)))
                p["idepth_scaled"] = p["idepth_zero_scaled"] = idn
                for q in range(8):
                    x, y = u + synth.PATTERN[q, 0], v + synth.PATTERN[q, 1]
                    tl, tr, bl = I[y, x], I[y, x + 1], I[y + 1, x]
                    gx, gy = np.float32(tr - tl), np.float32(bl - tl)
                    p["color"][q] = tl
                    p["weights"][q] = np.sqrt(c / (c + (gx * gx + gy * gy)), dtype=np.float32)
                p["host"] = hst
                k += 1
                m += 1
        fx, fy, cx, cy = self.K
        res = []
        for pi in range(len(pts)):
            p = pts[pi]
            hst = int(p["host"])
            Rh, th = self.poses[hst][:9].reshape(3, 3), self.poses[hst][9:]
            X = np.array([(p["u"] - cx) / fx, (p["v"] - cy) / fy, 1.0]) / float(p["idepth_scaled"])
            Xw = Rh @ X + th
            for t_ in range(n0):
                if t_ == hst:
                    continue
                Rt, tt = self.poses[t_][:9].reshape(3, 3), self.poses[t_][9:]
                Xc = Rt.T @ (Xw - tt)
                if Xc[2] <= 0.05:
                    continue
                Ku, Kv = fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy
                if 5.0 < Ku < w - 7.0 and 5.0 < Kv < h - 7.0:
                    res.append((pi, t_))
        return pts, res


# ------------------------------------------------------------------------------------------------
# geometry every chain derives from ITS OWN poses / calibration, in float as the reference does
# ------------------------------------------------------------------------------------------------
def host_to_frame(K4, host_c2w, frame_c2w, host_aff, frame_aff):
    """KRKi, Kt, aff of FullSystem::traceNewCoarse (FS/FullSystem.cpp:326-336), exposures 1"""
    K = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], dtype=np.float32)
    T = se3_mul(se3_inv(frame_c2w), host_c2w)
    R, t = T[:9].reshape(3, 3).astype(np.float32), T[9:].astype(np.float32)
    KRKi = mm3_f32(mm3_f32(K, R), kinv_f32(*K4))
    a = np.exp(frame_aff[0] - host_aff[0])
    return KRKi.reshape(-1), mm3_f32(K, t), np.array([a, frame_aff[1] - a * host_aff[1]], dtype=np.float32)


def level1_to_newest(K4, poses, newest):
    fx, fy, cx, cy = [np.float32(x) for x in K4]
    K0 = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
    K1 = np.array([[fx * np.float32(0.5), 0, (cx + np.float32(0.5)) / np.float32(2) - np.float32(0.5)],
                   [0, fy * np.float32(0.5), (cy + np.float32(0.5)) / np.float32(2) - np.float32(0.5)], [0, 0, 1]], dtype=np.float32)
    Ki0 = kinv_f32(fx, fy, cx, cy)
    KRKi, Kt = [], []
    for f in range(len(poses)):
        T = se3_mul(se3_inv(poses[newest]), poses[f])
        R, t = T[:9].reshape(3, 3).astype(np.float32), T[9:].astype(np.float32)
        KRKi.append(mm3_f32(mm3_f32(K1, R), Ki0).reshape(-1))
        Kt.append(mm3_f32(K1, t))
    return np.stack(KRKi), np.stack(Kt)


def pair_tfms(poses, affs):
    n = len(poses)
    out = np.zeros(n * n, dtype=PAIR_TFM_DTYPE)
    for hst in range(n):
        for tgt in range(n):
            T = se3_mul(se3_inv(poses[tgt]), poses[hst])
            o = out[hst + n * tgt]
            o["R"], o["t"] = T[:9].astype(np.float32), T[9:].astype(np.float32)
            a = np.exp(affs[tgt][0] - affs[hst][0])
            o["aff"] = (a, affs[tgt][1] - a * affs[hst][1])
    return out


def next_min_act_dist(d, n_points, desired):  # FS/FullSystem.cpp:377-399, in float like the member it updates
    d = np.float32(d)
    if n_points < desired * 0.66: d = np.float32(d - 0.8)
    if n_points < desired * 0.8: d = np.float32(d - 0.5)
    elif n_points < desired * 0.9: d = np.float32(d - 0.2)
    elif n_points < desired: d = np.float32(d - 0.1)
    if n_points > desired * 1.5: d = np.float32(d + 0.8)
    if n_points > desired * 1.3: d = np.float32(d + 0.5)
    if n_points > desired * 1.15: d = np.float32(d + 0.2)
    if n_points > desired: d = np.float32(d + 0.1)
    return float(min(max(d, np.float32(0)), np.float32(4)))


# ------------------------------------------------------------------------------------------------
# shared stage order + immature-point bookkeeping
# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class KeyframeLog:
    frameID: int
    tracked_pose: np.ndarray
    tracked_aff: np.ndarray
    flagged: list
    activated: list          # (host frameID, u, v) in insertion order
    deleted_immature: int
    n_points_before_opt: int
    rmse: float
    iterations: int
    window_ids: list
    window_poses: dict       # frameID -> camToWorld (12) after optimize()
    residual_set: set        # (host frameID, u, v, target frameID) after optimize()
    outliers_removed: int
    marg_points: int
    dropped_points: int
    point_set_after: set     # (host frameID, u, v) after flagPointsForRemoval
    new_immature: int
    marginalized: list       # [(frameID, camToWorld (12))]
    HM: np.ndarray
    bM: np.ndarray
    nonkf: list = None       # [(frameID, refToNew (12))] of the frames tracked + traced since the previous keyframe (kf_every > 1)
    vio: dict = None         # visual-inertial runs: scale, scale_zero, trapped, states {frameID: 21}, vel {frameID: 3}, HMi, bMi


class Chain:
    """Stage order of FullSystem::makeKeyFrame and the immature-point containers (FrameHessian::immaturePoints)."""

    def __init__(self, sc: Scenario):
        self.sc = sc
        self.tprm, self.aprm, self.pprm = TraceParams.default(), ActivateParams.default(), PixselParams.default()
        self.imm = {}        # frameID -> IMMATURE_DTYPE records, in FrameHessian::immaturePoints order
        self.imm_type = {}   # frameID -> my_type per record
        self.currentMinActDist = 2.0
        self.logs = []
        self.handles = {}    # frameID -> front-end handle of the keyframes in the window
        self.next_frame = sc.n0
        # visual-inertial mode (FS/FullSystem.cpp:800-807, 841-849, 878-886; FS/FullSystemOptimize.cpp:459-479): per keyframe the
        # FrameShell fields, the 21 IMU states (unscaled) and their linearisation point; the IMU part of CalibHessian
        self.vio = bool(getattr(sc, "vio", False))
        self.shells, self.imu_state, self.imu_zero = {}, {}, {}
        self.cal = dict(scale=1.0 / 200.0, scale_zero=1.0 / 200.0, trapped=0, init=0)
        self.scale_queue, self.scale_qi = np.linspace(-10, -100, 10), 0
        self.n_kf_total = 0
        self.stereo = bool(getattr(sc, "stereo", False))
        self.scale_state = [0, 0]     # FullSystem::scaleTrapped, scale_opt_fails
        self.scale_log = []
        self.track_hist = []     # camToWorld of the last tracked frames (initial guess of the tracker when kf_every > 1)
        # FrameHessian::imu_data per keyframe.  A keyframe that leaves the window hands its samples to its successor
        # (FS/FullSystemMarginalize.cpp:226-228), so the lists are per chain, not per scenario
        self.imu = {i: np.array(a, dtype=np.float64).reshape(-1, 7) for i, a in enumerate(sc.imu)} if self.vio else {}

    def imu_hand_over(self, window_before, fid):
        """samples of the leaving keyframe fid in front of those of the keyframe behind it in `window_before`; returns the window without fid"""
        i = window_before.index(fid)
        if self.vio and i + 1 < len(window_before):
            nxt = window_before[i + 1]
            self.imu[nxt] = np.concatenate([self.imu[fid], self.imu[nxt]], axis=0)
        return window_before[:i] + window_before[i + 1:]

    # ---- backend interface (implemented by DeviceChain / OracleChain)
    def n(self): raise NotImplementedError

    # ---- bootstrap: the window the initialiser would hand over
    def bootstrap(self):
        sc = self.sc
        hs = [self.front_end(sc.raw[i]) for i in range(sc.n0)]
        images = [self.irradiance(h) for h in hs]
        pts, res = sc.bootstrap_points(images)
        self.init_window(hs, [sc.poses[i] for i in range(sc.n0)], [sc.aff_true[i] for i in range(sc.n0)], pts, res)
        for i in range(sc.n0):
            self.handles[i] = hs[i]
            self.imm[i] = np.zeros(0, dtype=IMMATURE_DTYPE)
            self.imm_type[i] = np.zeros(0, np.float32)
        if self.vio:
            for i in range(sc.n0):
                # firstFrame->trackingRef = 0, newFrame->trackingRef = firstFrame (FS/FullSystem.cpp:1054-1062)
                self.shells[i] = dict(ts=float(sc.ts[i]), c2w=np.array(sc.poses[i], dtype=np.float64), vel=np.zeros(3), track_ref=i - 1)
                self.imu_state[i], self.imu_zero[i] = np.zeros(21), np.zeros(21)
            self.n_kf_total = sc.n0
        rmse, its = self.optimize(6)
        self.remove_outliers()
        self.tracker_set_ref()
        if self.stereo:
            self.scale_optimization(sc.n0 - 1)
        for i in range(sc.n0):      # makeNewTraces on every bootstrap keyframe: the sequence starts with candidates
            self.make_new_traces(i)
        self.last_rel = None
        return rmse, its

    def make_new_traces(self, frameID):   # FS/FullSystem.cpp:1071-1097
        u, v, typ = self.pixel_select(self.handles[frameID], self.sc.immature_density)
        rec = self.immature_init(self.handles[frameID], u, v)
        ok = np.isfinite(rec["energyTH"])
        self.imm[frameID] = rec[ok].copy()
        self.imm_type[frameID] = typ[ok].astype(np.float32).copy()
        return int(ok.sum())

    # ---- one new frame = one keyframe
    def trace_new_coarse(self, h_new, ids, poses, affs, K4, c2w, aff, keyframe):
        """FullSystem::traceNewCoarse (FS/FullSystem.cpp:311-350): every immature point of every keyframe against the new frame"""
        for i, fid in enumerate(ids):
            if len(self.imm[fid]) == 0:
                continue
            KRKi, Kt, a = host_to_frame(K4, poses[i], c2w, affs[i], aff)
            self.imm[fid] = self.trace(h_new, self.imm[fid], KRKi, Kt, a)

    def step(self):
        """Frames up to and including the next keyframe; None when the sequence ends before one."""
        sc = self.sc
        nonkf = []
        while True:
            if self.next_frame >= sc.n_frames:
                return None
            k = self.next_frame
            self.next_frame += 1
            is_kf = (k - sc.n0) % sc.kf_every == sc.kf_every - 1
            h_new = self.front_end(sc.raw[k])
            ids = self.window_ids()
            ref_pose = self.kf_pose(len(ids) - 1)
            if sc.kf_every == 1:
                # initial guess: constant motion (the last relative pose), FS/FullSystem.cpp:163-200 tries this one first
                if self.last_rel is None:
                    T_init = se3_mul(se3_inv(sc.poses[k]), sc.poses[k - 1])
                    T_init = se3_mul(synth.se3_exp12(np.array([0.002, -0.001, 0.001, 0.001, -0.001, 0.0005])), T_init)
                else:
                    T_init = self.last_rel
            elif len(self.track_hist) >= 1:
                # "no motion since the last frame" (the second entry of the reference's try list, FS/FullSystem.cpp:163-200).  One
                # frame's motion is well inside the tracker's basin; extrapolating two tracked poses instead feeds the tracking
                # error of a short baseline back into the next guess and drifts away within ten keyframes (tried)
                T_init = se3_mul(se3_inv(self.track_hist[-1]), ref_pose)
            else:
                T_init = se3_mul(se3_inv(sc.poses[k]), sc.poses[ids[-1]])
                T_init = se3_mul(synth.se3_exp12(np.array([0.002, -0.001, 0.001, 0.001, -0.001, 0.0005])), T_init)
            ok, T, aff, lres, flow = self.track(h_new, T_init, np.array(self.kf_aff(len(ids) - 1)))
            assert ok, "tracking lost"
            self.last_rel = T.copy()
            c2w = se3_mul(ref_pose, se3_inv(T))          # shell->camToWorld = trackingRef->camToWorld * camToTrackingRef
            self.track_hist = (self.track_hist + [c2w])[-2:]
            n = self.n()
            poses = [self.kf_pose(i) for i in range(n)]
            affs = [self.kf_aff(i) for i in range(n)]
            K4 = self.K()
            self.trace_new_coarse(h_new, ids, poses, affs, K4, c2w, aff, is_kf)
            if is_kf:
                break
            nonkf.append((k, T.copy()))     # makeNonKeyFrame: traced, nothing else; the frame is dropped
            self.release_plain(h_new)
        # ---- makeKeyFrame
        flagged = self.flag_frames([len(self.imm[f]) for f in ids])
        if self.vio:   # fh->setImuData; propagateImuState(allKeyFramesHistory.back(), coarseTracker->lastRef->imu_bias), :800-807
            # shell->trackingRef = coarseTracker->lastRef = the newest keyframe when the frame was tracked (FS/FullSystem.cpp:296)
            self.shells[k] = dict(ts=float(sc.ts[k]), c2w=np.array(c2w, dtype=np.float64), vel=np.zeros(3), track_ref=ids[-1])
            self.imu_state[k], self.imu_zero[k] = np.zeros(21), np.zeros(21)
            if self.cal["init"]:
                last = ids[-1]
                self.vio_propagate(k, last, self.vio_scaled(self.imu_state[last])[:6])
            self.n_kf_total += 1
        self.add_keyframe(h_new, c2w, aff, k)
        self.handles[k] = h_new
        self.imm[k] = np.zeros(0, dtype=IMMATURE_DTYPE)
        self.imm_type[k] = np.zeros(0, np.float32)
        self.add_old_point_residuals()
        activated, deleted = self.activate_points(flagged)
        npts = self.n_points()
        if self.vio and self.n_kf_total == 5:     # imu initialization, :841-849
            wid = self.window_ids()
            assert len(wid) == 5, "the window must still hold the first five keyframes"
            assert self.vio_initialize(wid), "IMU initialization failed"
            self.cal["init"] = 1
        rmse, its = self.optimize(6)
        ids2 = self.window_ids()
        window_poses = {fid: self.kf_pose(i).copy() for i, fid in enumerate(ids2)}
        if self.vio:
            for i, fid in enumerate(ids2):        # shell->camToWorld = PRE_camToWorld, FS/FullSystemOptimize.cpp:437-443
                self.shells[fid]["c2w"] = np.array(window_poses[fid], dtype=np.float64)
            if self.cal["init"]:                  # :459-479
                self.vio_update_vel(ids2[-1], ids2[-2])
                self.imu_zero[ids2[-1]] = self.imu_state[ids2[-1]].copy()
                if self.sc.imu_settings.enable_scale_opt:     # FS/FullSystemOptimize.cpp:471-473
                    self.cal["trapped"] = 1
                if not self.cal["trapped"]:
                    self.vio_try_trap()
                    if self.cal["trapped"]:
                        for fid in ids2:
                            self.imu_zero[fid] = self.imu_state[fid].copy()
        residual_set = self.residual_set()
        nout = self.remove_outliers()
        if self.vio and self.n_kf_total == 5:     # reset imu states for imu initialization, :878-886
            for i, fid in enumerate(ids2):
                self.imu_zero[fid] = self.imu_state[fid].copy()
                if i > 0:
                    self.vio_update_vel(fid, ids2[i - 1])
        self.tracker_set_ref()
        if self.stereo:               # scale optimization, FS/FullSystem.cpp:897-903
            self.scale_optimization(k)
        nmarg, ndrop = self.flag_points_for_removal()
        point_set = self.point_set()
        nimm = self.make_new_traces(k)
        marg = self.marginalize_flagged()
        for fid, _ in marg:
            self.release(self.handles.pop(fid))
            self.imm.pop(fid)
            self.imm_type.pop(fid)
            if self.vio:
                self.imu_state.pop(fid)
                self.imu_zero.pop(fid)
        HM, bM = self.prior()
        vio = None
        if self.vio:
            Hi, bi = self.prior_imu()
            vio = dict(scale=self.cal["scale"], scale_zero=self.cal["scale_zero"], trapped=self.cal["trapped"], init=self.cal["init"],
                       states={f: self.imu_state[f].copy() for f in self.window_ids()}, vel={f: self.shells[f]["vel"].copy() for f in self.window_ids()},
                       HMi=Hi, bMi=bi)
        self.logs.append(KeyframeLog(k, T, np.asarray(aff, dtype=np.float64), [ids[i] for i in np.flatnonzero(flagged)], activated, deleted, npts, rmse,
                                     its, ids2, window_poses, residual_set, nout, nmarg, ndrop, point_set, nimm, marg, HM, bM, nonkf, vio))
        return self.logs[-1]

    # ---- visual-inertial helpers shared by the chains (state layout and records); the arithmetic is per chain
    _IMU_K = np.repeat([100.0, 1.0, 100.0, 1000.0, 1000.0, 1000.0, 1000.0], 3)   # SCALE_BA, BG, SL_ROT, SQ_TRANS, SQ_ROT, SC_TRANS, SC_ROT

    def scale_optimization(self, k):
        """FullSystem::optimizeScale on the stereo partner of keyframe k; HCalib.setScaleScaledZero when it is accepted"""
        h = self.front_end(self.sc.raw_right[k])
        ref_scale = np.float32(200.0 * self.cal["scale"])      # shell->trackingRef->scale = HCalib.getScaleScaled() of the last optimize
        new_scale, err = self.optimize_scale_kf(h, float(ref_scale))
        self.release_plain(h)
        self.scale_log.append((k, new_scale, err, list(self.scale_state)))
        if new_scale > 0:
            self.cal["scale"] = self.cal["scale_zero"] = float(np.float32(1.0) / np.float32(200.0)) * float(new_scale)

    def vio_scaled(self, state):
        return self._IMU_K * np.asarray(state, dtype=np.float64)

    def vio_records(self, ids):
        """sosf_imu_frame records of the keyframes `ids` (window order) from the chain's state; camToWorld / evalPT_R are
        filled by whoever solves"""
        from sos_slam_amd.records import ImuFrame
        out, keep = [], []
        for i, fid in enumerate(ids):
            f = ImuFrame()
            f.timestamp = self.shells[fid]["ts"]
            f.camToWorld[:] = list(self.shells[fid]["c2w"])
            f.evalPT_R[:] = list(self.shells[fid]["c2w"][:9])
            f.state_imu[:] = list(self.imu_state[fid])
            f.state_imu_zero[:] = list(self.imu_zero[fid])
            # spline_valid needs shell->trackingRef == the shell of the keyframe standing before it in the window NOW
            # (OB/EnergyFunctional.cpp:318, :350): false for the successor of a marginalised middle keyframe
            f.trackingRefIsPrev = 1 if (i > 0 and self.shells[fid]["track_ref"] == ids[i - 1]) else 0
            arr = np.ascontiguousarray(self.imu[fid], dtype=np.float64).reshape(-1, 7)
            keep.append(arr)
            f.n_imu = len(arr)
            f.imu = arr.ctypes.data if len(arr) else None
            out.append(f)
        return out, keep

    def vio_calib(self):
        from sos_slam_amd.records import ImuCalib
        return ImuCalib(self.cal["scale"], self.cal["scale_zero"], int(self.cal["trapped"]), int(self.cal["init"]))

    def vio_take_calib(self, c):
        self.cal.update(scale=c.scale, scale_zero=c.scale_zero, trapped=int(c.scale_trapped), init=int(c.imu_initialized) or self.cal["init"])

    def activate_points(self, flagged_old):     # FS/FullSystem.cpp:376-533
        sc = self.sc
        self.currentMinActDist = next_min_act_dist(self.currentMinActDist, self.n_points(), sc.desired_points)
        n = self.n()
        ids = self.window_ids()
        newest = n - 1
        poses = [self.kf_pose(i) for i in range(n)]
        affs = [self.kf_aff(i) for i in range(n)]
        KRKi1, Kt1 = level1_to_newest(self.K(), poses, newest)
        cand, cand_host, cand_type, src = [], [], [], []
        for i, fid in enumerate(ids[:-1]):
            m = len(self.imm[fid])
            cand.append(self.imm[fid])
            cand_host.append(np.full(m, i, np.int32))
            cand_type.append(self.imm_type[fid])
            src += [(fid, j) for j in range(m)]
        cand = np.concatenate(cand) if cand else np.zeros(0, dtype=IMMATURE_DTYPE)
        cand_host = np.concatenate(cand_host) if cand_host else np.zeros(0, np.int32)
        cand_type = np.concatenate(cand_type) if cand_type else np.zeros(0, np.float32)
        host_flagged = np.zeros(n, np.uint8)
        host_flagged[:len(flagged_old)] = np.asarray(flagged_old, dtype=np.uint8)
        dec = self.select(sc.w // 2, sc.h // 2, newest, KRKi1, Kt1, self.active_points(), self.currentMinActDist, 3.0, cand, cand_host,
                          cand_type, host_flagged)
        todo = np.flatnonzero(dec == 1)
        act = self.activate(cand[todo], cand_host[todo], poses, affs)
        gone = set(np.flatnonzero(dec == -1).tolist())
        new_pts, masks, keys = [], [], []
        for j, a in zip(todo, act):
            if a["status"] == ACT_ACTIVATED:
                p = np.zeros(1, dtype=synth.POINT_DTYPE)[0]
                r = cand[j]
                p["u"], p["v"] = r["u"], r["v"]
                p["idepth_scaled"] = p["idepth_zero_scaled"] = a["idepth"]
                p["color"], p["weights"] = r["color"], r["weights"]
                p["host"] = cand_host[j]
                new_pts.append(p)
                masks.append(int(a["inMask"]))
                keys.append((ids[cand_host[j]], float(r["u"]), float(r["v"])))
                gone.add(int(j))
            elif a["status"] == ACT_DELETE or r_status(cand[j]) == IPS_OOB:
                gone.add(int(j))
        if new_pts:
            self.add_activated(np.array(new_pts, dtype=synth.POINT_DTYPE), np.array(masks, dtype=np.uint32))
        # removal with the reference's swap-from-the-back compaction (:519-531) per host
        by_host = {}
        for j in gone:
            fid, loc = src[j]
            by_host.setdefault(fid, set()).add(loc)
        for fid, locs in by_host.items():
            recs, typ = list(self.imm[fid]), list(self.imm_type[fid])
            alive = [j not in locs for j in range(len(recs))]
            i = 0
            while i < len(recs):
                if not alive[i]:
                    recs[i], typ[i], alive[i] = recs[-1], typ[-1], alive[-1]
                    recs.pop(); typ.pop(); alive.pop()
                    continue
                i += 1
            self.imm[fid] = np.array(recs, dtype=IMMATURE_DTYPE) if recs else np.zeros(0, dtype=IMMATURE_DTYPE)
            self.imm_type[fid] = np.array(typ, dtype=np.float32)
        return keys, len(gone) - len(keys)


def r_status(rec):
    return int(rec["lastTraceStatus"])


# ------------------------------------------------------------------------------------------------
# device chain: facade (C++) + HIP kernels
# ------------------------------------------------------------------------------------------------
class DeviceChain(Chain):
    def __init__(self, sc):
        super().__init__(sc)
        from sos_slam_amd import host, lib
        self.host, self.lib = host, lib
        self.sysm = host.System(sc.params)
        self.sysm.set_calib(sc.K)
        self.ctx = self.sysm.context()
        self.und = lib.Undistorter(self.ctx, lib.camera_parse(sc.cam_txt))
        self.sel = lib.PixelSelector(self.ctx, self.pprm, sc.pattern)
        self.trk = None
        if os.environ.get("SOS_TEST_RESIDENT") == "1":   # tests/test_gpu_variants.py: the same chain with the device-resident loop (k_gn_solve)
            self.sysm.set_resident(True)

    def close(self):
        if getattr(self, "iset", None) is not None:
            self.iset.close()
        self.sysm.close()

    def trace_new_coarse(self, h_new, ids, poses, affs, K4, c2w, aff, keyframe):
        """With frames between the keyframes the immature lists live in a device-resident set (sos_immset): a non-keyframe costs one
        launch and no copies; the records come back at the keyframe, where activatePointsMT / makeNewTraces / marginalisation edit them
        (and are put again before the next trace)."""
        if self.sc.kf_every == 1:
            return super().trace_new_coarse(h_new, ids, poses, affs, K4, c2w, aff, keyframe)
        if getattr(self, "iset", None) is None:
            self.iset, self.iset_keys = self.lib.ImmatureSet(self.ctx), {}
        for fid in list(self.iset_keys):          # keyframes that left the window
            if fid not in ids:
                self.iset.put(fid, self.imm.get(fid, np.zeros(0, dtype=IMMATURE_DTYPE))[:0])
                del self.iset_keys[fid]
        for fid in ids:                           # lists edited on the host since the last trace (every list after a keyframe)
            if self.iset_keys.get(fid) != "resident":
                self.iset.put(fid, self.imm[fid])
                self.iset_keys[fid] = "resident"
        tabs = [host_to_frame(K4, poses[i], c2w, affs[i], aff) for i in range(len(ids))]
        self.iset.trace(self.tprm, h_new, ids, np.stack([t[0].reshape(-1) for t in tabs]), np.stack([t[1] for t in tabs]),
                        np.stack([t[2] for t in tabs]))
        if keyframe:
            for fid in ids:
                self.imm[fid] = self.iset.get(fid)
                self.iset_keys[fid] = "host"      # about to be edited

    def n(self): return self.sysm.counts()[0]
    def n_points(self): return self.sysm.counts()[1]
    def K(self): return self.sysm.calib_value_scaled()
    def window_ids(self): return self.sysm.frame_ids()["frameID"].tolist()
    def kf_pose(self, i): return self.sysm.frame(i)["camToWorld"]

    def kf_aff(self, i):
        st = self.sysm.frame(i)["state"]
        return (st[6] * SCALE_A, st[7] * SCALE_B)

    def front_end(self, raw):
        slot = self.sysm.alloc_slot()
        self.und.frame(raw, 0.0, slot, want_image=False)
        return slot

    def irradiance(self, slot):
        return self.ctx.download_level(slot, 0)[0][..., 0].copy()

    def release(self, slot):
        pass   # the facade released the keyframe's slot with the frame (marginalizeFrame)

    def init_window(self, hs, poses, affs, pts, res):
        for i, (slot, T, aff) in enumerate(zip(hs, poses, affs)):
            f = np.zeros(1, dtype=synth.FRAME_INIT_DTYPE)[0]
            f["camToWorld"] = T
            f["state"][6], f["state"][7] = aff[0] / SCALE_A, aff[1] / SCALE_B
            f["ab_exposure"], f["frameID"], f["frameEnergyTH"] = 1.0, i, 8 * 8 * 8
            self.sysm.add_frame_from_slot(f, slot)
        self.sysm.add_points(pts)
        rr = np.zeros(len(res), dtype=synth.RESID_DTYPE)
        for k, (pi, t) in enumerate(res):
            rr[k] = (pi, pts[pi]["host"], t, synth.RF_ISNEW, synth.RES_IN, 0.0)
        self.sysm.add_residuals(rr)

    def tracker_set_ref(self):
        if self.trk is None:
            self.trk = self.host.HostTracker(self.sysm)
        self.trk.set_ref()

    def optimize_scale_kf(self, slot, ref_scale):
        K1 = np.asarray(self.K(), dtype=np.float32)
        return self.trk.optimize_scale_kf(slot, self.sc.stereo_tfm, K1, ref_scale, self.ctx.levels - 1, self.sc.scale_opt_thres, self.scale_state)

    def release_plain(self, slot):
        self.sysm.release_image(slot)

    def track(self, slot, T_init, ref_aff):
        levels = self.ctx.levels
        ok, T, aff, lres, flow = self.trk.track(slot, 1.0, T_init, ref_aff, levels - 1)
        return ok, T, aff, lres, flow

    def trace(self, slot, recs, KRKi, Kt, aff):
        return self.ctx.immature_trace(self.tprm, slot, recs, KRKi, Kt, aff)

    def flag_frames(self, num_immature):
        return self.sysm.flag_frames_for_marginalization(num_immature)

    def add_keyframe(self, slot, c2w, aff, frameID):
        f = np.zeros(1, dtype=synth.FRAME_INIT_DTYPE)[0]
        f["camToWorld"] = c2w
        f["state"][6], f["state"][7] = aff[0] / SCALE_A, aff[1] / SCALE_B
        f["ab_exposure"], f["frameID"], f["frameEnergyTH"] = 1.0, frameID, 8 * 8 * 8
        self.sysm.add_frame_from_slot(f, slot)

    def add_old_point_residuals(self):
        return self.sysm.add_new_frame_residuals()

    def active_points(self):
        hf, u, v, hi = self.sysm.point_keys()
        p = self.sysm.points()
        out = np.zeros(len(u), dtype=[("u", "f4"), ("v", "f4"), ("idepth_scaled", "f4"), ("host", "i4")])
        out["u"], out["v"], out["idepth_scaled"], out["host"] = u, v, p["idepth"], hi
        return out

    def select(self, w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged):
        return self.host.activate_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged)[0]

    def activate(self, recs, host_idx, poses, affs):
        n = self.n()
        calib = Calib.from_K(self.K())
        slots = [self.sysm.frame_slot(i) for i in range(n)]
        return self.ctx.immature_activate(self.aprm, calib, slots, pair_tfms(poses, affs), recs, host_idx)

    def add_activated(self, pts, masks):
        self.sysm.add_activated_points(pts, masks)

    def optimize(self, its):
        if self.vio and self.cal["init"]:
            self._push_imu()
        r = self.sysm.optimize(its)
        if self.vio and self.cal["init"]:
            self._pull_imu()
        return r

    # ---- visual-inertial: the facade's front-end functions and its own expanded prior (sosf_set_imu with NULL priors)
    def _fe(self):
        if not hasattr(self, "_imu_fe"):
            self._imu_fe = self.host.ImuFrontEnd()
        return self._imu_fe

    def _shell(self, fid):
        from sos_slam_amd.records import ImuShell
        sh = ImuShell()
        sh.timestamp = self.shells[fid]["ts"]
        sh.camToWorld[:] = list(self.shells[fid]["c2w"])
        sh.velInWorld[:] = list(self.shells[fid]["vel"])
        return sh

    def _push_imu(self):
        recs, self._imu_keep = self.vio_records(self.window_ids())
        self._cal_struct = self.vio_calib()
        self.sysm.set_imu(self.sc.imu_settings, self._cal_struct, recs)

    def _pull_imu(self):
        arr = self.sysm._imu[2]
        for i, fid in enumerate(self.window_ids()):
            self.imu_state[fid] = np.array(arr[i].state_imu[:])
        self.vio_take_calib(self._cal_struct)

    def vio_propagate(self, fid, last_fid, last_bias):
        recs, keep = self.vio_records([fid])
        sh, cal = self._shell(fid), self.vio_calib()
        self._fe().propagate_state(self.sc.imu_settings, cal, recs[0], sh, self._shell(last_fid), last_bias)
        self.imu_state[fid], self.imu_zero[fid] = np.array(recs[0].state_imu[:]), np.array(recs[0].state_imu_zero[:])
        self.shells[fid]["vel"] = np.array(sh.velInWorld[:])

    def vio_initialize(self, ids):
        recs, keep = self.vio_records(ids)
        for i, r in enumerate(recs):
            r.camToWorld[:] = list(self.kf_pose(i))          # PRE_camToWorld
        cal = self.vio_calib()
        ok, fo, so = self._fe().initialize(self.sc.imu_settings, cal, recs, [self._shell(f) for f in ids])
        for i, fid in enumerate(ids):
            self.imu_state[fid], self.imu_zero[fid] = np.array(fo[i].state_imu[:]), np.array(fo[i].state_imu_zero[:])
            self.shells[fid]["vel"] = np.array(so[i].velInWorld[:])
        self.vio_take_calib(cal)
        return ok

    def vio_update_vel(self, fid, last_fid):
        recs, keep = self.vio_records([fid])
        sh = self._shell(fid)
        self._fe().update_vel(recs[0], sh, self._shell(last_fid))
        self.shells[fid]["vel"] = np.array(sh.velInWorld[:])

    def vio_try_trap(self):
        cal = self.vio_calib()
        self.scale_queue, self.scale_qi = self._fe().try_trap_scale(cal, self.scale_queue, self.scale_qi, 1e-4)
        self.vio_take_calib(cal)

    def prior_imu(self):
        if not self.cal["init"]:
            return None, None
        return self.sysm.imu_prior()

    def remove_outliers(self):
        return self.sysm.remove_outliers()

    def flag_points_for_removal(self):
        return self.sysm.flag_points_for_removal()

    def pixel_select(self, slot, density):
        self.sel.make_maps(slot, density, want_map=False)
        return self.sel.list(pattern_padding=2)

    def immature_init(self, slot, u, v):
        return self.ctx.immature_init(self.tprm, slot, u, v)

    def marginalize_flagged(self):
        if self.vio and self.cal["init"]:
            self._push_imu()        # the IMU form of marginalizeFrame reads the records of the window as it is now
        win = self.window_ids()
        ids, poses = self.sysm.marginalize_flagged_frames()
        for i in ids:      # the facade merged the sample lists inside the call; the chain's own storage follows
            win = self.imu_hand_over(win, int(i))
        return [(int(i), p.copy()) for i, p in zip(ids, poses)]

    def prior(self):
        return self.sysm.get_prior()

    def residual_set(self):
        hf, u, v, _ = self.sysm.point_keys()
        ids = self.sysm.point_ids()
        key = {int(i): (int(a), float(b), float(c)) for i, a, b, c in zip(ids, hf, u, v)}
        pi, tf = self.sysm.residual_ids()
        return {key[int(p)] + (int(t),) for p, t in zip(pi, tf)}

    def point_set(self):
        hf, u, v, _ = self.sysm.point_keys()
        return {(int(a), float(b), float(c)) for a, b, c in zip(hf, u, v)}


class CppDeviceChain(DeviceChain):
    """The device chain with the frame-rate loop in C++ (sosf_sequence, csrc/host/sos_sequence.cpp): this class only feeds frames and
    fills the KeyframeLog from the sequence's results and snapshots.  Written in round 3 without GPU access, run under tests/emu in round
    4; device_chain() picks it for the visual chains."""

    def bootstrap(self):
        from sos_slam_amd.records import SequenceParams
        sc = self.sc
        hs = [self.front_end(sc.raw[i]) for i in range(sc.n0)]
        images = [self.irradiance(h) for h in hs]
        pts, res = sc.bootstrap_points(images)
        self.init_window(hs, [sc.poses[i] for i in range(sc.n0)], [sc.aff_true[i] for i in range(sc.n0)], pts, res)
        prm = SequenceParams.default(desired_points=sc.desired_points, immature_density=sc.immature_density, kf_every=sc.kf_every)
        self.seq = self.host.Sequence(self.sysm, prm, sc.pattern)
        self.seq.set_snapshots(True)
        if self.vio:
            self.seq.enable_imu(sc.imu_settings, [sc.ts[i] for i in range(sc.n0)], [sc.imu[i] for i in range(sc.n0)])
            self.n_kf_total = sc.n0
        sr = -1
        if self.stereo:
            self.seq.enable_stereo(sc.stereo_tfm, sc.scale_opt_thres)
            sr = self.front_end(sc.raw_right[sc.n0 - 1])
        rmse, its = self.seq.bootstrap(stereo_slot=sr)
        if self.stereo:
            self.release_plain(sr)
            self._log_scale(sc.n0 - 1, None)
        for i in range(sc.n0):
            self.handles[i] = hs[i]
            self.imm[i], self.imm_type[i] = self.seq.immature(i)
        self.last_rel = None
        return rmse, its

    def _log_scale(self, k, out):
        # (new_scale, error) of the bootstrap call are not returned by sosf_sequence_bootstrap_ex: the calibration says whether it was accepted
        c = self.seq.imu_calib()
        ns = float(out.newScale) if out is not None else (200.0 * c.scale if c.scale != 1.0 / 200.0 else -1.0)
        err = float(out.scaleError) if out is not None else float("nan")
        self.scale_state = self.seq.scale_state()
        self.scale_log.append((k, ns, err, list(self.scale_state)))
        self.cal["scale"], self.cal["scale_zero"] = c.scale, c.scale_zero

    def close(self):
        if getattr(self, "seq", None) is not None:
            self.seq.close()
            self.seq = None
        super().close()

    def step(self):
        sc = self.sc
        nonkf = []
        while True:
            if self.next_frame >= sc.n_frames:
                return None
            k = self.next_frame
            self.next_frame += 1
            slot = self.front_end(sc.raw[k])
            ids = self.window_ids()
            T_init = None
            if (sc.kf_every == 1 and self.last_rel is None) or (sc.kf_every > 1 and not self.track_hist):
                T_init = se3_mul(se3_inv(sc.poses[k]), sc.poses[k - 1] if sc.kf_every == 1 else sc.poses[ids[-1]])
                T_init = se3_mul(synth.se3_exp12(np.array([0.002, -0.001, 0.001, 0.001, -0.001, 0.0005])), T_init)
            sr = self.front_end(sc.raw_right[k]) if self.stereo else -1
            kw = dict(timestamp=float(sc.ts[k]), imu=sc.imu[k]) if self.vio else {}
            out = self.seq.add_active_frame(slot, k, T_init, stereo_slot=sr, **kw)
            if self.stereo:
                self.release_plain(sr)
            assert out.trackingOk == 1, "tracking lost"
            T = np.array(out.refToNew[:])
            self.last_rel = T.copy()
            self.track_hist = (self.track_hist + [np.array(out.camToWorld[:])])[-2:]
            if out.isKeyframe:
                break
            nonkf.append((k, T.copy()))
            self.release_plain(slot)
        self.handles[k] = slot
        sq = self.seq
        flagged = [int(f) for f in sq.snapshot(0)[:, 0]]
        activated = [(int(a), float(np.float32(b)), float(np.float32(c))) for a, b, c in sq.snapshot(1)]
        residual_set = {(int(a), float(np.float32(b)), float(np.float32(c)), int(d)) for a, b, c, d in sq.snapshot(2)}
        point_set = {(int(a), float(np.float32(b)), float(np.float32(c))) for a, b, c in sq.snapshot(3)}
        ids2 = self.window_ids()
        marg = [(int(out.margFrameIDs[i]), np.array(out.margCamToWorld[12 * i:12 * i + 12])) for i in range(out.nMargFrames)]
        # window as optimize() left it = the window now plus the keyframes that left at the end of makeKeyFrame, in frameID order
        win_opt = sorted(set(ids2) | {f for f, _ in marg})
        poses_now = {fid: self.kf_pose(i).copy() for i, fid in enumerate(ids2)}
        window_poses = {fid: (poses_now[fid] if fid in poses_now else dict(marg)[fid]) for fid in win_opt}
        for fid, _ in marg:
            self.handles.pop(fid, None)
            self.imm.pop(fid, None)
            self.imm_type.pop(fid, None)
        for fid in ids2:
            self.imm[fid], self.imm_type[fid] = sq.immature(fid)
        HM, bM = self.prior()
        vio = None
        if self.vio:
            self.n_kf_total += 1
            c = sq.imu_calib()
            self.cal.update(scale=c.scale, scale_zero=c.scale_zero, trapped=int(c.scale_trapped), init=int(c.imu_initialized))
            st = {f: sq.imu(f) for f in ids2}
            Hi, bi = self.prior_imu()
            vio = dict(scale=c.scale, scale_zero=c.scale_zero, trapped=int(c.scale_trapped), init=int(c.imu_initialized),
                       states={f: st[f][0].copy() for f in ids2}, vel={f: st[f][2].copy() for f in ids2}, HMi=Hi, bMi=bi)
        if self.stereo:
            self._log_scale(k, out)
        self.logs.append(KeyframeLog(k, T, np.array(out.aff[:]), flagged, activated, int(out.nDeletedImmature), int(out.nPointsBeforeOpt),
                                     float(out.rmse), int(out.iterations), win_opt, window_poses, residual_set, int(out.nOutliersRemoved),
                                     int(out.nMargPoints), int(out.nDroppedPoints), point_set, int(out.nNewImmature), marg, HM, bM, nonkf, vio))
        return self.logs[-1]


def device_chain(sc):
    """the device chain of the rolling tests: the frame-rate loop in C++ (sosf_sequence) -- what a SOS-SLAM maintainer would run -- is what
    the oracle chain is compared with, visual and visual-inertial.  SOS_ROLLING_CPP=0 puts the Python loop over the facade's stages in its
    place (tests/test_gpu_variants.py keeps one visual and one visual-inertial chain on it; tests/test_gpu_sequence_driver.py compares the
    two loops directly).  Both loops and the oracle chain form KRKi / Kt with the same float32 statement (immature_helpers.mm3_f32): with
    NumPy's BLAS products on one side the traces of one keyframe in four sat a few ulps apart, enough to move ten activations at the
    first keyframe of the stereo-inertial chain and, through one point marginalised on one side only, 11 % of a prior entry."""
    import os
    use_cpp = os.environ.get("SOS_ROLLING_CPP") != "0"
    return CppDeviceChain(sc) if use_cpp else DeviceChain(sc)


# ------------------------------------------------------------------------------------------------
# oracle chain: its own graph (second statement of the host logic) + the C restatement for the arithmetic
# ------------------------------------------------------------------------------------------------
class ORes:
    __slots__ = ("point", "target", "state_state", "state_energy", "center")

    def __init__(self, point, target):
        self.point, self.target = point, target
        self.state_state, self.state_energy = synth.RES_IN, 0.0
        self.center = np.zeros(3, np.float32)


class OPoint:
    def __init__(self, host, rec_u, rec_v, idepth, color, weights):
        self.host = host
        self.u, self.v = np.float32(rec_u), np.float32(rec_v)
        self.idepth = self.idepth_zero = np.float32(idepth)
        self.color, self.weights = np.array(color, np.float32), np.array(weights, np.float32)
        self.residuals = []          # EFPoint::residualsAll order (insert at the back, drop = swap with the back)
        self.last = [[None, synth.RES_OOB], [None, synth.RES_OOB]]
        self.numGoodResiduals = 0
        self.maxRelBaseline = np.float32(0)
        self.idepth_hessian = np.float32(0)
        self.HdiF = np.float32(0)
        self.priorF = np.float32(0)

    def key(self):
        return (self.host.frameID, float(self.u), float(self.v))

    def drop_residual(self, r):
        i = self.residuals.index(r)
        self.residuals[i] = self.residuals[-1]
        self.residuals.pop()


class OFrame:
    def __init__(self, frameID, handle, c2w, aff):
        self.frameID, self.handle = frameID, handle
        self.evalPT = np.array(c2w, dtype=np.float64)
        self.state = np.zeros(10)
        self.state[6], self.state[7] = aff[0] / SCALE_A, aff[1] / SCALE_B
        self.state_zero = self.state.copy()
        self.pre = self.evalPT.copy()      # PRE_camToWorld
        self.frameEnergyTH = np.float32(8 * 8 * 8)
        self.flagged = False
        self.points = []
        self.n_marg = self.n_out = 0


class OracleChain(Chain):
    def __init__(self, sc, truth=False):
        super().__init__(sc)
        self.truth = truth
        self.und = orc.Undistorter(sc.cam_txt)
        self.sel = orc.PixelSelector(self.pprm, sc.pattern, sc.w, sc.h)
        self.frames = []
        # CalibHessian(): setValueScaled then value_zero = value (FS/HessianBlocks.h:453-475): (1.0f / SCALE_F) * value_scaled
        self.calib_value = np.array([float(np.float32(1.0) / np.float32(50.0)) * v for v in sc.K])
        self.calib_value_zero = self.calib_value.copy()
        self.HM, self.bM = np.zeros((4, 4)), np.zeros(4)
        self.HMi = self.bMi = None      # visual-inertial: the prior in the expanded dimension, from the first IMU solve on
        self.trk = None
        self.sel_slot = None

    def close(self):
        pass

    # ---- queries
    def n(self): return len(self.frames)
    def n_points(self): return sum(len(f.points) for f in self.frames)
    def K(self): return np.array([50.0 * self.calib_value[0], 50.0 * self.calib_value[1], 50.0 * self.calib_value[2], 50.0 * self.calib_value[3]])
    def window_ids(self): return [f.frameID for f in self.frames]
    def kf_pose(self, i): return self.frames[i].pre
    def kf_aff(self, i): return (self.frames[i].state[6] * SCALE_A, self.frames[i].state[7] * SCALE_B)

    def front_end(self, raw):
        img = self.und.frame(raw, 0.0)
        dI, absg = orc.make_images(img)
        return dict(dI=dI, absg=absg)

    def irradiance(self, h):
        return h["dI"][0][..., 0].copy()

    def release(self, h):
        pass

    def _grow_prior(self):
        odim = self.HM.shape[0]
        HM, bM = np.zeros((odim + 8, odim + 8)), np.zeros(odim + 8)
        HM[:odim, :odim], bM[:odim] = self.HM, self.bM
        self.HM, self.bM = HM, bM

    # ---- visual-inertial: NumPy front-end (oracle/imu_frontend.py), orc_imu_* for the assembly, the expanded prior kept here
    def _Sd(self):
        S = self.sc.imu_settings
        return dict(gravity=np.array(S.gravity[:]), rot_imu_cam=np.array(S.rot_imu_cam[:]))

    def vio_propagate(self, fid, last_fid, last_bias):
        from oracle import imu_frontend as fe
        a, b = self.shells[fid], self.shells[last_fid]
        sc_, vel = fe.propagate_imu_state(self._Sd(), self.cal["scale"], a["ts"], self.imu[fid], b["ts"], b["c2w"][:9].reshape(3, 3), b["vel"],
                                          last_bias, fe.scaled_of(self.imu_state[fid]))
        self.imu_state[fid] = fe.state_of(sc_)
        self.imu_zero[fid] = self.imu_state[fid].copy()
        a["vel"] = vel

    def vio_initialize(self, ids):
        from oracle import imu_frontend as fe
        ts = [self.shells[f]["ts"] for f in ids]
        c2w = [self.shells[f]["c2w"] for f in ids]
        r = fe.initialize_imu(self._Sd(), self.cal["scale"], bool(self.sc.imu_settings.enable_scale_opt), ts, c2w, self.kf_pose(4)[:9],
                              [self.imu[f] for f in ids], [fe.scaled_of(self.imu_state[f]) for f in ids])
        if not self.sc.imu_settings.enable_scale_opt:        # setScaleScaledZero
            self.cal["scale"] = self.cal["scale_zero"] = fe.SCALE_SCALE_INVERSE * r["scale_scaled"]
        for i, f in enumerate(ids):
            self.shells[f]["vel"] = r["vel"][i].copy()
            if r["ok"]:
                self.imu_state[f] = fe.state_of(r["scaled"][i])
                self.imu_zero[f] = self.imu_state[f].copy()
        return r["ok"]

    def vio_update_vel(self, fid, last_fid):
        from oracle import imu_frontend as fe
        a, b = self.shells[fid], self.shells[last_fid]
        a["vel"] = fe.update_vel(fe.scaled_of(self.imu_state[fid]), a["ts"], a["c2w"][9:], b["ts"], b["c2w"][9:])

    def vio_try_trap(self):
        from oracle import imu_frontend as fe
        zero, trapped, self.scale_queue, self.scale_qi = fe.try_trap_scale(self.cal["scale"], self.scale_queue, self.scale_qi, 1e-4)
        self.cal["scale_zero"] = zero
        self.cal["trapped"] = int(trapped)

    def prior_imu(self):
        if self.HMi is None:
            return None, None
        return self.HMi.copy(), self.bMi.copy()

    def _window_records(self):
        recs, keep = self.vio_records(self.window_ids())
        for f, r in zip(self.frames, recs):
            r.camToWorld[:] = list(f.pre)
            r.evalPT_R[:] = list(f.evalPT[:9])
        return recs, keep

    def _stitched_delta(self):
        n = len(self.frames)
        d = np.zeros(4 + 8 * n)
        d[:4] = (self.calib_value - self.calib_value_zero).astype(np.float32)
        for i, f in enumerate(self.frames):
            d[4 + 8 * i:12 + 8 * i] = f.state[:8] - f.state_zero[:8]
        return d

    def init_window(self, hs, poses, affs, pts, res):
        for i, (h, T, aff) in enumerate(zip(hs, poses, affs)):
            self.frames.append(OFrame(i, h, T, aff))
            self._grow_prior()
        plist = []
        for p in pts:
            op = OPoint(self.frames[int(p["host"])], p["u"], p["v"], p["idepth_scaled"], p["color"], p["weights"])
            self.frames[int(p["host"])].points.append(op)
            plist.append(op)
        for pi, t in res:                      # sosf_add_residuals semantics: lastResiduals shifted per added residual
            op = plist[pi]
            r = ORes(op, self.frames[t])
            op.residuals.append(r)
            op.last[1] = op.last[0]
            op.last[0] = [r, synth.RES_IN]

    # ---- tracker
    def tracker_set_ref(self):   # CoarseTracker::setCoarseTrackingRef, FS/CoarseTracker.cpp:56-79, 232-242
        last = self.frames[-1]
        u, v, idp, hdi = [], [], [], []
        for f in self.frames:
            for p in f.points:
                r, st = p.last[0]
                if r is not None and st == synth.RES_IN and r.target is last:
                    u.append(r.center[0]); v.append(r.center[1]); idp.append(r.center[2]); hdi.append(p.HdiF)
        self.trk = orc.OracleTracker(self.sc.params, self.sc.w, self.sc.h)
        self.trk.set_truth_mode(self.truth)
        self.trk.set_ref(Calib.from_K(self.K()), last.handle["dI"], np.array(u, np.float32), np.array(v, np.float32), np.array(idp, np.float32),
                         np.array(hdi, np.float32))
        self.trk_ref_aff = np.array(self.kf_aff(len(self.frames) - 1))

    def optimize_scale_kf(self, h, ref_scale):
        K1 = np.asarray(self.K(), dtype=np.float32)
        return orc.optimize_scale_kf(self.trk, h["dI"], self.sc.stereo_tfm, K1, ref_scale, self.trk.levels - 1, self.sc.scale_opt_thres,
                                     self.scale_state)

    def release_plain(self, h):
        pass

    def track(self, h, T_init, ref_aff):
        levels = self.trk.levels
        ok, T, aff, lres, flow = self.trk.track(h["dI"], 1.0, 1.0, self.trk_ref_aff, T_init, ref_aff, levels - 1)
        return bool(ok), T, aff, lres, flow

    def trace(self, h, recs, KRKi, Kt, aff):
        return orc.immature_trace(self.tprm, h["dI"][0], recs, KRKi, Kt, aff)

    # ---- FullSystem::flagFramesForMarginalization, FS/FullSystemMarginalize.cpp:53-133
    def flag_frames(self, num_immature):
        fr = self.frames
        flagged = 0
        back = fr[-1]
        for f, ni in zip(fr, num_immature):
            n_in = len(f.points) + ni
            n_out = f.n_marg + f.n_out
            a = np.exp(f.state[6] * SCALE_A - back.state[6] * SCALE_A)      # fromToVecExposure(back -> f)[0], exposures 1
            if (n_in < MIN_POINTS_REMAINING * (n_in + n_out) or abs(np.log(np.float32(a))) > MAX_LOG_AFF_FAC) and len(fr) - flagged > MIN_FRAMES:
                f.flagged = True
                flagged += 1
        if len(fr) - flagged >= MAX_FRAMES:
            smallest, pick = 1.0, None
            latest = fr[-1]
            for f in fr:
                if f.frameID > latest.frameID - MIN_FRAME_AGE or f.frameID == 0:
                    continue
                score = 0.0
                for t in fr:
                    if t.frameID > latest.frameID - MIN_FRAME_AGE + 1 or t is f:
                        continue
                    score += 1 / (1e-5 + self._distance(f, t))
                score *= -np.sqrt(np.float32(self._distance(f, fr[-1])))
                if score < smallest:
                    smallest, pick = score, f
            if pick is not None:
                pick.flagged = True
        return np.array([f.flagged for f in fr])

    @staticmethod
    def _distance(host, target):     # FrameFramePrecalc::distanceLL (float)
        T = se3_mul(se3_inv(target.pre), host.pre)
        return float(np.float32(np.linalg.norm(T[9:])))

    def add_keyframe(self, h, c2w, aff, frameID):
        self.frames.append(OFrame(frameID, h, c2w, aff))
        self._grow_prior()
        if self.HMi is not None:     # step = 29, OB/EnergyFunctional.cpp:666-677
            od = self.HMi.shape[0]
            Hn, bn = np.zeros((od + 29, od + 29)), np.zeros(od + 29)
            Hn[:od, :od], bn[:od] = self.HMi, self.bMi
            self.HMi, self.bMi = Hn, bn

    def add_old_point_residuals(self):    # FS/FullSystem.cpp:818-832
        new = self.frames[-1]
        c = 0
        for f in self.frames[:-1]:
            for p in f.points:
                r = ORes(p, new)
                p.residuals.append(r)
                p.last[1] = p.last[0]
                p.last[0] = [r, synth.RES_IN]
                c += 1
        return c

    def active_points(self):
        n = self.n_points()
        out = np.zeros(n, dtype=[("u", "f4"), ("v", "f4"), ("idepth_scaled", "f4"), ("host", "i4")])
        k = 0
        for i, f in enumerate(self.frames):
            for p in f.points:
                out[k] = (p.u, p.v, p.idepth, i)
                k += 1
        return out

    def select(self, w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged):
        return orc.activate_select(w1, h1, newest, KRKi, Kt, act, min_dist, min_quality, cand, cand_host, cand_type, flagged)[0]

    def activate(self, recs, host_idx, poses, affs):
        calib = Calib.from_K(self.K())
        return orc.immature_activate(self.aprm, calib, [f.handle["dI"][0] for f in self.frames], pair_tfms(poses, affs), recs, host_idx)

    def add_activated(self, pts, masks):       # FS/FullSystemOptPoint.cpp:151-185
        nf = len(self.frames)
        newest, second = self.frames[-1], (self.frames[-2] if nf >= 2 else None)
        for p, m in zip(pts, masks):
            host = self.frames[int(p["host"])]
            op = OPoint(host, p["u"], p["v"], p["idepth_scaled"], p["color"], p["weights"])
            for t in range(nf):
                if not ((int(m) >> t) & 1) or self.frames[t] is host:
                    continue
                r = ORes(op, self.frames[t])
                op.residuals.append(r)
                if r.target is newest:
                    op.last[0] = [r, synth.RES_IN]
                elif r.target is second:
                    op.last[1] = [r, synth.RES_IN]
            host.points.append(op)

    # ---- optimize(): pack the graph, run the C restatement's loop, read everything back
    def _pack(self):
        n = len(self.frames)
        idx = {id(f): i for i, f in enumerate(self.frames)}
        plist = [p for f in self.frames for p in f.points]
        pts = np.zeros(len(plist), dtype=synth.POINT_DTYPE)
        rlist = []
        for k, p in enumerate(plist):
            pts[k]["u"], pts[k]["v"] = p.u, p.v
            pts[k]["idepth_scaled"], pts[k]["idepth_zero_scaled"] = p.idepth, p.idepth_zero
            pts[k]["color"], pts[k]["weights"] = p.color, p.weights
            pts[k]["priorF"] = p.priorF
            pts[k]["deltaF"] = np.float32(p.idepth - p.idepth_zero)
            pts[k]["host"] = idx[id(p.host)]
            for r in p.residuals:
                rlist.append((k, r))
        res = np.zeros(len(rlist), dtype=synth.RESID_DTYPE)
        for j, (k, r) in enumerate(rlist):
            # optimize() starts with resetOOB on every residual; isActive of the last apply is what the graph carries
            res[j] = (k, pts[k]["host"], idx[id(r.target)], synth.RF_ISNEW | synth.RF_ACTIVE, r.state_state, r.state_energy)
        ow = orc.OracleWindow(self.sc.params, n, pts, res)
        for i, f in enumerate(self.frames):
            ow.set_image(i, f.handle["dI"][0])
        fe = np.zeros(n, dtype=FRAME_INIT_EX_DTYPE)
        for i, f in enumerate(self.frames):
            fe[i]["camToWorld"], fe[i]["state"], fe[i]["state_zero"] = f.evalPT, f.state, f.state_zero
            fe[i]["ab_exposure"], fe[i]["frameID"], fe[i]["frameEnergyTH"] = 1.0, f.frameID, f.frameEnergyTH
        ow.host_init_ex(fe, self.calib_value, self.calib_value_zero, self.HM, self.bM)
        ow.set_truth_mode(self.truth)
        return ow, plist, rlist

    def _unpack_frames(self, ow):
        for i, f in enumerate(self.frames):
            fr = ow.frame(i)
            f.state, f.state_zero, f.pre = fr["state"].copy(), fr["state_zero"].copy(), fr["camToWorld"].copy()
            f.evalPT = ow.evalpt(i)
            f.frameEnergyTH = np.float32(fr["frameEnergyTH"])
        self.calib_value, _ = ow.calib_value()

    def optimize(self, its):
        ow, plist, rlist = self._pack()
        imu_on = self.vio and self.cal["init"]
        if imu_on:
            if self.HMi is None:     # the reference keeps HM expanded from the start; nothing was marginalised before this point
                self.HMi, self.bMi = orc.imu().expand(len(self.frames), self.HM, self.bM)
            recs, keep = self._window_records()
            cal = self.vio_calib()
            ow.set_imu(self.sc.imu_settings, cal, recs, self.HMi, self.bMi)
        rmse, it = ow.optimize(its)
        if imu_on:
            _, st = ow.imu_state()
            for f, x in zip(self.frames, st):
                self.imu_state[f.frameID] = x.copy()
            self.vio_take_calib(cal)
        self._unpack_frames(ow)
        po, ro, cen = ow.pts(), ow.res(), ow.center()
        ngr, mrb = ow.num_good_residuals(), ow.point_field("maxRelBaseline")
        idh, hdi = ow.point_field("idepth_hessian"), ow.point_field("HdiF")
        has_act = np.zeros(len(plist), bool)
        for k, p in enumerate(plist):
            p.idepth, p.idepth_zero = po[k]["idepth_scaled"], po[k]["idepth_zero_scaled"]
            p.numGoodResiduals += int(ngr[k])
            p.maxRelBaseline = max(p.maxRelBaseline, mrb[k])
            p.idepth_hessian, p.HdiF = idh[k], hdi[k]
        # linearizeAll(true): states, lastResiduals bookkeeping, removal of the residuals that are not active (:148-179)
        for j, (k, r) in enumerate(rlist):
            r.state_state, r.state_energy = int(ro[j]["state_state"]), float(ro[j]["state_energy"])
            r.center = cen[j].copy()
            p = r.point
            if p.last[0][0] is r:
                p.last[0][1] = r.state_state
            elif p.last[1][0] is r:
                p.last[1][1] = r.state_state
        for j, (k, r) in enumerate(rlist):
            if ro[j]["flags"] & 0x100:
                p = r.point
                if p.last[0][0] is r:
                    p.last[0][0] = None
                elif p.last[1][0] is r:
                    p.last[1][0] = None
                p.drop_residual(r)
        ow.close()
        return rmse, it

    def remove_outliers(self):     # FS/FullSystemOptimize.cpp:507-526
        c = 0
        for f in self.frames:
            i = 0
            while i < len(f.points):
                if not f.points[i].residuals:
                    f.n_out += 1
                    f.points[i] = f.points[-1]
                    f.points.pop()
                    c += 1
                    continue
                i += 1
        return c

    def _is_oob(self, p, to_marg):     # FS/HessianBlocks.h:619-643
        vis = sum(1 for r in p.residuals if r.state_state == synth.RES_IN and r.target in to_marg)
        if len(p.residuals) >= MIN_GOOD_ACTIVE_RES and p.numGoodResiduals > MIN_GOOD_RES + 10 and len(p.residuals) - vis < MIN_GOOD_ACTIVE_RES:
            return True
        if p.last[0][1] == synth.RES_OOB:
            return True
        if len(p.residuals) < 2:
            return False
        return p.last[0][1] == synth.RES_OUTLIER and p.last[1][1] == synth.RES_OUTLIER

    def flag_points_for_removal(self):    # FS/FullSystem.cpp:535-614 + dropPointsF + marginalizePointsF
        to_marg_frames = [f for f in self.frames if f.flagged]
        inliers, dropped = [], 0
        for f in self.frames:
            holes = set()
            for i, p in enumerate(f.points):
                if p.idepth < 0 or not p.residuals:
                    f.n_out += 1
                    holes.add(i); dropped += 1
                elif self._is_oob(p, to_marg_frames) or f.flagged:
                    if len(p.residuals) >= MIN_GOOD_ACTIVE_RES and p.numGoodResiduals >= MIN_GOOD_RES:
                        inliers.append(p)
                    else:
                        f.n_out += 1
                        dropped += 1
                    holes.add(i)
            i = 0
            alive = [j not in holes for j in range(len(f.points))]
            pts = f.points
            while i < len(pts):
                if not alive[i]:
                    pts[i], alive[i] = pts[-1], alive[-1]
                    pts.pop(); alive.pop()
                    continue
                i += 1
        margd = 0
        if inliers:
            # the window as the C restatement sees it at this moment: remaining points + the inliers that are about to leave
            keep = [p for f in self.frames for p in f.points]
            order = {id(f): i for i, f in enumerate(self.frames)}
            allp = sorted(keep + inliers, key=lambda p: order[id(p.host)])   # frames -> points (stable)
            saved = {id(f): f.points for f in self.frames}
            for f in self.frames:
                f.points = [p for p in allp if p.host is f]
            ow, plist, rlist = self._pack()
            for f in self.frames:
                f.points = saved[id(f)]
            pos = {id(p): k for k, p in enumerate(plist)}
            # per-point results of the last accumulate of optimize() decide marginalise vs drop
            idh = ow.point_field("idepth_hessian")
            for k, p in enumerate(plist):
                idh[k] = p.idepth_hessian
            sel = np.array([pos[id(p)] for p in inliers], dtype=np.int32)
            flag = ow.marginalize_points(sel)
            Hold, bold = self.HM, self.bM
            self.HM, self.bM = ow.get_prior()
            if self.HMi is not None:     # expandHbtoFitImu(H, b); HM += setting_margWeightFac * H, OB/EnergyFunctional.cpp:928-932
                dH, db = orc.imu().expand(len(self.frames), self.HM - Hold, self.bM - bold)
                self.HMi, self.bMi = self.HMi + dH, self.bMi + db
            for p, fl in zip(inliers, flag):
                if fl:
                    p.host.n_marg += 1
                    margd += 1
                else:
                    p.host.n_out += 1
            ow.close()
        return margd, dropped + (len(inliers) - margd)

    def pixel_select(self, h, density):     # makeMaps + the scan of FullSystem::makeNewTraces (FS/FullSystem.cpp:1083-1095)
        if self.sel_slot is not h:
            self.sel.make_hists(h["absg"][0])
            self.sel_slot = h
        m, num = self.sel.make_maps(h["dI"], h["absg"], density)
        pad = 2
        sub = m[pad + 1:self.sc.h - pad - 2, pad + 1:self.sc.w - pad - 2]
        ys, xs = np.nonzero(sub)                                    # row-major
        u, v = (xs + pad + 1).astype(np.int32), (ys + pad + 1).astype(np.int32)
        return u, v, m[v, u].astype(np.float32)

    def immature_init(self, h, u, v):
        return orc.immature_init(self.tprm, h["dI"][0], u, v)

    def marginalize_flagged(self):     # FS/FullSystem.cpp:926-931, OB/EnergyFunctional.cpp:730-889 (prior), FS/FullSystemMarginalize.cpp:143-236
        out = []
        i = 0
        while i < len(self.frames):
            f = self.frames[i]
            if not f.flagged:
                i += 1
                continue
            assert not f.points
            out.append((f.frameID, f.pre.copy()))
            ow, plist, rlist = self._pack()
            if self.HMi is not None:     # the IMU form, OB/EnergyFunctional.cpp:733-889 (orc_imu_marginalize_frame)
                recs, keep = self._window_records()
                pr, dp = ow.frame_prior(i)
                self.HMi, self.bMi = orc.imu().marginalize_frame(self.sc.imu_settings, self.vio_calib(), recs, i, self._stitched_delta(), pr, dp,
                                                                 self.HMi, self.bMi, marg_weight=float(self.sc.params["margWeightFac"]))
            self.HM, self.bM = ow.marginalize_frame_prior(i)
            ow.close()
            for g in self.frames:
                if g is f:
                    continue
                for p in g.points:
                    for r in list(p.residuals):
                        if r.target is f:
                            if p.last[0][0] is r:
                                p.last[0][0] = None
                            elif p.last[1][0] is r:
                                p.last[1][0] = None
                            p.drop_residual(r)
                            break
            self.imu_hand_over([g.frameID for g in self.frames], f.frameID)
            self.frames.pop(i)
            i = 0
        return out

    def prior(self):
        return self.HM.copy(), self.bM.copy()

    def residual_set(self):
        return {p.key() + (r.target.frameID,) for f in self.frames for p in f.points for r in p.residuals}

    def point_set(self):
        return {p.key() for f in self.frames for p in f.points}
