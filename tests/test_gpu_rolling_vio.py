"""Rolling-window parity in visual-inertial mode (BASELINE configs 2-3 in synthetic form: a monocular sequence with IMU samples
generated from the rendered trajectory).  The chain of tests/test_gpu_rolling_window.py plus, per keyframe, what setting_enable_imu
adds (FS/FullSystem.cpp:800-807, 841-849, 878-886; FS/FullSystemOptimize.cpp:459-479; OB/EnergyFunctional.cpp:666-677, 733-889,
928-932, 1053-1171): setImuData + propagateImuState, initializeImu at the fifth keyframe, the IMU branch of solveSystemF in every
optimize() from then on, updateVel / setImuStateZero / tryTrapScale after it, marginalizePointsF into the expanded prior and the
IMU form of marginalizeFrame.

Device chain: the facade keeps the expanded prior (sosf_set_imu with NULL priors), its front-end functions (sosf_imu_*) do the
state propagation; oracle chains (fp32 restatement, and fp64-accumulated as the yardstick): oracle/imu_frontend.py (NumPy),
orc_imu_* for the assembly, the prior kept in NumPy.  Free-running from the same raw frames and IMU samples.

Compared per keyframe: keyframe-level decisions (flagged / marginalised keyframes, window, IMU initialisation and scale trap:
identical; iteration counts: at most two knife-edge differences of the termination test over the sequence), poses in the window
and leaving it, metric scale, the 21 IMU states and the velocity of every keyframe, the expanded prior.  The visual-inertial
window is far more sensitive to fp32 rounding than the visual one (the cubic spline coefficients are barely constrained over
50 ms: the two ORACLE chains end 0.28 apart in the scaled IMU states and 3e-4 in the poses), so the yardstick is the only
meaningful bar: every quantity within FACTOR x the running maximum of the oracle's own fp32-vs-fp64 distance, or the stated
absolute tolerance, whichever is larger."""
import numpy as np
import pytest

from tests import rolling

pytestmark = pytest.mark.gpu

POSE_TOL, SCALE_TOL, STATE_TOL, VEL_TOL = 5e-5, 5e-4, 5e-2, 2e-3   # camToWorld entries; scale * 200; scaled IMU states; m/s
FACTOR = 5.0


def _scaled(A, ref):
    s = 1.0 / np.sqrt(np.abs(np.diag(ref)) + 10)
    return A * s[:, None] * s[None, :]


GEOMETRIES = {
    "qvga": dict(n_frames=20),
    # the EuRoC cam0 geometry at full size with the reference's default densities (setting_desiredPointDensity 2000,
    # setting_desiredImmatureDensity 1500): BASELINE config 2 in synthetic form
    "euroc_752x480": dict(w=752, h=480, n_frames=16, points0=1200, desired_points=2000.0, immature_density=1500.0),
    # stereo-inertial (BASELINE config 2, "EuRoC V1_02 stereo-inertial"): FullSystem::optimizeScale on the stereo partner of every
    # keyframe (seven guesses until trapped), setting_enable_scale_opt semantics in initializeImu / optimize / the IMU factors
    "euroc_752x480_stereo": dict(w=752, h=480, n_frames=14, points0=1200, desired_points=2000.0, immature_density=1500.0, stereo=True),
    # TUM-VI geometry (BASELINE config 3): 512 x 512, four pyramid levels
    "tumvi_512x512": dict(w=512, h=512, n_frames=14, points0=1000, desired_points=2000.0, immature_density=1500.0),
}


@pytest.mark.parametrize("geom", list(GEOMETRIES))
def test_rolling_window_visual_inertial(geom):
    sc = rolling.Scenario(vio=True, **GEOMETRIES[geom])
    dev, orc_, tru = rolling.device_chain(sc), rolling.OracleChain(sc), rolling.OracleChain(sc, truth=True)
    for c in (dev, orc_, tru):
        c.bootstrap()
    bad = []

    def check(cond, what):
        if not cond:
            bad.append(what)

    deferred = []     # (yardstick, value, tolerance, what)

    run = dict(pose=0.0, scale=0.0, state=0.0, vel=0.0, HM=0.0)
    worst = dict(pose=0.0, scale=0.0, state=0.0, vel=0.0, leave=0.0, leave_noise=0.0)
    left = its_diff = 0
    trap_apart = None
    from sos_slam_amd import host as _host
    _host.imu().solve_stats(reset=True)
    trapped_kfs = 0
    while dev.next_frame < sc.n_frames:
        lg, lo, lt = dev.step(), orc_.step(), tru.step()
        k = lg.frameID
        vg, vo, vt = lg.vio, lo.vio, lt.vio
        assert lg.flagged == lo.flagged and lg.window_ids == lo.window_ids, k
        if sc.stereo:      # the stereo scale of this keyframe: accepted on both sides, same state, scale and error within the bar
            (kg, ng, eg_, stg), (ko, no, eo_, sto) = dev.scale_log[-1], orc_.scale_log[-1]
            assert (kg, stg) == (ko, sto) and (ng > 0) == (no > 0), (dev.scale_log[-1], orc_.scale_log[-1])
            check(abs(ng - no) <= 2e-4 * abs(no) and abs(eg_ - eo_) <= 3e-2 * eo_, (k, "stereo scale", ng, no, eg_, eo_))   # the template (point set) differs by knife edges
        assert vg["init"] == vo["init"], k
        if vg["trapped"] != vo["trapped"]:
            # CalibHessian::tryTrapScale thresholds the spread of the last ten scales: a knife edge of a continuous quantity, like the
            # iteration count.  Two chains whose scales agree can trap one keyframe apart, and from there on they linearise differently
            # (first-estimate Jacobians on one side only): as in the seed ensembles (tests/rolling_ensemble.py) the chains are compared up
            # to here, and the event is only accepted where the two scales agree within the scale bar
            trap_apart = (k, abs(vg["scale"] - vo["scale"]) * 200)
            break
        its_diff += int(lg.iterations != lo.iterations)
        trapped_kfs += int(vg["trapped"])
        for fid in lg.window_ids:
            run["pose"] = max(run["pose"], np.abs(lo.window_poses[fid] - lt.window_poses[fid]).max())
        for fid in lg.window_ids:
            e = np.abs(lg.window_poses[fid] - lo.window_poses[fid]).max()
            worst["pose"] = max(worst["pose"], e)
            deferred.append(("pose", e, POSE_TOL, (k, fid, "pose", e)))
        e_s, n_s = abs(vg["scale"] - vo["scale"]) * 200, abs(vo["scale"] - vt["scale"]) * 200
        run["scale"] = max(run["scale"], n_s)
        worst["scale"] = max(worst["scale"], e_s)
        # judged at the end of the sequence against the maximum of the oracle's own fp32-vs-fp64 distance over ALL keyframes: the scale
        # is the weakly observable direction right after the IMU initialisation, where the yardstick's running maximum has seen one draw
        deferred.append(("scale", e_s, SCALE_TOL, (k, "scale", e_s)))
        assert set(vg["states"]) == set(vo["states"])
        for fid in vg["states"]:
            sg, so, st = (dev.vio_scaled(v["states"][fid]) for v in (vg, vo, vt))
            run["state"] = max(run["state"], np.abs(so - st).max())
            run["vel"] = max(run["vel"], np.abs(vo["vel"][fid] - vt["vel"][fid]).max())
        for fid in vg["states"]:
            sg, so = dev.vio_scaled(vg["states"][fid]), dev.vio_scaled(vo["states"][fid])
            e = np.abs(sg - so).max()
            worst["state"] = max(worst["state"], e)
            deferred.append(("state", e, STATE_TOL, (k, fid, "state_imu", e)))
            ev = np.abs(vg["vel"][fid] - vo["vel"][fid]).max()
            worst["vel"] = max(worst["vel"], ev)
            deferred.append(("vel", ev, VEL_TOL, (k, fid, "vel", ev)))
        assert [f for f, _ in lg.marginalized] == [f for f, _ in lo.marginalized], k
        for (fid, pg), (_, po), (_, pt) in zip(lg.marginalized, lo.marginalized, lt.marginalized):
            e_go, e_ot = np.abs(pg - po).max(), np.abs(po - pt).max()
            worst["leave"], worst["leave_noise"] = max(worst["leave"], e_go), max(worst["leave_noise"], e_ot)
            left += 1
            deferred.append(("pose", e_go, POSE_TOL, (fid, "leaves", e_go)))
        if vg["HMi"] is not None:
            assert vg["HMi"].shape == vo["HMi"].shape
            eg = np.abs(_scaled(vg["HMi"] - vt["HMi"], vt["HMi"])).max()
            eo = np.abs(_scaled(vo["HMi"] - vt["HMi"], vt["HMi"])).max()
            m = np.abs(_scaled(vt["HMi"], vt["HMi"])).max()
            run["HM"] = max(run["HM"], eo)
            # one point marginalised on one side only moves an entry by a few % (5 % seen at the second keyframe after the IMU
            # initialisation, where an entry of the young prior is the sum of a handful of marginalised points)
            check(eg <= FACTOR * run["HM"] + 8e-2 * m, (k, "HMi", eg, eo, m))
        print(f"KF {k}: its {lg.iterations}/{lo.iterations} rmse {lg.rmse:.5f}/{lo.rmse:.5f}; scale {vg['scale'] * 200:.6f}/{vo['scale'] * 200:.6f}/{vt['scale'] * 200:.6f} "
              f"trapped {vg['trapped']}; flagged {lg.flagged}; leaves {[f for f, _ in lg.marginalized]}; running oracle-vs-truth: pose {run['pose']:.2e} "
              f"scale {run['scale']:.2e} state {run['state']:.2e} vel {run['vel']:.2e}")
    print(f"{left} keyframes left; worst |dev-orc|: window pose {worst['pose']:.2e}, leaving pose {worst['leave']:.2e} (oracle fp32-vs-fp64 "
          f"{worst['leave_noise']:.2e}), scale {worst['scale']:.2e}, scaled IMU state {worst['state']:.2e}, velocity {worst['vel']:.2e}; "
          f"final scale {vg['scale'] * 200:.5f} (true {sc.scale_true})")
    for kind, val, tol, what in deferred:
        check(val < max(tol, FACTOR * run[kind]), what + (run[kind],))
    print("violations:", bad)
    # once the scale is trapped the facade's loop solves on the kept factor (sosf_imu_solve_prepare / _finish: first-estimate Jacobians,
    # OB/EnergyFunctional.cpp:1053-1171 with everything but the visual block constant): the form the solves of this chain actually took
    kept, rebuilt, literal = _host.imu().solve_stats()
    print(f"IMU solves of the device chain: {kept} on a kept factor, {rebuilt} factor rebuilds, {literal} literal")
    if trapped_kfs >= 2:
        assert kept > 0 and rebuilt >= 1, (kept, rebuilt, literal, trapped_kfs)
    dev.close()
    assert not bad, bad
    assert its_diff <= 2, its_diff
    if trap_apart is not None:
        print(f"scale trapped one keyframe apart at keyframe {trap_apart[0]} (scales {trap_apart[1]:.2e} apart): compared up to there")
        assert trap_apart[1] < max(SCALE_TOL, FACTOR * run["scale"]), trap_apart
        assert trap_apart[0] >= sc.n0 + 8 and left >= 5 and vg["init"] == 1, (trap_apart, left)
    else:
        assert left >= (10 if geom == "qvga" else 5) and vg["init"] == 1
    assert abs(vg["scale"] - vo["scale"]) * 200 < 5e-3 and abs(vo["scale"] * 200 - sc.scale_true) < 0.2
