"""The frame-rate loop in C++ (sosf_sequence: addActiveFrame -> trackNewestCoarse -> traceNewCoarse -> keyframe decision -> makeKeyFrame,
csrc/host/sos_sequence.cpp) against the same loop strung together in Python over the same facade calls (tests/rolling.py DeviceChain,
which the rolling-window tests hold against the oracle chains): same keyframes, same windows, same keyframes leaving in the same order,
poses within the sensitivity two free-running chains have (their float transforms are formed with differently ordered 3 x 3 products)."""
import os

import numpy as np
import pytest

from sos_slam_amd import synth
from sos_slam_amd.records import SequenceParams
from tests import rolling

_EMULATED = os.environ.get("SOS_EMU") == "1" and not __import__("torch").cuda.is_available()   # looser mono-VIO bar only under tests/emu

pytestmark = pytest.mark.gpu


def _cpp_chain(sc):
    """a DeviceChain used for its front end and window set-up only; the loop runs in sosf_sequence"""
    from sos_slam_amd import host
    d = rolling.DeviceChain(sc)
    hs = [d.front_end(sc.raw[i]) for i in range(sc.n0)]
    images = [d.irradiance(h) for h in hs]
    pts, res = sc.bootstrap_points(images)
    d.init_window(hs, [sc.poses[i] for i in range(sc.n0)], [sc.aff_true[i] for i in range(sc.n0)], pts, res)
    prm = SequenceParams.default(desired_points=sc.desired_points, immature_density=sc.immature_density, kf_every=sc.kf_every)
    seq = host.Sequence(d.sysm, prm, sc.pattern)
    return d, seq


@pytest.mark.parametrize("kf_every,n_frames", [(1, 14), (3, 4 + 3 * 5)])
def test_cpp_sequence_loop_matches_the_python_loop(kf_every, n_frames):
    _case(kf_every, n_frames)


def _case(kf_every, n_frames):
    kw = dict(n_frames=n_frames, kf_every=kf_every)
    if kf_every > 1:
        kw.update(step=0.07 / 3, rot=0.008 / 3)
    sc = rolling.Scenario(**kw)
    ref = rolling.DeviceChain(sc)
    r_ref, it_ref = ref.bootstrap()
    d, seq = _cpp_chain(sc)
    r_cpp, it_cpp = seq.bootstrap()
    assert it_ref == it_cpp and abs(r_ref - r_cpp) <= 1e-6 * r_ref
    for f in range(sc.n0):      # the first immature sets: pixel selection + constructors are bit-exact stages
        rec, ty = seq.immature(f)
        assert np.array_equal(rec["u"], ref.imm[f]["u"]) and np.array_equal(rec["energyTH"], ref.imm[f]["energyTH"])
        assert np.array_equal(ty, ref.imm_type[f])
    k, kfs, worst = sc.n0, 0, 0.0
    while True:
        lg = ref.step()
        if lg is None:
            break
        # the same frames through the C++ loop, up to and including the keyframe
        while True:
            slot = d.front_end(sc.raw[k])
            T_init = None
            if k == sc.n0:   # the first frame has no motion history: the scenario's guess, as the Python loop takes it
                ids = d.window_ids()
                T_init = rolling.se3_mul(rolling.se3_inv(sc.poses[k]), sc.poses[k - 1] if kf_every == 1 else sc.poses[ids[-1]])
                T_init = rolling.se3_mul(synth.se3_exp12(np.array([0.002, -0.001, 0.001, 0.001, -0.001, 0.0005])), T_init)
            out = seq.add_active_frame(slot, k, T_init)
            assert out.trackingOk == 1, k
            k += 1
            if out.isKeyframe:
                break
            d.sysm.release_image(slot)
        assert k - 1 == lg.frameID
        kfs += 1
        # (the Python loop logs the window as optimize() saw it; the C++ call returns after marginalizeFlaggedFrames)
        left = list(out.margFrameIDs[:out.nMargFrames])
        now = d.window_ids()
        assert sorted(now + left) == lg.window_ids, (k, now, left, lg.window_ids)
        assert left == [f for f, _ in lg.marginalized]
        assert out.iterations == lg.iterations or abs(out.rmse - lg.rmse) <= 1e-3 * lg.rmse
        e_trk = np.abs(np.array(out.refToNew[:]) - lg.tracked_pose).max()
        assert e_trk < 1e-4, (k, e_trk)
        for fid in lg.window_ids:    # poses after optimize(): of the keyframes that stay, and of those that left as they left
            pose = d.kf_pose(now.index(fid)) if fid in now else np.array(out.margCamToWorld[12 * left.index(fid):12 * left.index(fid) + 12])
            e = np.abs(pose - lg.window_poses[fid]).max()
            worst = max(worst, e)
            assert e < 2e-4, (k, fid, e)
        assert abs(out.nActivated - len(lg.activated)) <= max(4, 0.02 * len(lg.activated)), (out.nActivated, len(lg.activated))
        assert abs(out.nMargPoints - lg.marg_points) <= max(4, 0.03 * lg.marg_points)
        assert abs(out.nNewImmature - lg.new_immature) <= 2
    print(f"{kfs} keyframes through both loops; worst window pose difference {worst:.2e}")
    assert kfs >= 5
    seq.close()
    d.close()
    ref.close()


# the IMU / stereo branches of makeKeyFrame in the C++ loop (sosf_sequence_enable_imu / _enable_stereo, sosf_add_active_frame_ex) against
# the Python loop of the rolling visual-inertial tests
@pytest.mark.parametrize("stereo", [False, True])
def test_cpp_sequence_loop_visual_inertial(stereo):
    _case_vio(stereo)


def _case_vio(stereo):
    sc = rolling.Scenario(vio=True, stereo=stereo, n_frames=16)
    ref = rolling.DeviceChain(sc)
    r_ref, it_ref = ref.bootstrap()
    d, seq = _cpp_chain(sc)
    seq.enable_imu(sc.imu_settings, [sc.ts[i] for i in range(sc.n0)], [sc.imu[i] for i in range(sc.n0)])
    sr = -1
    if stereo:
        seq.enable_stereo(sc.stereo_tfm, sc.scale_opt_thres)
        sr = d.front_end(sc.raw_right[sc.n0 - 1])
    r_cpp, it_cpp = seq.bootstrap(stereo_slot=sr)
    if stereo:
        d.sysm.release_image(sr)
    assert it_ref == it_cpp and abs(r_ref - r_cpp) <= 1e-6 * r_ref
    K = rolling.Chain._IMU_K
    k, kfs, worst_pose, worst_state = sc.n0, 0, 0.0, 0.0
    last = None
    while True:
        lg = ref.step()
        if lg is None:
            lg = last
            break
        last = lg
        slot = d.front_end(sc.raw[k])
        T_init = None
        if k == sc.n0:
            T_init = rolling.se3_mul(rolling.se3_inv(sc.poses[k]), sc.poses[k - 1])
            T_init = rolling.se3_mul(synth.se3_exp12(np.array([0.002, -0.001, 0.001, 0.001, -0.001, 0.0005])), T_init)
        sr = d.front_end(sc.raw_right[k]) if stereo else -1
        out = seq.add_active_frame(slot, k, T_init, timestamp=float(sc.ts[k]), imu=sc.imu[k], stereo_slot=sr)
        if stereo:
            d.sysm.release_image(sr)
        assert out.trackingOk == 1 and out.isKeyframe == 1, k
        k += 1
        kfs += 1
        # (the Python loop logs the window as optimize() saw it; the C++ call returns after marginalizeFlaggedFrames)
        left = list(out.margFrameIDs[:out.nMargFrames])
        now = d.window_ids()
        assert sorted(now + left) == lg.window_ids, (k, now, left, lg.window_ids)
        assert left == [f for f, _ in lg.marginalized]
        cal = seq.imu_calib()
        assert int(cal.imu_initialized) == int(lg.vio["init"]), k          # initializeImu at the fifth keyframe in both loops
        if os.environ.get("SOS_SEQ_TRACE"):
            print(f"kf {k - 1}: scale cpp {cal.scale * 200:.6f} py {lg.vio['scale'] * 200:.6f} trapped {int(cal.scale_trapped)}/{lg.vio['trapped']} left {left}", flush=True)
        # two free-running loops: until the scale is trapped it is the weakly observable direction of a monocular visual-inertial window
        # and the loops drift apart along it (1e-5 at the IMU initialisation, 3e-3 six keyframes later under tests/emu); what holds the
        # C++ loop against the ORACLE chain, with the oracle's own fp32-vs-fp64 distance as the bar, is tests/test_gpu_rolling_vio.py
        tol = (5e-4 + 1e-3 * abs(lg.vio["scale"]) * 200) if stereo else 1e-2 * abs(lg.vio["scale"]) * 200
        assert abs(cal.scale - lg.vio["scale"]) * 200 <= tol, (k, cal.scale, lg.vio["scale"])
        for fid in lg.window_ids:    # poses after optimize(): of the keyframes that stay, and of those that left as they left
            pose = d.kf_pose(now.index(fid)) if fid in now else np.array(out.margCamToWorld[12 * left.index(fid):12 * left.index(fid) + 12])
            e = np.abs(pose - lg.window_poses[fid]).max()
            worst_pose = max(worst_pose, e)
            assert e < (5e-4 if stereo or not _EMULATED else 1.5e-3), (k, fid, e)   # (without the stereo scale the loops drift along the scale direction: 5.3e-4 at keyframe 15 under tests/emu)
            if fid not in now:       # the IMU states are logged (and kept) for the keyframes that stay
                continue
            st, ze, ve = seq.imu(fid)
            es = np.abs(K * (st - lg.vio["states"][fid])).max()
            worst_state = max(worst_state, es)
            # the rolling tests' own yardstick: two ORACLE chains end 0.28 apart here; without the stereo scale the loops also drift along
            # the scale direction (0.29 after 13 keyframes under tests/emu)
            assert es < (0.25 if stereo else 0.6), (k, fid, es)
            assert np.abs(ve - lg.vio["vel"][fid]).max() < 1e-2, (k, fid)
    print(f"{kfs} keyframes through both visual-inertial loops (stereo={stereo}); worst pose difference {worst_pose:.2e}, "
          f"worst scaled IMU state difference {worst_state:.2e}; scale trapped {int(seq.imu_calib().scale_trapped)}/{lg.vio['trapped']}")
    assert kfs >= 8
    seq.close()
    d.close()
    ref.close()
