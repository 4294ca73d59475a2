"""CPU known-answer test of the oracle's ImmaturePoint::traceOn restatement: on the synthetic scene the traced
inverse-depth interval of well-conditioned points must contain the inverse depth the scene was rendered with."""
import numpy as np

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import TraceParams
from tests import immature_helpers as ih


def test_traced_interval_contains_true_idepth():
    win = synth.make_window("T6", extra_frames=2, noise_sigma=0.5)
    prm = TraceParams.default()
    host = 2
    dI_host, _ = orc.make_images(win.images[host])
    u, v, idepth = ih.candidates(win, host, 150)
    pts = orc.immature_init(prm, dI_host[0], u, v)
    assert np.all(pts["lastTraceStatus"] == 5) and np.all(np.isnan(pts["idepth_max"])) and np.all(pts["idepth_min"] == 0)
    assert np.allclose(pts["energyTH"], 8 * 144.0)
    host_c2w = win.frames[host]["camToWorld"]
    status_seen = set()
    # trace against three frames in turn, as successive tracked frames would
    for frame_img, frame_c2w in ((win.images[host + 1], win.frames[host + 1]["camToWorld"]),
                                 (win.images[host + 2], win.frames[host + 2]["camToWorld"]),
                                 (win.extra_images[0], win.extra_poses[0])):
        dI_new, _ = orc.make_images(frame_img)
        KRKi, Kt, aff = ih.host_to_frame(win.K, host_c2w, frame_c2w)
        pts = orc.immature_trace(prm, dI_new[0], pts, KRKi, Kt, aff)
        status_seen |= set(int(s) for s in pts["lastTraceStatus"])
    good = (pts["lastTraceStatus"] == 0) & np.isfinite(idepth)
    assert good.sum() > 30, (good.sum(), status_seen)
    lo, hi = pts["idepth_min"][good].astype(np.float64), pts["idepth_max"][good].astype(np.float64)
    assert np.all(lo <= hi)
    width = hi - lo
    inside = (idepth[good] >= lo - 0.5 * width - 0.01) & (idepth[good] <= hi + 0.5 * width + 0.01)
    assert inside.mean() > 0.9, inside.mean()
    assert np.median(width / np.maximum(idepth[good], 1e-6)) < 0.5   # the interval really narrowed
    assert len(status_seen) >= 2


def test_activation_moves_idepth_towards_scene_value():
    """Known-answer test of the optimizeImmaturePoint restatement: starting from an interval whose midpoint is off,
    the 1-D LM must end closer to the inverse depth the scene was rendered with, with residuals IN towards the frames
    that see the point; degenerate candidates take the SKIP / DELETE exits."""
    from sos_slam_amd.records import ACT_ACTIVATED, ACT_DELETE, ACT_SKIP, ActivateParams, Calib
    win = synth.make_window("T6", noise_sigma=0.5, idepth_noise=0.0, state_noise=0.0)
    prm, aprm = TraceParams.default(), ActivateParams.default()
    calib = Calib.from_K(win.K)
    dI0 = [orc.make_images(win.images[f])[0][0] for f in range(win.n)]
    pairs = ih.pair_tfms(win)
    parts, hosts, truth = [], [], []
    for host in range(win.n):
        u, v, idepth = ih.candidates(win, host, 60, seed=host)
        keep = np.isfinite(idepth)
        pts = orc.immature_init(prm, dI0[host], u[keep], v[keep])
        parts.append(pts)
        hosts.append(np.full(len(pts), host, np.int32))
        truth.append(idepth[keep])
    pts, hosts, truth = np.concatenate(parts), np.concatenate(hosts), np.concatenate(truth)
    # the window's points sit at sub-pixel positions; the candidates are rounded to integers, so take the true idepth
    # only as approximately known (smooth surface): interval midpoint off by +6 %
    pts["idepth_min"] = (truth * 0.96).astype(np.float32)
    pts["idepth_max"] = (truth * 1.16).astype(np.float32)
    out = orc.immature_activate(aprm, calib, dI0, pairs, pts, hosts)
    act = out["status"] == ACT_ACTIVATED
    assert act.mean() > 0.7, np.bincount(out["status"] + 1)
    start = 0.5 * (pts["idepth_min"] + pts["idepth_max"])
    err0, err1 = np.abs(start[act] - truth[act]), np.abs(out["idepth"][act] - truth[act])
    assert np.median(err1) < 0.35 * np.median(err0), (np.median(err0), np.median(err1))
    assert np.all(out["inMask"][act] != 0)
    assert np.all((out["inMask"] >> hosts.astype(np.uint32)) & 1 == 0)   # never a residual towards the own host
    assert np.all(out["Hdd"][act] >= aprm.minIdepthH_act)
    # exits: no interval yet (idepth_max NaN) -> every residual OOB, H = 0 -> SKIP; NaN energyTH -> DELETE
    bad = pts[:4].copy()
    bad["idepth_max"][:2] = np.nan
    bad["energyTH"][2:] = np.nan
    o2 = orc.immature_activate(aprm, calib, dI0, pairs, bad, hosts[:4])
    assert list(o2["status"][:2]) == [ACT_SKIP, ACT_SKIP] and np.all(o2["Hdd"][:2] == 0)
    assert np.all(o2["status"][2:] == ACT_DELETE)
    # minObs above the window size -> outlier exit
    o3 = orc.immature_activate(ActivateParams.default(minObs=win.n), calib, dI0, pairs, pts[:20], hosts[:20])
    assert np.all((o3["status"] == ACT_DELETE) | (o3["status"] == ACT_SKIP))
