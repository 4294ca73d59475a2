"""One keyframe's worth of the whole chain on the device against the oracle chain, stage by stage:
raw 8-bit frame -> photometric / geometric undistortion -> pyramid -> pixel selection -> ImmaturePoint constructors ->
traceOn against later frames -> candidate selection (distance map) -> optimizeImmaturePoint -> the activated points and
their residuals join the window -> optimize().  Index sets (selected pixels, trace statuses, accept / delete / activate
decisions, residual targets, residual states after the optimisation) are identical; poses agree within the fp32
accumulation yardstick of the backend tests."""
import dataclasses

import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import (ACT_ACTIVATED, IMMATURE_DTYPE, ActivateParams, Calib, PixselParams, TraceParams, random_pattern)
from tests import helpers as hp
from tests import immature_helpers as ih

pytestmark = pytest.mark.gpu


def _extend(win, new_pts, new_res):
    """append points / residuals and restore the frames -> points order of EnergyFunctional::allPoints"""
    pts = np.concatenate([win.points, new_pts])
    res = np.concatenate([win.resid, new_res])
    order = np.argsort(pts["host"], kind="stable")
    inv = np.empty(len(order), np.int64)
    inv[order] = np.arange(len(order))
    pts = pts[order]
    res = res.copy()
    res["point"] = inv[res["point"]]
    res = res[np.argsort(res["point"], kind="stable")]
    return dataclasses.replace(win, points=pts, resid=res)


def test_keyframe_chain_matches_oracle():
    from sos_slam_amd import host, lib
    win = synth.make_window("W7", extra_frames=1)
    n, w, h = win.n, win.w, win.h
    newest = n - 1
    ctx = lib.Context(w, h)
    K32 = np.array(win.K, np.float32)
    calib = Calib.from_K(K32.astype(np.float64))

    # ---- front end: raw frames through the (pass-through pinhole) camera file, then the pyramid
    cam_txt = f"Pinhole {K32[0]:.9g} {K32[1]:.9g} {K32[2]:.9g} {K32[3]:.9g} 0\n{w} {h}\nnone\n{w} {h}\n"
    und_g, und_o = lib.Undistorter(ctx, lib.camera_parse(cam_txt)), orc.Undistorter(cam_txt)
    dI, absg = [], []
    for f in range(n):
        raw = np.clip(np.rint(win.images[f]), 0, 255).astype(np.uint8)
        img_g = und_g.frame(raw, exposure=0.0, slot=f)
        img_o = und_o.frame(raw, exposure=0.0)
        assert np.array_equal(img_g, img_o)
        d, a = orc.make_images(img_o)
        dI.append(d)
        absg.append(a)
    # ---- makeNewTraces on keyframe `src`: pixel selection, then the constructors
    src = 2
    pattern = random_pattern(w * h)
    pprm, tprm, aprm = PixselParams.default(), TraceParams.default(), ActivateParams.default()
    sel_g, sel_o = lib.PixelSelector(ctx, pprm, pattern), orc.PixelSelector(pprm, pattern, w, h)
    sel_o.make_hists(absg[src][0])
    m_g, num_g = sel_g.make_maps(src, 800.0)
    m_o, num_o = sel_o.make_maps(dI[src], absg[src], 800.0)
    assert num_g == num_o and np.array_equal(m_g, m_o)
    u, v, typ = sel_g.list(pattern_padding=2)
    assert len(u) > 300
    im_g, im_o = ctx.immature_init(tprm, src, u, v), orc.immature_init(tprm, dI[src][0], u, v)
    ok = np.isfinite(im_o["energyTH"])          # FS/FullSystem.cpp:1091-1094
    im_g, im_o, typ = im_g[ok], im_o[ok], typ[ok]
    # ---- traceNewCoarse against the following frames
    for tgt in range(src + 1, n):
        KRKi, Kt, aff = ih.host_to_frame(win.K, win.frames[src]["camToWorld"], win.frames[tgt]["camToWorld"])
        im_g = ctx.immature_trace(tprm, tgt, im_g, KRKi, Kt, aff)
        im_o = orc.immature_trace(tprm, dI[tgt][0], im_o, KRKi, Kt, aff)
    for f in IMMATURE_DTYPE.names:
        if f != "pad":
            assert np.array_equal(im_g[f], im_o[f], equal_nan=True), f
    # ---- activatePointsMT: selection on the host, optimisation on the device
    KRKi1, Kt1 = ih.level1_to_newest(win, newest)
    pairs = ih.pair_tfms(win)
    hosts = np.full(len(im_g), src, np.int32)
    flagged = np.zeros(n, np.uint8)
    act_pts = win.points[::3]
    d_g, _ = host.activate_select(w // 2, h // 2, newest, KRKi1, Kt1, act_pts, 1.0, 3.0, im_g, hosts, typ, flagged)
    d_o, _ = orc.activate_select(w // 2, h // 2, newest, KRKi1, Kt1, act_pts, 1.0, 3.0, im_o, hosts, typ, flagged)
    assert np.array_equal(d_g, d_o)
    todo = np.flatnonzero(d_g == 1)
    a_g = ctx.immature_activate(aprm, calib, np.arange(n), pairs, im_g[todo], hosts[todo])
    a_o = orc.immature_activate(aprm, calib, [dI[f][0] for f in range(n)], pairs, im_o[todo], hosts[todo])
    for f in a_g.dtype.names:
        if f != "pad":
            assert np.array_equal(a_g[f], a_o[f], equal_nan=True), f
    won = todo[a_g["status"] == ACT_ACTIVATED]
    assert len(won) > 50
    # ---- the new points and their residuals join the window (FS/FullSystemOptPoint.cpp:151-185), then optimize()
    new_pts = np.zeros(len(won), dtype=synth.POINT_DTYPE)
    new_res = []
    for k, (ci, act) in enumerate(zip(won, a_g[a_g["status"] == ACT_ACTIVATED])):
        p = new_pts[k]
        p["u"], p["v"] = im_g["u"][ci], im_g["v"][ci]
        p["idepth_scaled"] = p["idepth_zero_scaled"] = act["idepth"]
        p["color"], p["weights"] = im_g["color"][ci], im_g["weights"][ci]
        p["host"] = src
        for t in range(n):
            if (int(act["inMask"]) >> t) & 1:
                new_res.append((len(win.points) + k, src, t, synth.RF_ISNEW, synth.RES_IN, 0.0))
    win2 = _extend(win, new_pts, np.array(new_res, dtype=synth.RESID_DTYPE))
    ow = hp.oracle_window(win2)
    sysm = host.System.from_window(win2)
    rm_o, it_o = ow.optimize(4)
    rm_g, it_g = sysm.optimize(4)
    assert it_o == it_g and abs(rm_o - rm_g) <= 1e-4 * rm_o
    ro = ow.res()
    alive_o = (ro["flags"] & 0x100) == 0          # residuals the final linearizeAll(true) kept (as in test_gpu_optimize.py)
    rg = sysm.residuals()
    assert len(rg["state_state"]) == int(alive_o.sum())
    assert np.array_equal(rg["state_state"], ro["state_state"][alive_o])
    for f in range(n):
        assert np.abs(sysm.frame(f)["state"] - ow.frame(f)["state"]).max() < 1e-4
        assert np.abs(sysm.frame(f)["camToWorld"] - ow.frame(f)["camToWorld"]).max() < 1e-4
    sysm.close()
    for x in (sel_g, und_g):
        x.close()
    ctx.close()
