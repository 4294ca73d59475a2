"""GPU parity of the immature-point kernels (ImmaturePoint constructor and traceOn) against the oracle: every field of
every record bit-exact, over successive traces (UNINITIALIZED -> GOOD / SKIPPED / BADCONDITION / OUTLIER / OOB)."""
import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import synth
from sos_slam_amd.records import IMMATURE_DTYPE, TraceParams
from tests import immature_helpers as ih

pytestmark = pytest.mark.gpu


def _same(a, b):
    for f in IMMATURE_DTYPE.names:
        if f == "pad":
            continue
        if not np.array_equal(a[f], b[f], equal_nan=True):
            bad = np.flatnonzero(~np.all(np.isclose(a[f].reshape(len(a), -1), b[f].reshape(len(b), -1), rtol=0, atol=0,
                                                   equal_nan=True), axis=1))
            return f"{f}: {len(bad)} records differ, first {bad[:5]}"
    return None


@pytest.mark.parametrize("name,count", [("T6", 700), ("W7", 3000)])
def test_init_and_successive_traces_bit_exact(name, count):
    from sos_slam_amd import lib
    win = synth.make_window(name, extra_frames=2)
    prm = TraceParams.default()
    ctx = lib.Context(win.w, win.h)
    for i in range(win.n):
        ctx.make_pyramid(i, win.images[i])
    ctx.make_pyramid(win.n, win.extra_images[0])
    host = 1
    u, v, _ = ih.candidates(win, host, count)
    # a few positions at the border and one NaN pixel in the pattern
    u[:4] = [0, 1, win.w - 1, win.w - 2]
    v[:4] = [0, 1, win.h - 1, win.h - 2]
    dI_host, _ = orc.make_images(win.images[host])
    p_o = orc.immature_init(prm, dI_host[0], u, v)
    p_g = ctx.immature_init(prm, host, u, v)
    assert _same(p_g, p_o) is None, _same(p_g, p_o)
    host_c2w = win.frames[host]["camToWorld"]
    seen = set()
    frames = [(host + 1, win.images[host + 1], win.frames[host + 1]["camToWorld"]),
              (host + 2, win.images[host + 2], win.frames[host + 2]["camToWorld"]),
              (win.n, win.extra_images[0], win.extra_poses[0]),
              (0, win.images[0], win.frames[0]["camToWorld"])]
    for slot, img, c2w in frames:
        dI_new, _ = orc.make_images(img)
        KRKi, Kt, aff = ih.host_to_frame(win.K, host_c2w, c2w, frame_aff=(0.02, 1.5))
        p_o = orc.immature_trace(prm, dI_new[0], p_o, KRKi, Kt, aff)
        p_g = ctx.immature_trace(prm, slot, p_g, KRKi, Kt, aff)
        assert _same(p_g, p_o) is None, (slot, _same(p_g, p_o))
        seen |= set(int(s) for s in p_o["lastTraceStatus"])
    assert {0, 1}.issubset(seen) and len(seen) >= 3, seen   # GOOD, OOB and at least one more class occurred
    # other settings: no GN refinement, wider search
    prm2 = TraceParams.default(GNIterations=0, maxPixSearch=0.05, slackInterval=0.5)
    p0 = orc.immature_init(prm2, dI_host[0], u, v)
    dI_new, _ = orc.make_images(win.images[host + 1])
    KRKi, Kt, aff = ih.host_to_frame(win.K, host_c2w, win.frames[host + 1]["camToWorld"])
    a = orc.immature_trace(prm2, dI_new[0], p0, KRKi, Kt, aff)
    b = ctx.immature_trace(prm2, host + 1, p0, KRKi, Kt, aff)
    assert _same(b, a) is None, _same(b, a)
    # the one-launch form over several hosts equals the per-host calls
    hosts = [0, 1, 2]
    parts, tabs = [], []
    for hh in hosts:
        uu, vv, _ = ih.candidates(win, hh, 200, seed=hh)
        parts.append(ctx.immature_init(prm, hh, uu, vv))
        tabs.append(ih.host_to_frame(win.K, win.frames[hh]["camToWorld"], win.extra_poses[0]))
    per_host = [ctx.immature_trace(prm, win.n, pp, *tt) for pp, tt in zip(parts, tabs)]
    allp = ctx.immature_trace_all(prm, win.n, np.concatenate(parts), np.concatenate([np.full(len(pp), k, np.int32) for k, pp in enumerate(parts)]),
                                  np.stack([t[0].reshape(-1) for t in tabs]), np.stack([t[1] for t in tabs]), np.stack([t[2] for t in tabs]))
    assert _same(allp, np.concatenate(per_host)) is None
    # empty input
    assert len(ctx.immature_trace(prm, host + 1, p0[:0], KRKi, Kt, aff)) == 0
    ctx.close()


def _act_same(a, b):
    for f in a.dtype.names:
        if f == "pad":
            continue
        if not np.array_equal(a[f], b[f], equal_nan=True):
            bad = np.flatnonzero(~((a[f] == b[f]) | ((a[f] != a[f]) & (b[f] != b[f]))))
            return f"{f}: {len(bad)} records differ, first {bad[:5]}: {a[f][bad[:3]]} vs {b[f][bad[:3]]}"
    return None


@pytest.mark.parametrize("name,per_host", [("T6", 150), ("W7", 300), ("W12", 120)])
def test_activation_bit_exact(name, per_host):
    """optimizeImmaturePoint through the C-ABI against the oracle: status, inverse depth, IN mask, energy, Hdd, bd and
    iteration count of every candidate identical (windows of 6, 7 and 12 keyframes: one and two residual chunks)."""
    from sos_slam_amd import lib
    from sos_slam_amd.records import ActivateParams, Calib
    win = synth.make_window(name)
    prm = TraceParams.default()
    calib = Calib.from_K(win.K)
    ctx = lib.Context(win.w, win.h)
    dI0 = []
    for i in range(win.n):
        ctx.make_pyramid(i, win.images[i])
        dI0.append(orc.make_images(win.images[i])[0][0])
    rng = np.random.default_rng(11)
    aff = [(0.01 * rng.standard_normal(), 2.0 * rng.standard_normal()) for _ in range(win.n)]
    pairs = ih.pair_tfms(win, aff)
    parts, hosts = [], []
    for host in range(win.n):
        u, v, idepth = ih.candidates(win, host, per_host, seed=host)
        pts = ctx.immature_init(prm, host, u, v)
        known = np.isfinite(idepth)
        # intervals: around the window's inverse depth where known (off-centre, different widths), wide guesses elsewhere
        wdt = rng.uniform(0.02, 0.4, len(pts))
        off = rng.uniform(-0.1, 0.1, len(pts))
        mid = np.where(known, idepth * (1 + off), rng.uniform(0.05, 1.5, len(pts)))
        pts["idepth_min"] = (mid * (1 - wdt)).astype(np.float32)
        pts["idepth_max"] = (mid * (1 + wdt)).astype(np.float32)
        parts.append(pts)
        hosts.append(np.full(len(pts), host, np.int32))
    pts, hosts = np.concatenate(parts), np.concatenate(hosts)
    # edge cases: never traced (NaN upper bound), negative / huge inverse depth, NaN energyTH, border pixels
    pts["idepth_max"][0:3] = np.nan
    pts["idepth_min"][3:6] = -3.0
    pts["idepth_max"][6:9] = 40.0
    pts["energyTH"][9:12] = np.nan
    pts["u"][12:15] = [2, win.w - 3, 5]
    pts["v"][12:15] = [2, win.h - 3, win.h - 4]
    for aprm in (ActivateParams.default(), ActivateParams.default(GNIts=6, minObs=2, minIdepthH_act=30.0, huberTH=4.0)):
        o_o = orc.immature_activate(aprm, calib, dI0, pairs, pts, hosts)
        o_g = ctx.immature_activate(aprm, calib, np.arange(win.n), pairs, pts, hosts)
        assert _act_same(o_g, o_o) is None, _act_same(o_g, o_o)
        assert {-1, 0, 1}.issubset(set(int(s) for s in o_o["status"])), np.bincount(o_o["status"] + 1)
    # image slots in another order than the frame idx
    perm = np.arange(win.n)[::-1].copy()
    ctx2 = lib.Context(win.w, win.h)
    for i in range(win.n):
        ctx2.make_pyramid(int(perm[i]), win.images[i])
    o_p = ctx2.immature_activate(ActivateParams.default(), calib, perm, pairs, pts, hosts)
    assert _act_same(o_p, orc.immature_activate(ActivateParams.default(), calib, dI0, pairs, pts, hosts)) is None
    assert len(ctx.immature_activate(ActivateParams.default(), calib, np.arange(win.n), pairs, pts[:0], hosts[:0])) == 0
    ctx2.close()
    ctx.close()


def test_activate_points_flow_index_sets():
    """FullSystem::activatePointsMT end to end: traced candidates -> host-side selection (distance map) -> device-side
    optimizeImmaturePoint -> the sets of activated / deleted / kept candidates and the new points' inverse depths and
    residual targets, identical to the oracle's flow."""
    from sos_slam_amd import host, lib
    from sos_slam_amd.records import ActivateParams, Calib
    win = synth.make_window("W7")
    newest = win.n - 1
    prm, aprm, calib = TraceParams.default(), ActivateParams.default(), Calib.from_K(win.K)
    ctx = lib.Context(win.w, win.h)
    dI0 = []
    for i in range(win.n):
        ctx.make_pyramid(i, win.images[i])
        dI0.append(orc.make_images(win.images[i])[0][0])
    # candidates of every older keyframe, traced against two later keyframes (device; checked against the oracle elsewhere)
    parts, hosts = [], []
    for hst in range(newest):
        u, v, _ = ih.candidates(win, hst, 400, seed=hst)
        pts = ctx.immature_init(prm, hst, u, v)
        for tgt in (hst + 1, newest) if hst + 1 != newest else (newest,):
            KRKi, Kt, aff = ih.host_to_frame(win.K, win.frames[hst]["camToWorld"], win.frames[tgt]["camToWorld"])
            pts = ctx.immature_trace(prm, tgt, pts, KRKi, Kt, aff)
        parts.append(pts)
        hosts.append(np.full(len(pts), hst, np.int32))
    cand, chost = np.concatenate(parts), np.concatenate(hosts)
    ctype = np.ones(len(cand), np.float32)
    act = win.points[::4]
    KRKi1, Kt1 = ih.level1_to_newest(win, newest)
    flagged = np.zeros(win.n, np.uint8)
    flagged[0] = 1
    pairs = ih.pair_tfms(win)
    w1, h1 = win.w // 2, win.h // 2
    min_dist = host.next_min_act_dist(2.0, len(act), 2000.0)
    assert min_dist == orc.next_min_act_dist(2.0, len(act), 2000.0)

    def flow(select, activate):
        dec, _ = select(w1, h1, newest, KRKi1, Kt1, act, min_dist, 3.0, cand, chost, ctype, flagged)
        todo = np.flatnonzero(dec == 1)
        res = activate(cand[todo], chost[todo])
        return dec, todo, res

    d_o, t_o, r_o = flow(orc.activate_select, lambda p, h: orc.immature_activate(aprm, calib, dI0, pairs, p, h))
    d_g, t_g, r_g = flow(host.activate_select, lambda p, h: ctx.immature_activate(aprm, calib, np.arange(win.n), pairs, p, h))
    assert np.array_equal(d_o, d_g) and np.array_equal(t_o, t_g)
    assert _act_same(r_g, r_o) is None, _act_same(r_g, r_o)
    activated = t_g[r_g["status"] == 1]
    assert len(activated) > 100 and (d_g == -1).sum() > 0 and (d_g == 0).sum() > 0
    # what FS/FullSystem.cpp:489-509 does with the results: new points, deletions, survivors
    gone = set(np.flatnonzero(d_g == -1)) | set(t_g[(r_g["status"] == -1) | ((r_g["status"] == 0) & (cand["lastTraceStatus"][t_g] == 1))])
    survivors = set(range(len(cand))) - gone - set(activated)
    assert len(survivors) + len(gone) + len(activated) == len(cand)
    ctx.close()


def test_resident_sets_equal_array_calls():
    """sos_immset: the immature points of several keyframes stay on the device over a sequence of frames (traceNewCoarse per frame with
    no host traffic); the records downloaded at the end -- and in between -- are bit-identical to the same sequence of
    sos_immature_trace_all calls on host arrays.  Lists are replaced / removed the way makeKeyFrame does."""
    from sos_slam_amd import lib
    win = synth.make_window("W7", extra_frames=2)
    prm = TraceParams.default()
    ctx = lib.Context(win.w, win.h)
    for i in range(win.n):
        ctx.make_pyramid(i, win.images[i])
    ctx.make_pyramid(win.n, win.extra_images[0])
    ctx.make_pyramid(win.n + 1, win.extra_images[1])
    keys = [3, 0, 5, 2]                      # host keyframes (arbitrary ids), list sizes that are not multiples of 64
    sizes = [700, 65, 1500, 1]
    arrays = {}
    st = lib.ImmatureSet(ctx)
    for k, sz in zip(keys, sizes):
        u, v, _ = ih.candidates(win, k, sz, seed=k)
        arrays[k] = ctx.immature_init(prm, k, u, v)
        st.put(k, arrays[k])
    assert [st.count(k) for k in keys] == sizes and st.count(77) == 0

    def tables(ks, c2w):
        t = [ih.host_to_frame(win.K, win.frames[k]["camToWorld"], c2w, frame_aff=(0.01, 0.7)) for k in ks]
        return np.stack([x[0].reshape(-1) for x in t]), np.stack([x[1] for x in t]), np.stack([x[2] for x in t])

    def array_trace(ks, slot, c2w):
        KR, KT, AF = tables(ks, c2w)
        allp = ctx.immature_trace_all(prm, slot, np.concatenate([arrays[k] for k in ks]),
                                      np.concatenate([np.full(len(arrays[k]), j, np.int32) for j, k in enumerate(ks)]), KR, KT, AF)
        o = 0
        for k in ks:
            arrays[k] = allp[o:o + len(arrays[k])]
            o += len(arrays[k])

    frames = [(win.n, win.extra_poses[0]), (win.n + 1, win.extra_poses[1]), (6, win.frames[6]["camToWorld"])]
    for slot, c2w in frames[:2]:
        st.trace(prm, slot, keys, *tables(keys, c2w))
        array_trace(keys, slot, c2w)
    for k in keys:
        assert _same(st.get(k), arrays[k]) is None, (k, _same(st.get(k), arrays[k]))
    # a keyframe decision: key 0 leaves, key 5 keeps a subset (activated / deleted points removed), a new keyframe 6 arrives
    st.put(0, arrays[0][:0])
    del arrays[0]
    arrays[5] = np.ascontiguousarray(arrays[5][::3])
    st.put(5, arrays[5])
    u, v, _ = ih.candidates(win, 6, 2000, seed=9)
    arrays[6] = ctx.immature_init(prm, 6, u, v)
    st.put(6, arrays[6])
    keys2 = [3, 5, 2, 6]
    assert st.count(0) == 0 and st.count(5) == len(arrays[5]) and st.count(6) == 2000
    # a key without points in the list is skipped; the same key twice is refused
    st.trace(prm, frames[2][0], [3, 5, 2, 0], *tables([3, 5, 2, 0][:3] + [3], frames[2][1]))
    array_trace([3, 5, 2], *frames[2])
    with pytest.raises(lib.SosError):
        st.trace(prm, frames[2][0], [3, 3], *tables([3, 3], frames[2][1]))
    st.trace(prm, win.n, keys2, *tables(keys2, win.extra_poses[0]))
    array_trace(keys2, win.n, win.extra_poses[0])
    seen = set()
    for k in keys2:
        assert _same(st.get(k), arrays[k]) is None, (k, _same(st.get(k), arrays[k]))
        seen |= set(int(s) for s in arrays[k]["lastTraceStatus"])
    assert len(seen) >= 3, seen
    st.close()
