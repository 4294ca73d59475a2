"""CPU micro-benchmark of the IMU branch of solveSystemF on the host (csrc/host/sos_imu.cpp): what the visual-inertial iteration of
`bench.py --imu` spends around the device's stitch.  No GPU needed -- the functions work on the stitched H / b the device hands
over, which are synthetic here (dense SPD, the sizes of W12), with a prior that is dense over the IMU states too.

    python tools/imu_solve_bench.py [W12] [reps]          SOS_TIMING_IMU=1 prints the phases of every call

Reports, per solve: the literal form (whole KKT system built and factorised), the kept factor with the prior compared by value
(sosf_imu_solve) and named (what the facade does: prior_id = its write counter), split into sosf_imu_solve_prepare (runs while the
accumulation is in flight) and sosf_imu_solve_finish (after the device's H / b); and the cost of rebuilding the factor (once per
optimize(): the linearisation points move between keyframes)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sos_slam_amd import host, synth
from sos_slam_amd.host import _p


def main():
    window = sys.argv[1] if len(sys.argv) > 1 else "W12"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    win = synth.make_window(window)
    S, cal, fr, keep = synth.make_imu_records(win, consistent=True)
    HMi, bMi = synth.expand_prior_imu(win)
    n = win.n
    d0 = 4 + 8 * n
    rng = np.random.default_rng(1)
    A = rng.normal(size=(d0, d0 + 4))
    H_top = np.ascontiguousarray(A @ A.T * 50 + np.eye(d0) * 200)
    B = rng.normal(size=(d0, 6))
    H_sc = np.ascontiguousarray(B @ B.T)
    b_top, b_sc, delta = rng.normal(size=d0) * 10, rng.normal(size=d0), rng.normal(size=d0) * 1e-3
    # a prior that is dense over the IMU states too (what frame marginalisations leave behind)
    # the prior's IMU part.  SOS_BENCH_PRIOR=dense: a rank-8 term over ALL expanded states (the worst case of rounds 1-3: no structure
    # at all).  Default: what frame marginalisations leave behind in a running chain (tools: /tmp probes of rolling chains):
    # the IMU states of a keyframe couple to their own keyframe, to the neighbours' IMU states and to every pose / the calibration
    dimI = HMi.shape[0]
    if os.environ.get("SOS_BENCH_PRIOR") == "dense":
        Mq = rng.normal(size=(dimI, 8))
        HM = np.ascontiguousarray(HMi + Mq @ Mq.T)
    else:
        HM = HMi.copy()
        vis = np.concatenate([np.arange(5)] + [5 + 29 * i + np.arange(8) for i in range(n)])
        for i in range(n - 1):
            rows = np.concatenate([vis, 5 + 29 * i + 8 + np.arange(21), 5 + 29 * (i + 1) + 8 + np.arange(21)])
            G = np.zeros((dimI, 6))
            G[rows] = rng.normal(size=(len(rows), 6))
            HM += G @ G.T
        HM = np.ascontiguousarray(HM)
    f = host.imu()
    L = f.L
    arr = f._frames(fr)
    x, ss, si = np.zeros(d0), C.c_double(0), np.zeros((n, 21))

    def two_calls(pid):
        t0 = time.perf_counter()
        L.sosf_imu_solve_prepare(C.byref(S), C.byref(cal), n, arr, _p(HM), _p(bMi), _p(delta), 1e-5, pid)
        t1 = time.perf_counter()
        L.sosf_imu_solve_finish(_p(H_top), _p(b_top), _p(H_sc), _p(b_sc), _p(x), C.byref(ss), _p(si))
        return t1 - t0, time.perf_counter() - t1

    out = {"window": window, "n": n, "dim_expanded": HMi.shape[0]}
    f.solve_mode(0)
    two_calls(0)
    x_lit = x.copy()
    ts = np.array([sum(two_calls(0)) for _ in range(reps)]) * 1e6
    out["literal_us"] = float(np.median(ts))
    f.solve_mode(1)
    for name, pid in (("kept_compared", 0), ("kept_named", 7)):
        two_calls(pid)
        ts = np.array([two_calls(pid) for _ in range(reps)]) * 1e6
        out[name + "_prepare_us"], out[name + "_finish_us"] = float(np.median(ts[:, 0])), float(np.median(ts[:, 1]))
    out["kept_vs_literal_x"] = float(np.abs(x - x_lit).max() / np.abs(x_lit).max())
    rb = []
    for k in range(min(reps, 30)):       # a new linearisation point every time: factor rebuilt (the literal form takes over after 3 in a row, so
        fr[1].state_imu_zero[20] += 1e-9   # every fourth call repeats the inputs)
        arr = f._frames(fr)
        for rep in range(2):
            t = sum(two_calls(7))
            if rep == 0 and k % 3 != 2:
                rb.append(t)
    out["rebuild_us"] = float(np.median(rb) * 1e6)
    # the scale not trapped yet (Jacobians at the current states: nothing can be kept): the same per-keyframe elimination built and used by
    # every call (round 5) against the dense KKT system of rounds 1-4
    trapped = cal.scale_trapped
    cal.scale_trapped = 0
    for name, mode in (("untrapped_dense", 0), ("untrapped_structured", 1)):
        f.solve_mode(mode)
        two_calls(0)
        xs = x.copy()
        out[name + "_us"] = float(np.median(np.array([sum(two_calls(0)) for _ in range(reps)]) * 1e6))
        if mode == 0:
            x_ud = xs
        else:
            out["untrapped_structured_vs_dense_x"] = float(np.abs(xs - x_ud).max() / np.abs(x_ud).max())
    cal.scale_trapped = trapped
    f.solve_mode(1)
    out["stats_kept_rebuilt_literal"] = f.solve_stats()
    print(out)


if __name__ == "__main__":
    main()
