"""CPU micro-benchmark of the IMU branch of solveSystemF on the host (sosf_imu_solve, csrc/host/sos_imu.cpp): the 0.5 ms the
visual-inertial iteration of `bench.py --imu` spends between the device's stitch and its back-substitution.  No GPU needed -- the
function works on the stitched H / b the device hands over, which are synthetic here (dense SPD, the sizes of W12).

    python tools/imu_solve_bench.py [W12] [reps]          SOS_TIMING_IMU=1 prints the phases of every call
"""
import sys
import time

import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sos_slam_amd import host, synth


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
    H_top = A @ A.T * 50 + np.eye(d0) * 200
    B = rng.normal(size=(d0, 6))
    H_sc = B @ B.T
    b_top, b_sc, delta = rng.normal(size=d0) * 10, rng.normal(size=d0), rng.normal(size=d0) * 1e-3
    # a prior that is dense over the IMU states too (what frame marginalisations leave behind)
    Mq = rng.normal(size=(HMi.shape[0], 8))
    HM = HMi + Mq @ Mq.T
    f = host.imu()
    x0 = f.solve(S, cal, fr, H_top, b_top, H_sc, b_sc, HM, bMi, delta)[0]
    for _ in range(10):
        f.solve(S, cal, fr, H_top, b_top, H_sc, b_sc, HM, bMi, delta)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        x = f.solve(S, cal, fr, H_top, b_top, H_sc, b_sc, HM, bMi, delta)[0]
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print({"window": window, "n": n, "dim_expanded": HMi.shape[0], "median_us": float(np.median(ts)), "min_us": float(ts.min()),
           "p90_us": float(np.percentile(ts, 90)), "x_norm": float(np.linalg.norm(x0)), "repeatable": bool(np.array_equal(x, x0))})


if __name__ == "__main__":
    main()
