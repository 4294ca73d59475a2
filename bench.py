#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X hot path (BASELINE.json: point-residuals/sec + GN-iter/sec
on a 12-KF x 4096-point window).

A "step" is one Gauss-Newton iteration of the sliding-window bundle adjustment, i.e. one loop body of
FullSystem::optimize (FS/FullSystemOptimize.cpp:358-413): accumulate A/L/SC on the device, fp64 stitch,
host solve, back-substitution, step + precalc, re-linearise every active residual, applyRes.  `value` =
point-residuals linearised per second through whole iterations, summed over all ranks.

  python bench.py --gpus 1 --steps K --warmup W           single GPU
  python bench.py --gpus N ...                            starts N ranks of itself under torch.distributed.run (one per GPU, RCCL)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   the same with the launcher outside

Multi-GPU: every rank holds the same keyframes and a shard of the points; the packed fp32 H/b
accumulators are all-reduced over RCCL once per iteration (by the library itself, on its stream, inside the prefetched
accumulate chain: sos_ba_set_comm), every rank then runs the identical fp64 stitch + solve.
  --scaling weak   (default) every rank its own W12 shard of 4096 points: per-GPU work fixed
  --scaling strong BASELINE.json config 5: ONE window (default W16: 16 KF x 8192 points) whose points are sharded over
                   the ranks (contiguous slices of the allPoints order balanced by residual count)
The timed region is bracketed by barrier + torch.cuda.synchronize and the MAX over ranks is reported.  A Gauss-Newton
iteration takes under 0.1 ms, so every step is timed as the mean of `--inner` consecutive iterations (default: enough for
a timed region of about 1 s); ms_per_step and value are per single iteration.

Output: ONE contract line is what a run is judged by, and it is the LAST JSON line of stdout.  The default N = 1 run prints the line
twice: first with `"partial": true` as soon as the headline loop, the roofline and the CPU baseline are measured (so that a side
measurement that hangs or is killed by the caller's clock cannot cost the line), then again, complete (`"partial": false`), with the side
entries (tracker, keyframe, visual_inertial, device_solve, resident_loop).  `--no-sides` prints the line once.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LINEARIZE_BYTES_PER_RESIDUAL = 776  # 80 point + 384 gather + 296 J + 16 state
TOP_BYTES_PER_RESIDUAL = 312        # top accumulate = J read 296 + indices/flags 16
# what the FUSED kernel has to move at least (the Jacobian tile never leaves the chip): 80 point copies + 384 texels +
# 16 state + 32 JpJdF + 32 point terms + 12 centre + 12 tile block sums (384 B per 32 residuals) ~ 568 B
FUSED_MIN_BYTES_PER_RESIDUAL = 568
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--window", default=None, help="W7 / W12 / W16 (default: W12, or W16 with --scaling strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--inner", type=int, default=0, help="GN iterations per timed step (0 = enough for a ~1 s region)")
    ap.add_argument("--resident", action="store_true",
                    help="device-resident Gauss-Newton loop (solve / step / precalc kernels; measured slower than the host solve)")
    ap.add_argument("--imu", action="store_true",
                    help="visual-inertial window (BASELINE.json configs 2-3): the IMU / spline block of solveSystemF "
                         "(OB/EnergyFunctional.cpp:1053-1171) sits between stitch and solve on the host")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--side", choices=("imu", "gnsolve", "tracker", "keyframe"), default=None, help=argparse.SUPPRESS)   # one side measurement as its own process
    ap.add_argument("--no-sides", action="store_true", help="headline loop only: no keyframe / visual-inertial / variants entries")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--variants", action="store_true", help="also time the opt-in launch variants, each in a process of its own (not in the default run)")
    a = ap.parse_args()
    if a.window is None:
        a.window = "W16" if a.scaling == "strong" else "W12"
    if a.inner <= 0:
        a.inner = max(1, -(-14000 // max(a.steps, 1)))   # ~0.07 ms per iteration: steps x inner >= 14000 iterations, a timed region of ~1 s
    return a


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks of this script under torch.distributed.run
    (one per GPU, RCCL) and let rank 0's line through.  Fails loudly when the node has fewer than N GPUs."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("SOS_BENCH_SINGLE_GPU") != "1" and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node has {have} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline(win, seconds):
    """The C restatement of the reference's CPU path (kind "port"), timing build: -O3 -march=native like the reference's
    own CMakeLists.txt:26, compiled on this box; the reference's 6-thread fork/join pool with its chunking
    (util/IndexThreadReduce.h) and, beside it, one thread.  A bounded number of GN iterations of the same window."""
    from oracle import oracle as orc
    cores = min(6, os.cpu_count() or 1)

    def run(nthreads, budget):
        ow = orc.window_from_synth(win, fast=True)
        ow.reset_oob()
        th = np.array([ow.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
        ow.linearize(th, nthreads=nthreads)
        ow.apply_res()
        ow.gn_iteration(0, nthreads=nthreads)  # warm-up
        t0 = time.perf_counter()
        it = 0
        while True:
            ow.gn_iteration(it + 1, nthreads=nthreads)
            it += 1
            if time.perf_counter() - t0 > budget or it >= 400:
                break
        dt = time.perf_counter() - t0
        ow.close()
        return win.R * it / dt, it / dt, it

    v6, g6, it6 = run(cores, seconds * 0.55)
    v1, g1, it1 = run(1, seconds * 0.35)

    def phases(nthreads, reps=3):   # where the threads help and where they do not (ms per call)
        ow = orc.window_from_synth(win, fast=True)
        ow.reset_oob()
        th = np.array([ow.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
        ow.linearize(th, nthreads=nthreads)
        ow.apply_res()
        out = {}
        for name, fn in (("linearize", lambda: ow.linearize(th, nthreads=nthreads)), ("accumulate_stitch", lambda: ow.accumulate(nthreads=nthreads)),
                         ("gn_iteration", lambda: ow.gn_iteration(0, nthreads=nthreads))):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            out[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
        ow.close()
        return out

    ph = {"threads_1": phases(1), f"threads_{cores}": phases(cores)}
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:  # noqa: BLE001
        model = "unknown"
    return {"value": v6, "unit": "point-residuals/s", "cores": cores, "kind": "port",
            "sample": f"{it6} Gauss-Newton iterations of the same {win.name} window ({win.R} residuals) with the "
                      f"reference's {cores}-thread pool structure; 1 thread: {it1} iterations",
            "build": "gcc -O3 -march=native (oracle/Makefile: fast), built on this host", "host_cpu": model,
            "host_logical_cpus": os.cpu_count(),
            "gn_iter_per_s": g6, "value_1thread": v1, "gn_iter_per_s_1thread": g1, "thread_scaling": g6 / g1, "phases": ph}


GN_LOOP_NAMES = {0: "host solve (blocked LDL^T) and host-side step, device everything else",
                 1: "host solve (blocked LDL^T), device everything else incl. the step (poses, precalc, deltas from x)",
                 2: "device-resident (k_gn_solve)", 3: "energy-checked steps (host solve)"}


def main():
    args = parse()
    if args.side == "imu":   # child of side_process(): the measurement alone, one JSON line
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.write(json_fd, (json.dumps(imu_timing(args.window, int(os.environ.get("LOCAL_RANK", "0")))) + "\n").encode())
        return
    if args.side in ("tracker", "keyframe"):   # the per-frame / per-keyframe side measurements, each on its own
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        json_fd = os.dup(1)
        os.dup2(2, 1)
        fn = tracker_timing if args.side == "tracker" else keyframe_timing
        os.write(json_fd, (json.dumps(fn(args.window, int(os.environ.get("LOCAL_RANK", "0")))) + "\n").encode())
        return
    if args.side == "gnsolve":   # k_gn_solve alone: it has never run on an MI355X, so not in the process that owns the line
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        from sos_slam_amd import synth
        json_fd = os.dup(1)
        os.dup2(2, 1)
        n = synth.WINDOWS[args.window]["n"]
        os.write(json_fd, (json.dumps(device_solve_timing(n, int(os.environ.get("LOCAL_RANK", "0")))) + "\n").encode())
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit(f"bench.py --gpus {args.gpus} under a launcher with WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to stdout whenever a
    # communicator has existed -- every N > 1 run): file descriptor 1 is pointed at stderr for the life of the process and the line
    # goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # SOS_BENCH_SINGLE_GPU=1: a rehearsal of the N > 1 flow on a box with one GPU -- every rank on device 0, gloo instead of RCCL (which
    # refuses two ranks on one device), the exchange through the torch.distributed hooks on host copies.  Timings mean nothing in this
    # mode; it exists to show that the ranks issue matching collectives and that rank 0 prints the line.
    single_gpu = os.environ.get("SOS_BENCH_SINGLE_GPU") == "1"
    dev = 0 if single_gpu else local_rank
    tdev = "cpu" if single_gpu else "cuda"
    torch.cuda.set_device(dev)
    dist = None
    force_dist = world == 1 and os.environ.get("SOS_BENCH_FORCE_DIST") == "1"  # exercise the exchange path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo" if single_gpu else "nccl", rank=rank, world_size=world)

    from sos_slam_amd import host, lib, synth
    from sos_slam_amd import distributed as sdist

    if args.scaling == "strong":
        # config 5: ONE window, its points sharded over the ranks (all residuals of a point on one rank)
        full = synth.make_window(args.window)
        win = synth.take_shard(full, synth.shard_points(full, rank, world)) if world > 1 else full
    else:
        # every rank: same frames / images (seed), its own point set (point_seed): per-GPU work fixed
        win = synth.make_window(args.window, point_seed=synth.SEED + 1000 * rank if world > 1 else None)
    sysm = host.System.from_window(win, device=dev)
    comm = None
    exchange = "enqueued by the library on its stream"
    if dist is not None:
        if single_gpu:
            sdist.attach(sysm, dist, torch, device_pointers=True)
            exchange = "gloo on host copies (single-GPU rehearsal)"
        elif os.environ.get("SOS_BENCH_HOOKS") == "1":  # callback-based exchange through torch.distributed (slower)
            sdist.attach(sysm, dist, torch)
            exchange = "rccl via torch.distributed hooks"
        else:  # RCCL collectives enqueued by the library itself on its stream
            err = None
            try:
                comm = sdist.NativeComm(dist, torch, dev)
                comm.attach(sysm)
            except Exception as e:  # noqa: BLE001 -- every rank has to take the same exchange path
                err, comm = e, None
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=tdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:  # some rank could not set up the library's communicator: exchange through torch.distributed
                if rank == 0:
                    print(f"[bench] library-owned RCCL communicator unavailable ({err}); using the torch.distributed hooks", file=sys.stderr)
                if comm is not None:
                    comm.close()
                    comm = None
                sdist.attach(sysm, dist, torch)
                exchange = "rccl via torch.distributed hooks"
    if args.resident:
        sysm.set_resident(True)
    imu_keep = None
    if args.imu:   # IMU records of the window's keyframes + the prior in the expanded dimension (sosf_set_imu)
        S_imu, cal_imu, fr_imu, keep_imu = synth.make_imu_records(win, consistent=True)
        HMi, bMi = synth.expand_prior_imu(win)
        sysm.set_imu(S_imu, cal_imu, fr_imu, HMi, bMi)
        imu_keep = (S_imu, cal_imu, fr_imu, keep_imu, HMi, bMi)
    sysm.prepare()
    sysm.set_pipeline(True)  # the loop below never stops on `canbreak`: every step may prefetch the next accumulate
    R_local = win.R

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    inner = args.inner
    for i in range(args.warmup * inner):
        sysm.gn_iteration(i)
    host.timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps * inner):     # K steps, each the mean of `inner` consecutive Gauss-Newton iterations
        sysm.gn_iteration(args.warmup * inner + i)
    barrier()
    dt = time.perf_counter() - t0
    phases = host.timing()
    loop_mode = sysm.loop_mode()      # what the iterations actually ran as, not what was asked for
    res_in = sysm.stats()             # residuals that were IN in the last solve of the timed region (all ranks: the counts ride in the exchange)
    last_x = np.asarray(sysm.lastX(), dtype=np.float64)   # the last solve's step: identical on every rank, and (strong mode) for every N up to summation order
    iters = args.steps * inner
    R_total = R_local
    if dist is not None:
        t = torch.tensor([dt, float(R_local)], dtype=torch.float64, device=tdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0].item())
        R_total = int(t[1].item())

    out = None
    exchange_us = None
    if dist is not None and comm is not None:   # the collective alone (every rank has to take part)
        import ctypes as C
        ms_x = C.c_float(0)
        lib.load().sos_ba_time_kernel(L_host_ba(sysm), b"exchange", None, 200, C.byref(ms_x))
        exchange_us = ms_x.value * 1e3
    if rank == 0:
        ms_per_step = dt / iters * 1e3
        value = R_total * iters / dt
        # ---- roofline of the dominant kernel (linearize): HIP events on the library's stream
        L = lib.load()
        import ctypes as C
        ms = C.c_float(0)
        th = np.array([sysm.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32)
        # the kernel the timed loop runs: PointFrameResidual::linearize + applyRes + AccumulatedTopHessian::addPoint<0>
        # fused (the Jacobian tile is reduced in LDS and never written) = F1 + F3 + F7 of SURVEY.md 8(a)
        L.sos_ba_time_kernel(L_host_ba(sysm), b"linearize_fused", th.ctypes.data_as(C.c_void_p), 300, C.byref(ms))
        fused_ms = ms.value
        fused_bytes = LINEARIZE_BYTES_PER_RESIDUAL + TOP_BYTES_PER_RESIDUAL
        achieved = R_local * fused_bytes / (fused_ms * 1e-3) / 1e9
        # the stand-alone linearize (J stored; final linearizeAll(true), marginalisation) for comparison
        L.sos_ba_time_kernel(L_host_ba(sysm), b"linearize", th.ctypes.data_as(C.c_void_p), 300, C.byref(ms))
        lin_ms = ms.value
        kern = {"linearize_fused_us": round(fused_ms * 1e3, 2)}
        achieved_min = R_local * FUSED_MIN_BYTES_PER_RESIDUAL / (fused_ms * 1e-3) / 1e9
        # yardsticks under the same launch conditions: an empty kernel of the linearisation's grid / block / LDS size, and a
        # streaming read of as many bytes as the fused kernel's algorithmic traffic
        L.sos_ba_time_kernel(L_host_ba(sysm), b"lin_floor", th.ctypes.data_as(C.c_void_p), 300, C.byref(ms))
        floor_ms = ms.value
        L.sos_ba_time_kernel(L_host_ba(sysm), b"stream_equal", th.ctypes.data_as(C.c_void_p), 300, C.byref(ms))
        stream_ms = ms.value
        L.sos_ba_time_kernel(L_host_ba(sysm), b"stream_large", th.ctypes.data_as(C.c_void_p), 50, C.byref(ms))
        large_ms = ms.value   # coalesced read of a 1 GiB buffer: the chip's streaming bandwidth without launch effects
        for name in ("apply_res", "top_accumulate", "sc_accumulate", "sc_gram_prep", "reduce", "stitch", "resub_fused", "sc_gram_abs",
                     "abs_reduce_stitch1", "abs_stitch2"):
            L.sos_ba_time_kernel(L_host_ba(sysm), name.encode(), th.ctypes.data_as(C.c_void_p), 200, C.byref(ms))
            kern[name + "_us"] = round(ms.value * 1e3, 2)
        out = {
            "metric": "point-residuals/sec", "value": value, "unit": "point-residuals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.window}: {win.n} KF x {win.P} points per GPU, {win.w}x{win.h}, "
                                    f"{R_local} point-residuals per GPU" if args.scaling == "weak" or world == 1 else
                                    f"{args.window} (BASELINE.json config 5): ONE window of {win.n} KF, {R_total} point-residuals, "
                                    f"{win.w}x{win.h}, points sharded over {world} GPUs ({win.P} points / {R_local} residuals on rank 0)") +
                                   ("; visual-inertial: IMU / spline factors of every keyframe assembled and eliminated on the host inside the solve" if args.imu else "") +
                                   "; step = one Gauss-Newton iteration (accumulate A/L/SC, fp64 stitch, solve, back-substitute, "
                                   f"step, re-linearise, applyRes), timed as the mean of {inner} consecutive iterations",
                       "gn_loop": GN_LOOP_NAMES.get(loop_mode, f"mode {loop_mode}") + ("" if (loop_mode == 2) == bool(args.resident) else
                                                                                      " -- --resident was requested but the facade did not take it (communicator / hooks attached)"),
                       "window": args.window, "keyframes": win.n, "points_per_gpu": win.P, "inner_repeat": inner,
                       "timed_region_ms": round(dt * 1e3, 2),
                       "residuals_per_gpu": R_local, "residuals_total": R_total,
                       "resInA_last_iteration": res_in["resInA"], "resInL_last_iteration": res_in["resInL"],
                       "parallelism": ("single GPU" if dist is None else
                                       f"{world} ranks, points sharded, frames replicated; one RCCL all-reduce of the "
                                       "packed fp32 accumulator + one all-gather of newest-frame energies per "
                                       "iteration, " + exchange)},
            "gn_iter_per_s": iters / dt,
            "imu": bool(args.imu),
            "last_step_l2": float(np.linalg.norm(last_x)), "last_step_head": [float(v) for v in last_x[:6]],
            "exchange_allreduce_us": exchange_us,
            "exchange": exchange_info(sysm, win, world),
            "linearize_point_residuals_per_s": R_local / (lin_ms * 1e-3),
            "kernels_us": dict(linearize_us=round(lin_ms * 1e3, 2), **kern),
            "host_phases_us": {k: round(v / iters * 1e6, 1) for k, v in zip(
                ("gn_accumulate_wait", "assemble", "ldlt", "step_and_precalc", "precalc", "gn_step_call_and_post", "post",
                 "backup"), phases)},
            "roofline": {"kernel": "k_linearize (fused: linearize + applyRes + top-Hessian tile sums, J kept in LDS)",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, **pmc_traffic(args.window),
                         "bytes_per_residual": fused_bytes,
                         "min_bytes_per_residual": FUSED_MIN_BYTES_PER_RESIDUAL, "achieved_min_bytes": achieved_min,
                         "frac_min_bytes": achieved_min / HBM_PEAK_GBS,
                         "large_buffer_stream_gbs": (1 << 30) / (large_ms * 1e-3) / 1e9,
                         "frac_of_large_buffer_stream": achieved / ((1 << 30) / (large_ms * 1e-3) / 1e9),
                         "bytes_per_residual_parts": {"linearize": LINEARIZE_BYTES_PER_RESIDUAL, "top_accumulate": TOP_BYTES_PER_RESIDUAL},
                         "avg_launch_us": fused_ms * 1e3,
                         "launch_floor_us": floor_ms * 1e3, "equal_bytes_stream_read_us": stream_ms * 1e3,
                         "measured_stream_peak": R_local * fused_bytes / (stream_ms * 1e-3) / 1e9,
                         "frac_of_measured_stream": stream_ms / fused_ms,
                         "note": "launch_floor_us = empty kernel with the same grid, block and LDS size; "
                                 "equal_bytes_stream_read_us = coalesced 16 B/lane read of residuals x bytes_per_residual "
                                 "bytes (both back to back on the same stream, like avg_launch_us); measured_stream_peak = "
                                 "those bytes / that time in GB/s, frac_of_measured_stream = achieved / measured_stream_peak "
                                 "(the north star's 'measured HBM roofline'; `peak` / `frac` use the nominal 8 TB/s); "
                                 "achieved / frac price the SURVEY 8(d) bytes of the functions the kernel replaces (incl. the "
                                 "J write + J read the fusion removed), achieved_min_bytes / frac_min_bytes only what the fused "
                                 "kernel must move; large_buffer_stream_gbs = coalesced read of 1 GiB in the same run",
                         "unfused_linearize": {"avg_launch_us": lin_ms * 1e3, "bytes_per_residual": LINEARIZE_BYTES_PER_RESIDUAL,
                                               "achieved": R_local * LINEARIZE_BYTES_PER_RESIDUAL / (lin_ms * 1e-3) / 1e9,
                                               "frac": R_local * LINEARIZE_BYTES_PER_RESIDUAL / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
    if comm is not None:
        comm.close()
    sysm.close()
    if rank == 0:
        if world > 1:
            out["roofline"]["note"] += "; N > 1: rank 0's kernel on its own shard"
        out["cpu_baseline"] = None   # timed on rank 0 at N = 1 only
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(win, args.cpu_seconds)
            out["speedup_vs_cpu_gn_iter"] = out["gn_iter_per_s"] / out["cpu_baseline"]["gn_iter_per_s"]
            # (the thread scaling of the CPU port swings by 1.5x between boxes of the pool: the 1-thread figure is the stable denominator)
            out["speedup_vs_cpu_gn_iter_1thread"] = out["gn_iter_per_s"] / out["cpu_baseline"]["gn_iter_per_s_1thread"]
        sides = world == 1 and not args.imu and not args.no_sides
        if sides:
            # The headline measurements + roofline + cpu_baseline are complete: the line goes out NOW, and again, enriched with the side
            # measurements, at the end -- so whatever a side measurement does (a hang, a kill by the caller's clock) cannot cost the
            # line.  A reader that takes the first or the last JSON line of stdout gets a complete contract line either way.
            os.write(json_fd, (json.dumps(dict(out, partial=True)) + "\n").encode())
            # side measurements: a failure in one of them must not cost the headline line
            # (each in a process of its own: most of what they run has not been on an MI355X since round 3 -- a fault of the native
            # code must cost the entry, not the process that owns the line)
            out["tracker"] = side_process("tracker", args.window)
            out["keyframe"] = side_process("keyframe", args.window)
            if "optimize_ms" in out["keyframe"]:
                out["optimize_ms"] = out["keyframe"]["optimize_ms"]
                out["keyframe_ms"] = out["keyframe"]["keyframe_ms"]
            out["visual_inertial"] = side_process("imu", args.window)
            out["device_solve"] = side_process("gnsolve", args.window, timeout=120)
            # the device-resident loop (k_gn_solve in the chain, no host between two kernels): opt-in in the library until it has been
            # timed; measured here in a process of its own (no device-side waits in it: the host polls a mapped slot with a timeout)
            try:
                out["resident_loop"] = variant_timing(args.window, only=("resident",)).get("resident")
            except Exception as e:  # noqa: BLE001
                out["resident_loop"] = {"error": repr(e)}
            if args.variants:
                # opt-in: the same loop under each remaining switch, each in a process of its own
                try:
                    out["variants"] = variant_timing(args.window)
                except Exception as e:  # noqa: BLE001
                    out["variants"] = {"error": repr(e)}
        out["partial"] = False
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def side_process(what, window, timeout=240, env=None):
    """A side measurement in a process of its own: whatever happens to it (an exception, a crash of the native code, a hang) costs
    its entry of the line, not the line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--side", what, "--window", window]
    try:
        e = dict(os.environ)
        e.update(env or {})
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout, env=e)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"exit code {r.returncode}"}
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


# Opt-in paths that have not been measured on an MI355X from inside the build (GPU access was closed during rounds 3-4): the same loop
# under each switch, in a process of its own, reported BESIDE the headline -- never instead of it.  `last_step_l2` / `resInA` show that
# a variant computed the same iteration (the absolute-coordinate path differs in the last digits by design).  Order: the switches
# that only change plain launches first; the ones whose kernels WAIT on the device (grid barriers, the mailbox of the pre-launched step --
# bounded spins, csrc/sos_ba.hip) last, and all of them after every measurement this process takes itself (main()).
VARIANTS = (
    ("abs_schur", {"SOS_ABS_SC": "1"}, []),                                         # one Gram per chunk instead of the n^3 relative blocks + 2-stage stitch
    ("abs_schur_signal_in_kernel", {"SOS_ABS_SC": "1", "SOS_STITCH_SIGNAL_IN_KERNEL": "1"}, []),
    ("stitch_signal_in_kernel", {"SOS_STITCH_SIGNAL_IN_KERNEL": "1"}, []),          # the stitch's last kernel raises the host flag itself (no k_publish)
    ("resident", {}, ["--resident"]),                                                 # solve on the device (k_gn_solve), the host out of the loop
    ("resident_abs_schur", {"SOS_ABS_SC": "1"}, ["--resident"]),
    ("lin_xcd_ranges", {"SOS_LIN_XCD": "1"}, []),
    # (the `visual_inertial_overlap` entry of round 4 -- SOS_IMU_OVERLAP=1 on the --imu loop -- was retired with round 5's split into side
    # processes: the knob is a host-side order selector whose A/B is `--imu` with and without the variable, two command lines, not a variant
    # of the visual loop)                                     # k_linearize2: a contiguous tile range per XCD (judged by FETCH_SIZE, tools/pmc_probe.py)
)


def variant_timing(window, timeout=90, budget=300.0, only=None):
    import subprocess
    out = {}
    t_begin = time.perf_counter()
    for name, env, extra in VARIANTS:
        if only is not None and name not in only:
            continue
        if time.perf_counter() - t_begin > budget:   # the default run has to end within minutes whatever a variant does
            out[name] = {"error": "skipped: the variants' time budget of %.0f s was spent" % budget}
            continue
        e = dict(os.environ)
        e.update(env)
        cmd = [sys.executable, os.path.abspath(__file__), "--window", window, "--steps", "20", "--warmup", "3", "--inner", "100",
               "--no-cpu-baseline", "--no-sides"] + extra
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout, env=e)
            lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"exit code {r.returncode}"}
                continue
            d = json.loads(lines[-1])
            k = d.get("kernels_us", {})
            out[name] = {"env": env, "args": extra, "us_per_iteration": round(d["ms_per_step"] * 1e3, 2), "gn_loop": d["config"]["gn_loop"][:60],
                         "resInA": d["config"]["resInA_last_iteration"], "last_step_l2": d["last_step_l2"],
                         "host_phases_us": d.get("host_phases_us"),
                         "kernels_us": {q: k.get(q) for q in ("sc_gram_prep_us", "reduce_us", "stitch_us", "sc_gram_abs_us", "abs_reduce_stitch1_us",
                                                             "abs_stitch2_us", "linearize_fused_us", "resub_fused_us")}}
        except Exception as ex:  # noqa: BLE001
            out[name] = {"error": repr(ex)[:200]}
    return out


def keyframe_timing(window, device):
    """What ONE keyframe pays for the backend, as opposed to the steady-state iteration above (FS/FullSystem.cpp:853-927):
    optimize() on a freshly packed window -- pack + upload, first linearisation, the Gauss-Newton iterations until the step test
    passes, setEvalPT, linearizeAll(true) -- then removeOutliers + setCoarseTrackingRef, flagPointsForRemoval with the point
    marginalisation, and marginalizeFrame of the flagged keyframes.  Medians over fresh systems (the graph changes); the handles'
    buffers are warm (one optimize() before the timed one, as in a running system)."""
    from sos_slam_amd import host, synth
    win = synth.make_window(window)
    rows = []
    for _ in range(4):
        sysm = host.System.from_window(win, device=device)
        ht = host.HostTracker(sysm)
        sysm.optimize(6)
        sysm.invalidate_pack()
        t0 = time.perf_counter()
        _, its = sysm.optimize(6)
        t1 = time.perf_counter()
        sysm.remove_outliers()
        ht.set_ref()
        t2 = time.perf_counter()
        sysm.flag_frames_for_marginalization(np.zeros(win.n, np.int32))
        nm, nd = sysm.flag_points_for_removal()
        t3 = time.perf_counter()
        ids, _ = sysm.marginalize_flagged_frames(cap=win.n)
        t4 = time.perf_counter()
        sysm.set_min_opt_iterations(6)
        sysm.invalidate_pack()
        t5 = time.perf_counter()
        _, its6 = sysm.optimize(6)
        t6 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t6 - t5) * 1e3, its, nm, nd, len(ids), its6))
        sysm.close()
    r = np.array([x[:5] for x in rows[1:]])
    med = np.median(r, axis=0)
    try:
        act_sel = activation_select_timing(win)
    except Exception as e:   # noqa: BLE001 -- a side figure must not cost the entry
        act_sel = {"error": repr(e)[:200]}
    try:
        act_dev = activation_device_timing(win, device)
    except Exception as e:   # noqa: BLE001
        act_dev = {"error": repr(e)[:200]}
    # the whole backend part of makeKeyFrame (FS/FullSystem.cpp:783-931): activation (host selection + device optimisation) + optimize() +
    # removeOutliers / tracking reference + point and frame marginalisation + makeNewTraces
    act_ms = sum(d.get(k, 0.0) for d, k in ((act_sel, "activate_select_ms"), (act_dev, "immature_activate_ms"), (act_dev, "make_new_traces_ms")))
    return {"activation_select": act_sel, "activation_device": act_dev, "keyframe_with_activation_ms": float(med[:4].sum()) + float(act_ms),
            "optimize_ms": float(med[0]), "optimize_iterations": int(rows[-1][5]),
            "host_threads": int(__import__("sos_slam_amd.host", fromlist=["load"]).load().sosf_get_host_threads()),   # graph walks of the pack / final linearisation
            "remove_outliers_set_tracking_ref_ms": float(med[1]), "flag_points_marginalize_points_ms": float(med[2]),
            "marginalize_frames_ms": float(med[3]), "keyframe_ms": float(med[:4].sum()),
            "points_marginalized": int(rows[-1][6]), "points_dropped": int(rows[-1][7]), "frames_marginalized": int(rows[-1][8]),
            "optimize_6_iterations_ms": float(med[4]), "optimize_6_iterations_ran": int(rows[-1][9]),
            "note": "optimize_ms: pack + first linearisation + iterations until the step test passes + final linearizeAll(true); "
                    "optimize_6_iterations_ms: the same on the window after the marginalisations with setting_minOptIterations = 6; "
                    "keyframe_ms = the four stages above"}


def activation_device_timing(win, device, n_activate=800, density=1500.0, reps=7):
    """The device half of FullSystem::activatePointsMT (FS/FullSystem.cpp:472-531: optimizeImmaturePoint of the chosen candidates against
    every keyframe of the window, sos_immature_activate) and FullSystem::makeNewTraces on the new keyframe (FS/FullSystem.cpp:1071-1097:
    PixelSelector::makeMaps at setting_desiredImmatureDensity + the ImmaturePoint constructors, sos_pixsel_make_maps / _list +
    sos_immature_init).  The candidates stand on the window's own points (their inverse-depth interval around the point's value, as
    earlier traces leave it), `n_activate` of them -- what the selection above picks per keyframe at 2000 desired points."""
    from sos_slam_amd import lib
    from sos_slam_amd.records import PAIR_TFM_DTYPE, ActivateParams, Calib, PixselParams, TraceParams
    n = win.n
    ctx = lib.Context(win.w, win.h, device=device)
    try:
        for i in range(n):
            ctx.make_pyramid(i, win.images[i])
        tprm, aprm, calib = TraceParams.default(), ActivateParams.default(), Calib.from_K(win.K)

        def inv(T):
            R, t = T[:9].reshape(3, 3), T[9:]
            return np.concatenate([R.T.reshape(-1), -(R.T @ t)])

        def mul(A, B):
            Ra, ta, Rb, tb = A[:9].reshape(3, 3), A[9:], B[:9].reshape(3, 3), B[9:]
            return np.concatenate([(Ra @ Rb).reshape(-1), Ra @ tb + ta])

        pairs = np.zeros(n * n, dtype=PAIR_TFM_DTYPE)   # PRE_RTll / PRE_tTll / PRE_aff_mode of host -> target (FS/HessianBlocks.cpp:431-461)
        for h in range(n):
            for t in range(n):
                T = mul(inv(win.frames[t]["camToWorld"]), win.frames[h]["camToWorld"])
                o = pairs[h + n * t]
                o["R"] = T[:9].astype(np.float32); o["t"] = T[9:].astype(np.float32); o["aff"] = (1.0, 0.0)
        rng = np.random.default_rng(5)
        parts, hosts = [], []
        per_host = max(1, n_activate // max(1, n - 1))
        for h in range(n - 1):   # (the newest keyframe has no immature points yet)
            sel = np.flatnonzero(win.points["host"] == h)[:per_host]
            if len(sel) == 0:
                continue
            u = np.rint(win.points["u"][sel]).astype(np.int32)
            v = np.rint(win.points["v"][sel]).astype(np.int32)
            pts = ctx.immature_init(tprm, h, u, v)
            mid = win.points["idepth_scaled"][sel].astype(np.float64) * (1 + rng.uniform(-0.05, 0.05, len(sel)))
            wdt = rng.uniform(0.05, 0.3, len(sel))
            pts["idepth_min"] = (mid * (1 - wdt)).astype(np.float32)
            pts["idepth_max"] = (mid * (1 + wdt)).astype(np.float32)
            parts.append(pts)
            hosts.append(np.full(len(pts), h, np.int32))
        pts, hosts = np.concatenate(parts), np.concatenate(hosts)
        slots = np.arange(n)
        ctx.immature_activate(aprm, calib, slots, pairs, pts, hosts)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = ctx.immature_activate(aprm, calib, slots, pairs, pts, hosts)
            ts.append(time.perf_counter() - t0)
        act_ms = float(np.median(ts)) * 1e3
        pattern = np.random.default_rng(7).integers(0, 256, win.w * win.h).astype(np.uint8)
        sel = lib.PixelSelector(ctx, PixselParams.default(), pattern)
        newest = n - 1
        sel.make_maps(newest, density, want_map=False)
        ts, cnt = [], 0
        for _ in range(reps):
            t0 = time.perf_counter()
            sel.make_maps(newest, density, want_map=False)
            u, v, _t = sel.list(pattern_padding=2)
            new_pts = ctx.immature_init(tprm, newest, u, v)
            ts.append(time.perf_counter() - t0)
            cnt = len(new_pts)
        sel.close()
        return {"immature_activate_ms": act_ms, "candidates_optimised": int(len(pts)), "activated": int(np.sum(res["status"] == 1)),
                "make_new_traces_ms": float(np.median(ts)) * 1e3, "new_immature_points": int(cnt)}
    finally:
        ctx.close()


def activation_select_timing(win, n_cand_per_point=4, reps=9):
    """The host half of FullSystem::activatePointsMT (FS/FullSystem.cpp:375-470) at the window's size: distance map of the active points
    in the newest keyframe at level 1 + the ordered candidate loop (sosf_activate_select; the device half is sos_immature_activate).
    The window carries no immature points, so the candidates are synthetic: n_cand_per_point per active point, uniformly over the
    image, traced states as a running sequence leaves them.  Needs no GPU."""
    from sos_slam_amd import host
    from sos_slam_amd.records import IMMATURE_DTYPE
    n, newest = win.n, win.n - 1
    w1, h1 = win.w // 2, win.h // 2
    fx, fy, cx, cy = [np.float32(v) for v in win.K]
    K1 = np.array([[fx * np.float32(0.5), 0, (cx + np.float32(0.5)) / np.float32(2) - np.float32(0.5)],
                   [0, fy * np.float32(0.5), (cy + np.float32(0.5)) / np.float32(2) - np.float32(0.5)], [0, 0, 1]], dtype=np.float32)
    Ki0 = np.linalg.inv(np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)).astype(np.float32)
    Tn = win.frames[newest]["camToWorld"]
    Rn, tn = Tn[:9].reshape(3, 3), Tn[9:]
    KRKi, Kt = [], []
    for f in range(n):
        Tf = win.frames[f]["camToWorld"]
        R = (Rn.T @ Tf[:9].reshape(3, 3)).astype(np.float32)
        t = (Rn.T @ (Tf[9:] - tn)).astype(np.float32)
        KRKi.append(((K1 @ R).astype(np.float32) @ Ki0).astype(np.float32).reshape(-1))
        Kt.append((K1 @ t).astype(np.float32))
    KRKi, Kt = np.stack(KRKi), np.stack(Kt)
    act = np.zeros(len(win.points), dtype=[("u", "f4"), ("v", "f4"), ("idepth_scaled", "f4"), ("host", "i4")])
    for k in ("u", "v", "idepth_scaled", "host"):
        act[k] = win.points[k]
    rng = np.random.default_rng(5)
    nc = n_cand_per_point * len(win.points)
    cand = np.zeros(nc, dtype=IMMATURE_DTYPE)
    hosts = np.sort(rng.integers(0, newest, nc)).astype(np.int32)
    cand["u"], cand["v"] = rng.integers(4, win.w - 4, nc), rng.integers(4, win.h - 4, nc)
    idm = rng.uniform(0.1, 1.5, nc)
    cand["idepth_min"], cand["idepth_max"] = idm * 0.9, idm * 1.1
    cand["quality"], cand["lastTracePixelInterval"] = rng.uniform(1.0, 8.0, nc), rng.uniform(0.0, 12.0, nc)
    cand["lastTraceStatus"] = rng.choice([0, 0, 0, 3, 4, 1, 2, 5], nc)
    ctype = rng.choice([1.0, 2.0, 4.0], nc).astype(np.float32)
    flagged = np.zeros(n, np.uint8)
    flagged[1 % n] = 1
    ts, dec = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        dec, _ = host.activate_select(w1, h1, newest, KRKi, Kt, act, 2.0, 3.0, cand, hosts, ctype, flagged)
        ts.append(time.perf_counter() - t0)
    return {"activate_select_ms": float(np.median(ts) * 1e3), "candidates": int(nc), "active_points": int(len(act)),
            "chosen_for_optimisation": int((dec == 1).sum()), "deleted": int((dec == -1).sum()),
            "note": "host half of activatePointsMT (distance map + ordered candidate loop) on synthetic candidates; not part of keyframe_ms"}


def imu_timing(window, device, iters=400):
    """The same loop body with the IMU / spline block of solveSystemF (BASELINE.json configs 2-3; OB/EnergyFunctional.cpp:1053-1171):
    the device work is unchanged, the host assembles the IMU factors and solves the KKT system of dimension 4 + 1 + 29 n + constraints.
    `python bench.py --imu` times it as the headline loop; this is the short form for the default line."""
    from sos_slam_amd import host, synth
    win = synth.make_window(window)
    sysm = host.System.from_window(win, device=device)
    S, cal, fr, keep = synth.make_imu_records(win, consistent=True)
    HMi, bMi = synth.expand_prior_imu(win)
    sysm.set_imu(S, cal, fr, HMi, bMi)
    sysm.prepare()
    sysm.set_pipeline(True)
    for i in range(20):
        sysm.gn_iteration(i)
    host.timing(reset=True)
    t0 = time.perf_counter()
    for i in range(iters):
        sysm.gn_iteration(20 + i)
    dt = time.perf_counter() - t0
    ph = host.timing()
    api = host.imu()
    out = {"ms_per_iteration": dt / iters * 1e3, "gn_iter_per_s": iters / dt,
           # the solve in two calls (sos_imu.cpp): _prepare runs while the accumulation is in flight, _finish after the device's H / b
           "host_kkt_prepare_us": ph[1] / iters * 1e6, "host_kkt_finish_us": ph[2] / iters * 1e6, "host_kkt_solve_us": (ph[1] + ph[2]) / iters * 1e6,
           "kkt_dimension": 4 + 1 + 29 * win.n + 6 * (win.n - 2) + 3, "loop_mode": sysm.loop_mode(), "resInA_last_iteration": sysm.stats()["resInA"],
           "kkt_form": "scale trapped (first-estimate Jacobians): IMU states + multipliers eliminated once, kept while the linearisation "
                       "points and the prior stand; per iteration right-hand sides + border solve + two substitutions",
           "solves_kept_rebuilt_literal": list(api.solve_stats(reset=True))}
    try:   # what one optimize() pays once (new linearisation points after every keyframe), and the literal form for comparison
        arr = sysm._imu[2]
        arr[1].state_imu_zero[20] += 1e-12
        t0 = time.perf_counter()
        sysm.gn_iteration(20 + iters)
        out["iteration_with_factor_rebuild_ms"] = (time.perf_counter() - t0) * 1e3
        before = api.solve_mode(0)
        for i in range(5):
            sysm.gn_iteration(21 + iters + i)
        t0 = time.perf_counter()
        for i in range(60):
            sysm.gn_iteration(26 + iters + i)
        out["literal_form_ms_per_iteration"] = (time.perf_counter() - t0) / 60 * 1e3
        api.solve_mode(before)
        out["solves_kept_rebuilt_literal_after"] = list(api.solve_stats(reset=True))
    except Exception as e:   # a side measurement of a side measurement
        out["rebuild_literal_error"] = repr(e)[:200]
    sysm.close()
    return out


def device_solve_timing(n, device, reps=200):
    """k_gn_solve alone (the single-workgroup solve of the device-resident loop, csrc/sos_gn_resident.inc) on a random positive definite
    system of the window's dimension: HIP-event wall time per launch and the kernel's own phase stamps.  What it has to beat is the host
    round trip of the default loop (host_phases_us: gn_accumulate_wait + assemble + ldlt + the step's launch)."""
    from sos_slam_amd import lib
    d = 4 + 8 * n
    rng = np.random.default_rng(3)
    A = rng.standard_normal((d, d))
    H = A @ A.T + d * np.eye(d)
    z, zv = np.zeros((d, d)), np.zeros(d)
    ctx = lib.Context(64, 64, device=device)
    x, ph = ctx.gn_solve_system(np.triu(H), rng.standard_normal(d), z, zv, z, zv, zv, reps=reps)
    ctx.close()
    return {"dim": d, "launches": reps, "wall_us_per_launch": round(float(ph[3]), 2), "assemble_us": round(float(ph[0]), 2),
            "factorise_us": round(float(ph[1]), 2), "substitute_us": round(float(ph[2]), 2)}


def tracker_timing(window, device):
    """CoarseTracker / ScaleOptimizer latencies on the device (the other half of the hot path):
    trackNewestCoarse and optimizeScale of a new frame against the newest keyframe of the same window type."""
    from sos_slam_amd import host, synth
    from sos_slam_amd.synth import se3_exp12 as se3_exp, se3_mul12 as se3_mul
    win = synth.make_window(window, extra_frames=2)
    sysm = host.System.from_window(win, device=device)
    sysm.optimize(3)
    ht = host.HostTracker(sysm)
    pc_n = ht.set_ref()
    levels = int(np.count_nonzero(pc_n))
    new_slot, st_slot = sysm.upload_image(win.extra_images[0]), sysm.upload_image(win.extra_images[1])
    ref, new = win.frames[win.n - 1]["camToWorld"], win.extra_poses[0]
    Rr, tr, Rn, tn = ref[:9].reshape(3, 3), ref[9:], new[:9].reshape(3, 3), new[9:]
    T0 = np.concatenate([(Rn.T @ Rr).reshape(-1), Rn.T @ (tr - tn)])
    Tinit = se3_mul(se3_exp(np.array([0.004, -0.003, 0.002, 0.002, -0.002, 0.001])), T0)
    K1 = np.array(sysm.calib_value_scaled(), np.float32)

    def timeit(fn, reps=20):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e3

    out = {"template_pixels_per_level": [int(x) for x in pc_n[:levels]],
           "set_coarse_tracking_ref_ms": timeit(ht.set_ref),
           "track_newest_coarse_ms": timeit(lambda: ht.track(new_slot, 1.0, Tinit, np.zeros(2), levels - 1)),
           "optimize_scale_ms": timeit(lambda: ht.optimize_scale(st_slot, win.stereo_tfm, K1, 1.2, levels - 1))}
    # where one launch of trackNewestCoarse spends its time, from the kernel's own stamps (FS/CoarseTracker.cpp:366-552): residual
    # passes / all-gather of the chunk sums / bookkeeping between evaluations (8 x 8 solve, SE3::exp)
    try:
        ht.track(new_slot, 1.0, Tinit, np.zeros(2), levels - 1)
        out["track_lm_profile"] = ht.lm_profile(0)
    except Exception as e:  # noqa: BLE001
        out["track_lm_profile"] = {"error": repr(e)}
    sysm.close()
    return out


def exchange_info(sysm, win, world):
    """which per-iteration exchange the ranks run and how many bytes it moves (N > 1; null at N = 1)"""
    if world <= 1:
        return None
    import ctypes as C
    from sos_slam_amd import lib
    dim = 4 + 8 * win.n
    if os.environ.get("SOS_ABS_SC"):
        return {"kind": "all-reduce of the stitched fp64 system [H_A b_A | H_sc b_sc | count] (absolute-coordinate path)", "bytes": 8 * (2 * (dim * dim + dim) + 1)}
    dev, nfl = C.c_void_p(), C.c_size_t(0)
    try:
        lib.load().sos_ba_acc_buffer(L_host_ba(sysm), C.byref(dev), C.byref(nfl))
    except Exception:  # noqa: BLE001
        return {"kind": "all-reduce of the packed fp32 accumulator", "bytes": None}
    return {"kind": "all-reduce of the packed fp32 accumulator [top_A | top_L | accD | accE | accEB | Hcc | bc | counts] + all-gather of the newest-frame energies",
            "bytes": 4 * int(nfl.value)}


def kernel_sources_sha():
    """sha256 over the sources the roofline kernel is compiled from (what a committed counter value must have been measured on)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("sos_ba.hip", "sos_common.h", "sos_devmath.h"):
        with open(os.path.join(ROOT, "sos_slam_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(window):
    """roofline.traffic: HBM-side bytes per launch of the roofline kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; tools/collect_profiles.sh, tools/pmc_json.py).
    Counters cannot be read from inside the process, so the value comes from profiles/pmc_<window>.json -- and only when that file was
    collected on THESE kernel sources (its `kernel_sources_sha` equals the hash of the sources in the tree): otherwise `traffic` is null
    and the committed value is reported beside it as `traffic_stale`, with the commit it was measured at."""
    path = os.path.join(ROOT, "profiles", f"pmc_{window}.json")
    try:
        with open(path) as f:
            d = json.load(f)
        val = d["k_linearize_fused"]["traffic_bytes_per_launch"]
    except Exception:
        return {"traffic": None, "traffic_source": None}
    src = {"file": f"profiles/pmc_{window}.json", "source_commit": d.get("source_commit"), "kernel_sources_sha": d.get("kernel_sources_sha")}
    if d.get("kernel_sources_sha") == kernel_sources_sha():
        return {"traffic": val, "traffic_source": src}
    return {"traffic": None, "traffic_stale": val, "traffic_source": dict(src, note="collected on other kernel sources than this tree's: not this run's traffic")}


def L_host_ba(sysm):
    from sos_slam_amd import host
    import ctypes as C
    return C.c_void_p(host.load().sosf_ba(sysm.h_))


if __name__ == "__main__":
    main()
