"""bench.py end to end against tests/emu (TEST INFRASTRUCTURE; timings meaningless): shows that the line is printed twice -- complete after the
headline + roofline + cpu_baseline, enriched at the end -- and that every side entry is filled.   python tools/emu_bench.py --window T6 --steps 2 --warmup 1 --inner 2 --cpu-seconds 1"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu
from sos_slam_amd import build as _b
_b.HIP_LIB, _b.HOST_LIB = build_emu.build()
_b.build_all = lambda *a, **k: (_b.HIP_LIB, _b.HOST_LIB)
import torch
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
sys.argv = ["bench.py"] + sys.argv[1:]
os.chdir(ROOT)
import bench
# side processes would start the real (GPU-less) bench: run the sides in-process
def _side(what, window, timeout=240, env=None):
    if what == "gnsolve":
        from sos_slam_amd import synth
        return bench.device_solve_timing(synth.WINDOWS[window]["n"], 0, reps=3)
    if what == "tracker":
        return bench.tracker_timing(window, 0)
    if what == "keyframe":
        return bench.keyframe_timing(window, 0)
    return bench.imu_timing(window, 0, iters=3)
bench.side_process = _side
# the launch variants are child processes of the real bench as well: not run here (their command lines run through this tool one by one,
# e.g. `--no-cpu-baseline --no-sides --resident`)
bench.variant_timing = lambda window, **k: {n: {"error": "child process: not run under tools/emu_bench.py"} for n in (k.get("only") or [v[0] for v in bench.VARIANTS])}
bench.main()
