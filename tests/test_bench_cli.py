"""bench.py's launcher logic on a box WITHOUT a GPU (this container): it must fail loudly, never fall back to a CPU path, and a request
for more ranks than there are devices must be refused before anything is started."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_refuses_more_ranks_than_devices():
    import torch
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 1
    p = _run("--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr, p.stderr[-400:]
    assert p.stdout.strip() == ""          # no line at all rather than a line with the wrong n_gpus


def test_no_cpu_fallback():
    if _have_gpu():
        return
    p = _run("--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and "no CPU fallback" in p.stderr, p.stderr[-400:]
    assert p.stdout.strip() == ""


def test_gpus_flag_must_match_the_launcher():
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr, p.stderr[-400:]


def test_side_measurement_process_fails_loudly_without_a_gpu_and_costs_only_its_entry():
    if _have_gpu():
        return
    p = _run("--side", "imu", "--window", "W7")
    assert p.returncode != 0 and "no CPU fallback" in p.stderr and p.stdout.strip() == ""
    sys.path.insert(0, ROOT)
    try:
        import bench
        r = bench.side_process("imu", "W7", timeout=120)   # the parent's view of the same failure: an error entry, no exception
    finally:
        sys.path.pop(0)
    assert set(r) == {"error"} and "exit code" in r["error"]


def test_activation_select_stage_runs_without_a_gpu():
    """bench.py -> keyframe -> activation_select: the host half of activatePointsMT timed on synthetic candidates (pure host code)"""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from sos_slam_amd import synth
    r = bench.activation_select_timing(synth.make_window("T6"), reps=2)
    assert r["candidates"] == 4 * r["active_points"] and 0 < r["chosen_for_optimisation"] < r["candidates"] and r["deleted"] > 0
    assert 0 < r["activate_select_ms"] < 100
