"""bench.py's N > 1 flow rehearsed on the one GPU of the test box (SOS_BENCH_SINGLE_GPU=1: every rank on device 0, gloo instead of RCCL, the
exchange through the torch.distributed hooks on host copies): the ranks issue matching collectives (no hang), rank 0 prints exactly ONE
line on stdout, and the totals are those of the scaling mode -- weak: every rank its own W12 point set; strong: ONE W16 window sharded by
point (BASELINE.json config 5).  Timings mean nothing in this mode."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, scaling, steps=3, warmup=1, inner=None, launcher=False, extra_env=None):
    """bench.py --gpus n.  launcher=False: the plain command the driver runs (bench.py starts its own ranks); True: torchrun outside."""
    env = dict(os.environ, SOS_BENCH_SINGLE_GPU="1", **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    pre = [sys.executable]
    if launcher:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        pre += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    p = subprocess.run(pre + [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup), "--scaling", scaling,
                              "--inner", str(inner or 2), "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-800:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


MODE1 = "host solve (blocked LDL^T), device everything else incl. the step"


@pytest.mark.needs_device   # torch.cuda / RCCL / bench.py timing: not something tests/emu stands in for
def test_two_ranks_weak_and_strong():
    w = _run(2, "weak")
    assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["steps"] == 3 and w["value"] > 0
    assert w["config"]["residuals_total"] > 1.9 * w["config"]["residuals_per_gpu"]          # two different point sets of one size
    s = _run(2, "strong")
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and "W16" in s["config"]["workload"]
    from sos_slam_amd import synth
    assert s["config"]["residuals_total"] == synth.make_window("W16").R                    # the shards add up to the one window
    for d in (w, s):
        assert "roofline" in d and d["roofline"]["bound"] == "hbm" and d["higher_is_better"] is True
        # the line names the loop that RAN: the device-side step also with an exchange attached (x and the frame states are replicated)
        assert d["config"]["gn_loop"].startswith(MODE1), d["config"]["gn_loop"]
        assert d["cpu_baseline"] is None and d["config"]["resInA_last_iteration"] > 0.5 * d["config"]["residuals_total"]
    t = _run(2, "weak", launcher=True)   # the launcher outside, as the task's contract spells it
    assert t["n_gpus"] == 2 and abs(t["last_step_l2"] - w["last_step_l2"]) <= 1e-12 + 1e-9 * w["last_step_l2"]


@pytest.mark.needs_device   # torch.cuda / RCCL / bench.py timing: not something tests/emu stands in for
def test_sharded_first_step_equals_the_unsharded_one():
    """Strong mode, ONE Gauss-Newton iteration: the step the solve produces from the window sharded over two and three ranks (partial
    accumulators summed by the exchange) is the step of the unsharded window up to fp32 summation order."""
    import numpy as np
    ref = _run(1, "strong", steps=1, warmup=0, inner=1)
    assert ref["last_step_l2"] > 1e-3
    for n in (2, 3):
        d = _run(n, "strong", steps=1, warmup=0, inner=1)
        assert d["config"]["residuals_total"] == ref["config"]["residuals_total"]
        assert abs(d["last_step_l2"] - ref["last_step_l2"]) <= 1e-5 * ref["last_step_l2"], (n, d["last_step_l2"], ref["last_step_l2"])
        assert np.allclose(d["last_step_head"], ref["last_step_head"], rtol=1e-3, atol=1e-9)
        assert d["config"]["gn_loop"].startswith(MODE1) and d["config"]["resInA_last_iteration"] == ref["config"]["resInA_last_iteration"]


@pytest.mark.needs_device   # torch.cuda / RCCL / bench.py timing: not something tests/emu stands in for
def test_one_rank_rccl_exchange_is_bit_identical_to_the_plain_run():
    """The library-enqueued RCCL path (all-reduce of the packed accumulator, all-gather of the newest-frame energies, chained publish)
    with ONE rank against the run without a communicator: same loop mode, the same step bit for bit after six iterations."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    outs = []
    for force in ("0", "1"):
        e = dict(env, SOS_BENCH_FORCE_DIST=force)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "0", "--inner", "2", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
        assert p.returncode == 0, p.stderr[-800:]
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][0]))
    a, b = outs
    assert b["config"]["parallelism"].startswith("1 ranks") and "enqueued by the library" in b["config"]["parallelism"], b["config"]["parallelism"]
    assert a["config"]["gn_loop"].startswith(MODE1) and b["config"]["gn_loop"] == a["config"]["gn_loop"]
    assert a["last_step_l2"] == b["last_step_l2"] and a["last_step_head"] == b["last_step_head"]
    assert a["config"]["resInA_last_iteration"] == b["config"]["resInA_last_iteration"]
