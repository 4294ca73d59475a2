"""The device-resident Gauss-Newton loop under a communicator (sos_ba_gn_resident_* with sos_ba_set_comm; csrc/sos_gn_resident.inc): N = 1
and N > 1 run the SAME chain -- accumulate | reduce | all-reduce of the packed accumulator | stitch | k_gn_solve | step | linearise |
all-gather of the newest frame's energies (OB/AccumulatedTopHessian.cpp:254-259 and OB/AccumulatedSCHessian.cpp:105-129 are the reductions
the exchange stands for).

  * one rank: the run with a communicator is bit-identical to the run without one (the exchange of one rank is the identity);
  * two ranks (processes, each with half of the points): the sharded resident loop against the sharded default loop (host solve) and
    against the unsharded window -- same iterations, same index sets, poses equal up to the solvers' round-off and, against the unsharded
    run, up to fp32 summation order.

On the GPU box the communicator is RCCL (torch's librccl, one rank -- two ranks cannot share the one device of a box).  Under tests/emu
(SOS_EMU=1) it is tests/emu/fake_rccl.cpp: the same entry points over shared memory between PROCESSES of the emulated library, which is
what lets the two-rank case run at all without hardware -- a logic check of the exchange code, never a multi-GPU result."""
import ctypes as C
import glob
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from sos_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emulated():
    if os.environ.get("SOS_EMU") != "1":
        return False
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


def _use_emulated_libraries():
    """what tests/conftest.py does at session start, for a process pytest did not start"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from sos_slam_amd import build as _b
    _b.HIP_LIB, _b.HOST_LIB = build_emu.build()
    _b.build_all = lambda *a, **k: (_b.HIP_LIB, _b.HOST_LIB)
    os.environ["SOS_FAKE_RCCL_EMU_LIB"] = _b.HIP_LIB
    return build_emu.build_fake_rccl()


def _fake_comm(idbytes, world, rank):
    from sos_slam_amd import build as _b
    from sos_slam_amd import lib as _lib
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    os.environ["SOS_FAKE_RCCL_EMU_LIB"] = _b.HIP_LIB
    L = _lib.load()
    assert L.sos_rccl_load(build_emu.build_fake_rccl().encode()) == 0
    if idbytes is None:
        buf = (C.c_ubyte * 128)()
        assert L.sos_rccl_unique_id(buf) == 0
        idbytes = bytes(buf)
    h = C.c_void_p()
    rc = L.sos_comm_create((C.c_ubyte * 128).from_buffer_copy(idbytes), world, rank, 0, C.byref(h))
    assert rc == 0, rc
    return L, h, idbytes


def _attach(sysm, h):
    from sos_slam_amd import host
    rc = host.load().sosf_set_comm(sysm.h_, h)
    assert rc == 0, rc


def _detach(sysm):
    from sos_slam_amd import host
    host.load().sosf_set_comm(sysm.h_, None)


def _run(win, resident, attach=None, iters=5):
    from sos_slam_amd import host
    sysm = host.System.from_window(win)
    sysm.set_resident(resident)
    if attach is not None:
        _attach(sysm, attach)
    rmse, its = sysm.optimize(iters)
    pi, tf = sysm.residual_ids()
    out = dict(rmse=float(rmse), its=int(its), mode=int(sysm.loop_mode()), x=sysm.lastX().copy(), idepth=sysm.points()["idepth"].copy(),
               poses=np.array([sysm.frame(f)["camToWorld"] for f in range(win.n)]), th=[float(sysm.frame(f)["frameEnergyTH"]) for f in range(win.n)],
               ids=sorted(zip(pi.tolist(), tf.tolist())), resInA=int(sysm.stats()["resInA"]))
    if attach is not None:
        _detach(sysm)
    sysm.close()
    return out


@pytest.mark.parametrize("name", ["T6", "W7"])
def test_one_rank_communicator_is_the_identity(name):
    win = synth.make_window(name)
    if _emulated():
        L, h, _ = _fake_comm(None, 1, 0)
        close = lambda: L.sos_comm_destroy(h)  # noqa: E731
    else:
        import socket
        import torch
        import torch.distributed as dist
        from sos_slam_amd import distributed as sdist
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        nc = sdist.NativeComm(dist, torch, 0)
        h = nc.h
        close = lambda: (nc.L.sos_comm_destroy(h), dist.destroy_process_group())  # noqa: E731
    try:
        plain = _run(win, True)
        comm = _run(win, True, attach=h)
    finally:
        close()
    assert plain["mode"] == 2 and comm["mode"] == 2, (plain["mode"], comm["mode"])   # both really ran the resident chain
    assert plain["its"] == comm["its"] and plain["rmse"] == comm["rmse"] and plain["resInA"] == comm["resInA"]
    assert np.array_equal(plain["x"], comm["x"]) and np.array_equal(plain["idepth"], comm["idepth"]) and np.array_equal(plain["poses"], comm["poses"])
    assert plain["th"] == comm["th"]


def _rank_main(rank, world, name, idbytes, resident, q):
    try:
        _use_emulated_libraries()
        win = synth.make_window(name)
        shard = synth.take_shard(win, synth.shard_points(win, rank, world))
        L, h, _ = _fake_comm(idbytes, world, rank)
        out = _run(shard, resident, attach=h)
        L.sos_comm_destroy(h)
        q.put((rank, out))
    except BaseException as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))


def _two_ranks(name, resident):
    ctx = mp.get_context("spawn")
    _, h0, idbytes = _fake_comm(None, 1, 0)   # (an id of this process's own; its one-rank communicator is not used)
    from sos_slam_amd import lib as _lib
    _lib.load().sos_comm_destroy(h0)
    buf = (C.c_ubyte * 128)()
    assert _lib.load().sos_rccl_unique_id(buf) == 0
    idbytes = bytes(buf)
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank_main, args=(r, 2, name, idbytes, resident, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    try:
        for _ in ps:
            r, out = q.get(timeout=1500)
            assert not isinstance(out, str), out
            res[r] = out
    finally:
        for p in ps:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        for f in glob.glob("/dev/shm/sos_fake_rccl_%d_*" % os.getpid()):
            os.unlink(f)
    return res


@pytest.mark.skipif(not _emulated(), reason="two ranks need two devices (the driver's N > 1 bench) or the emulator's process ranks")
@pytest.mark.parametrize("name", ["T6"])
def test_two_ranks_run_the_resident_chain(name):
    win = synth.make_window(name)
    whole = _run(win, True)
    res_sh = _two_ranks(name, True)
    def_sh = _two_ranks(name, False)
    assert whole["mode"] == 2
    for r in (0, 1):
        assert res_sh[r]["mode"] == 2, res_sh[r]["mode"]          # the sharded run really took the resident chain
        assert def_sh[r]["mode"] != 2
    # the ranks agree with each other exactly: same all-reduced system, same deterministic solve, replicated frame states
    a, b = res_sh[0], res_sh[1]
    assert a["its"] == b["its"] and np.array_equal(a["x"], b["x"]) and np.array_equal(a["poses"], b["poses"]) and a["th"] == b["th"]
    assert a["resInA"] == b["resInA"]                              # (the count is a global one)
    # resident against default, both sharded: the same H / b up to which kernels formed the tile sums, the same step up to the solvers
    d = def_sh[0]
    assert a["its"] == d["its"] and abs(a["rmse"] - d["rmse"]) <= 1e-6 * d["rmse"]
    assert np.abs(a["poses"] - d["poses"]).max() < 1e-6
    assert np.allclose(a["th"], d["th"], rtol=1e-5)
    # sharded against the unsharded window: fp32 summation order of the accumulators
    assert a["its"] == whole["its"] and abs(a["rmse"] - whole["rmse"]) <= 1e-4 * whole["rmse"]
    assert np.abs(a["poses"] - whole["poses"]).max() < 1e-5
    assert a["resInA"] == whole["resInA"]
    # the shards' points together are the window's points, stepped alike
    n0, n1 = len(res_sh[0]["idepth"]), len(res_sh[1]["idepth"])
    assert n0 + n1 == len(whole["idepth"])
