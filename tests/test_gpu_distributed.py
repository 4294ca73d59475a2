"""The RCCL exchange path on one GPU: a world-size-1 `nccl` process group drives the facade's all-reduce and
order-statistic hooks (sos_slam_amd/distributed.py).  With one rank the all-reduce is the identity, so the hooked
run must reproduce the un-hooked run bit for bit -- this pins the plumbing (packed device buffer handed to
torch zero-copy, stream ordering around the collective, hook signatures); the N > 1 arithmetic is covered by
tests/test_distributed_gloo.py on CPU."""
import os
import socket

import numpy as np
import pytest

from sos_slam_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.needs_device   # torch.cuda / RCCL / bench.py timing: not something tests/emu stands in for
def test_exchange_paths_equal_plain_run():
    import torch
    import torch.distributed as dist
    from sos_slam_amd import distributed as sdist
    from sos_slam_amd import host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        win = synth.make_window("T6")
        out, priors = [], []
        comm = sdist.NativeComm(dist, torch, 0)
        for mode in ("plain", "hooks", "native"):
            sysm = host.System.from_window(win)
            if mode == "plain":
                sysm.set_resident(False)   # the exchange paths solve on the host: compare like with like (same kernels, same solver)
            if mode == "hooks":
                sdist.attach(sysm, dist, torch)
            elif mode == "native":  # RCCL all-reduce / all-gather enqueued by the library on its own stream
                comm.attach(sysm)
            rmse, its = sysm.optimize(4)
            pts = sysm.points()
            out.append((rmse, its, sysm.lastX().copy(), pts["idepth"].copy(),
                        [sysm.frame(f)["frameEnergyTH"] for f in range(win.n)]))
            # keyframe-rate exchange: the marginalisation prior update is summed over ranks (native: RCCL fp64 all-reduce
            # on the library's stream; hooks: the fp64 callback)
            ids = sysm.point_ids()
            sel = ids[win.points["host"][ids] == 0][:16]
            sysm.marginalize_points(sel)
            priors.append(sysm.get_prior())
            if mode == "native":
                # a few pipelined loop bodies as bench.py runs them (prefetched accumulate incl. the all-reduce)
                sysm.prepare()
                sysm.set_pipeline(True)
                for it in range(3):
                    sysm.gn_iteration(it)
                assert np.isfinite(sysm.lastX()).all()
                comm.close()
            sysm.close()
        a, hooks, native = out
        # the library-enqueued RCCL path runs the same kernels as the plain run (pipelined iterations, block sums on the
        # matrix cores): bit-identical.  The host-callback path cannot prefetch, so its intermediate iterations take the
        # stored-tile kernels, whose fp32 block sums round differently: equal to fp32 accumulation noise.
        assert a[0] == native[0] and a[1] == native[1]
        assert np.array_equal(a[2], native[2]) and np.array_equal(a[3], native[3]) and a[4] == native[4]
        assert abs(a[0] - hooks[0]) <= 1e-5 * a[0] and a[1] == hooks[1]
        assert np.abs(a[2] - hooks[2]).max() <= 1e-5
        assert np.abs(a[3] - hooks[3]).max() <= 1e-5 * np.abs(a[3]).max()
        assert np.allclose(a[4], hooks[4], rtol=1e-3)
        assert np.array_equal(priors[0][0], priors[2][0]) and np.array_equal(priors[0][1], priors[2][1])
        assert np.abs(priors[0][0]).max() > 0
        assert np.abs(priors[0][0] - priors[1][0]).max() <= 1e-3 * np.abs(priors[0][0]).max()
    finally:
        dist.destroy_process_group()
