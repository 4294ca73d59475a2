"""The facade's per-keyframe graph walks on helper threads (csrc/host/sos_pool.hpp: the record walk of packWindow,
FS/FullSystemOptimize.cpp:316-329, and the consumer of linearizeAll(true), :125-182 -- loops the reference runs on its IndexThreadReduce
workers): contiguous ranges laid out by a prefix sum, so the snapshot, the active list, the removal list and with them every result are
the serial loop's, byte for byte, for any thread count."""
import hashlib

import numpy as np
import pytest

from sos_slam_amd import host, synth

pytestmark = pytest.mark.gpu


def _digest(name, threads):
    win = synth.make_window(name)
    sysm = host.System.from_window(win)
    sysm.set_host_threads(threads)
    assert sysm.host_threads() == threads
    h = hashlib.sha256()
    out = []
    for rnd in range(2):                       # two keyframe-like rounds: optimize, drop / marginalise, optimize the changed graph
        rm, it = sysm.optimize(4)
        pi, tf = sysm.residual_ids()
        res = sysm.residuals() if hasattr(sysm, "residuals") else None
        for a in (sysm.lastX(), sysm.points()["idepth"], np.array([sysm.frame(f)["camToWorld"] for f in range(win.n)]),
                  np.array([sysm.frame(f)["frameEnergyTH"] for f in range(win.n)], np.float32), pi, tf):
            h.update(np.ascontiguousarray(a).tobytes())
        out.append((float(rm), int(it), len(pi)))
        ids = sysm.point_ids()
        sel = ids[win.points["host"][ids] == rnd][:24]
        sysm.marginalize_points(sel)
        Hp, bp = sysm.get_prior()
        h.update(np.ascontiguousarray(Hp).tobytes()); h.update(np.ascontiguousarray(bp).tobytes())
    sysm.set_host_threads(4)
    sysm.close()
    return h.hexdigest(), out


@pytest.mark.parametrize("name", ["T4", "W7"])
def test_results_do_not_depend_on_the_host_thread_count(name):
    ref = _digest(name, 1)
    for t in (2, 3, 7):
        got = _digest(name, t)
        assert got == ref, (t, got[1], ref[1])


def test_two_systems_driven_from_two_threads():
    """The helper threads belong to one walk at a time; a second system optimised concurrently from another thread of the process walks by
    itself -- both end exactly where they end when run one after the other."""
    import threading
    ref = [_digest("T4", 4), _digest("T6", 4)]
    got = [None, None]

    def work(i, name):
        got[i] = _digest(name, 4)

    ts = [threading.Thread(target=work, args=(0, "T4")), threading.Thread(target=work, args=(1, "T6"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert got == ref
