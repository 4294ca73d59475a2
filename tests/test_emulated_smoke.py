"""CPU suite: a slice of the -m gpu parity tests against tests/emu, the lockstep CPU emulation of the device library (SOS_EMU=1), in a
child process -- so that a round without GPU access still EXECUTES the kernels' source against the oracle and the fixtures (logic only: a
run under the emulation is never GPU evidence, tests/emu/README.md).  On a box with a GPU the child ignores SOS_EMU and this test is skipped:
the real suite runs there."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu() or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="a GPU is visible (the real -m gpu suite runs) or no host clang++")
@pytest.mark.parametrize("target", [
    "tests/test_golden_t6.py",                                   # backend, tracker and immature-point fixtures of the T6 window
    "tests/test_gpu_backend.py -k T4",                            # bit-exact linearisation / accumulation / back-substitution on the small window
    # the exchange code with TWO ranks (processes of the emulation over tests/emu/fake_rccl.cpp): default and device-resident loop, sharded
    # against each other and against the unsharded window; and k_gn_solve with its pivot-row micro-test
    "tests/test_gpu_resident_comm.py -k two_ranks",
    "tests/test_gpu_gn_solve.py -k row_update",
])
def test_gpu_parity_slice_under_emulation(target):
    env = dict(os.environ, SOS_EMU="1")
    r = subprocess.run([sys.executable, "-m", "pytest"] + target.split() + ["-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]


@pytest.mark.skipif(_have_gpu() or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="a GPU is visible (the driver runs smoke() there) or no host clang++")
def test_graft_entry_smoke_under_emulation():
    """__graft_entry__.smoke() -- what the driver runs on the MI355X before the bench -- with the two library paths pointed at the emulation:
    its asserts against the oracle hold for the sources as they are now."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import build_emu\n"
            "from sos_slam_amd import build as _b\n"
            "_b.HIP_LIB, _b.HOST_LIB = build_emu.build()\n"
            "_b.build_all = lambda *a, **k: (_b.HIP_LIB, _b.HOST_LIB)\n"
            "import __graft_entry__ as g\n"
            "g.smoke()\n"
            "print('EMU_SMOKE_OK')\n") % (os.path.join(ROOT, "tests", "emu"), ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "EMU_SMOKE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("smoke ok") == 2, r.stdout[-500:]
