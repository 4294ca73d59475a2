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
])
def test_gpu_parity_slice_under_emulation(target):
    env = dict(os.environ, SOS_EMU="1")
    r = subprocess.run([sys.executable, "-m", "pytest"] + target.split() + ["-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]
