"""The facade's blocked LDL^T on the LARGE systems (the visual-inertial KKT solve, dimension 401 at twelve keyframes): the in-place
form used by sosf_imu_solve and the opt-in helper threads (SOS_SOLVE_THREADS) against the unblocked reference solve.  CPU only.
The threaded runs happen in child processes with a timeout -- a barrier that does not release is a failure, not a hung suite."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from sos_slam_amd import host
out = {}
for n in (100, 192, 401):
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n + 4))
    A = B @ B.T / n + np.eye(n)
    # an indefinite border like the multipliers of the KKT system (zero diagonal, quasi-definite in this order)
    k = 24 if n > 100 else 0
    if k:
        J = np.zeros((k, n - k)); J[np.arange(k), rng.integers(0, n - k, k)] = 1.0; J += 0.1 * rng.normal(size=J.shape) * (rng.random(J.shape) < 0.05)
        A[n - k:, :n - k] = J; A[:n - k, n - k:] = J.T; A[n - k:, n - k:] = 0
    b = rng.normal(size=n)
    xs = [host.ldlt_solve(A, b, 0) for _ in range(3)]
    assert all(np.array_equal(xs[0], x) for x in xs[1:])
    xr = host.ldlt_solve(A, b, 1)
    # the partial factorisation of the kept-factor visual-inertial solve (leading block = everything but the last n/4 unknowns; the
    # multipliers sit at the end of the matrix above, so here the trailing block is indefinite and the leading one positive definite)
    xp = host.ldlt_partial_solve(A, b, n - n // 4)
    out[str(n)] = {"x": xs[0].tolist(), "err_ref": float(np.abs(xs[0] - xr).max() / np.abs(xr).max()),
                   "resid": float(np.linalg.norm(A @ xs[0] - b) / np.linalg.norm(b)),
                   "xp": xp.tolist(), "err_partial": float(np.abs(xp - xr).max() / np.abs(xr).max())}
print(json.dumps(out))
"""


def _run(threads):
    env = dict(os.environ)
    env.pop("SOS_SOLVE_THREADS", None)
    if threads:
        env["SOS_SOLVE_THREADS"] = str(threads)
    p = subprocess.run([sys.executable, "-c", _CHILD % ROOT], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_large_solve_single_thread_and_helper_threads_agree():
    base = _run(None)
    for n, r in base.items():
        assert r["err_ref"] < 1e-9 and r["resid"] < 1e-10, (n, r)
    two, three = _run(2), _run(3)
    for n in base:
        assert two[n]["err_ref"] < 1e-9 and two[n]["resid"] < 1e-10
        # rows are owned by absolute index: what is computed for an element does not depend on the number of threads
        assert two[n]["x"] == three[n]["x"], n
        assert two[n]["xp"] == three[n]["xp"], n
        assert two[n]["err_partial"] < 1e-9 and base[n]["err_partial"] < 1e-9
        # below the threshold dimension the helpers are not used at all
        if int(n) < 192:
            assert two[n]["x"] == base[n]["x"]
        else:
            assert np.allclose(two[n]["x"], base[n]["x"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("nb", [0, 1, 5, 6, 7, 15, 16, 17, 31, 32, 33, 48, 96, 101, 111, 128])
def test_partial_factorisation_border_widths(nb):
    """The trailing block's update of ldlt_partial_factor works on packed 6 x 16 register tiles, right-aligned over the border
    columns: widths around the tile and row-group boundaries (a ragged first tile, none, a last row group that reads the padding)
    against the dense solve, with AVX-512 and with the fallback path."""
    m = 57
    n = m + nb
    rng = np.random.default_rng(1000 + nb)
    B = rng.normal(size=(n, n + 3))
    A = B @ B.T / n + np.eye(n)
    A[5, 5] = 1e-3                      # a pivot the threshold test moves
    b = rng.normal(size=n)
    xr = np.linalg.solve(A, b)
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from sos_slam_amd import host; "
            "d = np.load(sys.argv[1]); print(json.dumps(host.ldlt_partial_solve(d['A'], d['b'], int(d['m'])).tolist()))" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "sys.npz")
        np.savez(f, A=A, b=b, m=m)
        xs = []
        for no512 in (False, True):
            env = dict(os.environ)
            env.pop("SOS_NO_AVX512", None)
            if no512:
                env["SOS_NO_AVX512"] = "1"
            p = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
            assert p.returncode == 0, p.stderr[-2000:]
            xs.append(np.array(json.loads(p.stdout.strip().splitlines()[-1])))
    for x in xs:
        assert np.abs(x - xr).max() <= 1e-10 * np.abs(xr).max()
    assert np.abs(xs[0] - xs[1]).max() <= 1e-12 * np.abs(xr).max()


def test_partial_factorisation_tiny_leading_blocks():
    """leading blocks shorter than the padding in front of the first border tile, and none at all"""
    from sos_slam_amd import host
    for m, nb in [(3, 5), (1, 1), (2, 17), (9, 40), (1, 101), (0, 7)]:
        n = m + nb
        rng = np.random.default_rng(n * 31 + m)
        B = rng.normal(size=(n, n + 3))
        A = B @ B.T / n + np.eye(n)
        b = rng.normal(size=n)
        xr = np.linalg.solve(A, b)
        assert np.abs(host.ldlt_partial_solve(A, b, m) - xr).max() <= 1e-12 * np.abs(xr).max(), (m, nb)
