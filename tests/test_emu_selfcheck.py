"""Who checks the checker: tests/emu (the lockstep CPU emulation the -m gpu suite can run against, SOS_EMU=1) on kernels with known
answers -- the cross-lane operations and DPP controls the product kernels use, the MFMA lane <-> element layout, workgroup barriers,
co-resident workgroups behind a device-wide barrier, a kernel enqueued before its input that waits for a host flag in mapped memory.
Runs on the CPU; also the early warning that tests/emu still builds."""
import ctypes as C
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="the emulator is built with the ROCm host clang++")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def L():
    import build_emu
    lib = C.CDLL(build_emu.build_selftest())
    for f in ("emu_selftest_lane_ops", "emu_selftest_mfma", "emu_selftest_transpose", "emu_selftest_grid_barrier", "emu_selftest_mailbox",
              "emu_selftest_mfma_f64", "emu_selftest_row_newbcast", "emu_selftest_lds_limit"):
        getattr(lib, f).restype = C.c_int
    return lib


def test_cross_lane_operations(L):
    rng = np.random.default_rng(1)
    n = 512
    v = rng.normal(size=n).astype(np.float32)
    ox, od, up = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    bal, first = np.zeros(n // 64, np.uint64), np.zeros(n, np.int32)
    assert L.emu_selftest_lane_ops(_p(v), n, _p(ox), _p(od), _p(bal), _p(first), _p(up)) == 0
    for w in range(n // 64):
        x = v[64 * w:64 * w + 64]
        s = x.copy()                                  # the butterfly, in its own order
        for o in (32, 16, 8, 4, 2, 1):
            s = (s + s[np.arange(64) ^ o]).astype(np.float32)
        assert np.array_equal(ox[64 * w:64 * w + 64], s)
        a = x.copy()                                  # the DPP scan: row_shr 1, 2, 4, 8 inside 16-lane rows, then row_bcast 15 / 31
        for k in (1, 2, 4, 8):
            sh = np.zeros(64, np.float32)
            for r in range(4):
                sh[16 * r + k:16 * r + 16] = a[16 * r:16 * r + 16 - k]
            a = (a + sh).astype(np.float32)
        b = a.copy()
        b[16:32] += a[15]; b[48:64] += a[47]
        c = b.copy()
        c[32:64] += b[31]
        assert np.array_equal(od[64 * w:64 * w + 64], c)
        assert abs(float(od[64 * w + 63]) - float(x.astype(np.float64).sum())) < 1e-4
        mask = sum(1 << i for i in range(64) if x[i] > 0)
        assert int(bal[w]) == mask
        pos = np.flatnonzero(x > 0)
        f = first[64 * w:64 * w + 64]
        assert np.all(f[x <= 0] == -1) and np.all(f[pos] == 64 * w + pos[0])    # readfirstlane inside a divergent region
        e = x.copy()
        idx = np.arange(64)
        e[idx % 8 != 0] = x[idx[idx % 8 != 0] - 1]                              # __shfl_up(v, 1, 8): the first lane of a segment keeps its own
        assert np.array_equal(up[64 * w:64 * w + 64], e)


@pytest.mark.parametrize("K", [4, 32])
def test_mfma_lane_layout(L, K):
    rng = np.random.default_rng(K)
    A, B = rng.normal(size=(16, K)).astype(np.float32), rng.normal(size=(K, 16)).astype(np.float32)
    D = np.zeros((16, 16), np.float32)
    assert L.emu_selftest_mfma(_p(A), _p(B), _p(D), K) == 0
    assert np.abs(D - A.astype(np.float64) @ B.astype(np.float64)).max() < 1e-5 * K


@pytest.mark.parametrize("K", [4, 16])
def test_mfma_f64_lane_layout(L, K):
    """v_mfma_f64_16x16x4_f64 (round 5, k_gn_solve's trailing update): its accumulator rows are (lane >> 4) + 4 * reg, NOT the f32 form's
    4 * (lane >> 4) + reg (cdna_hip_programming.md) -- an asymmetric B tells the two apart."""
    rng = np.random.default_rng(K + 1)
    A, B = rng.normal(size=(16, K)), rng.normal(size=(K, 16))
    D = np.zeros((16, 16))
    assert L.emu_selftest_mfma_f64(_p(A), _p(B), _p(D), K) == 0
    assert np.abs(D - A @ B).max() < 1e-13 * K
    assert np.abs(D - (A @ B).T).max() > 1e-3


def test_row_newbcast_of_64_bit_values(L):
    """v_mov_b64_dpp row_newbcast:n (k_gn_solve's pivot-row broadcast): lane n of each 16-lane row to all 16 lanes of that row."""
    v = np.random.default_rng(2).normal(size=64)
    out = np.zeros(192)
    assert L.emu_selftest_row_newbcast(_p(v), _p(out)) == 0
    for q, n in enumerate((0, 5, 15)):
        assert np.array_equal(out[64 * q:64 * q + 64], np.repeat(v[n::16][:4], 16))


def test_lds_limit_of_a_compute_unit():
    """More than 160 KB of LDS per workgroup does not exist on gfx950: the emulator refuses the launch (it ends the process, so the probe
    runs in a child) instead of emulating a kernel that could not be launched."""
    import subprocess
    code = ("import ctypes, sys; sys.path.insert(0, %r); import build_emu; L = ctypes.CDLL(build_emu.build_selftest()); "
            "L.emu_selftest_lds_limit(int(sys.argv[1])); print('ran')" % os.path.join(ROOT, "tests", "emu"))
    ok = subprocess.run([sys.executable, "-c", code, "150"], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0 and "ran" in ok.stdout, ok.stderr[-400:]
    bad = subprocess.run([sys.executable, "-c", code, "170"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "160 KB" in bad.stderr, (bad.returncode, bad.stderr[-400:])


def test_divergent_barrier_is_refused():
    """s_barrier counts waves: lanes of ONE wave waiting in different __syncthreads statements (a barrier in both arms of a lane-divergent
    branch -- round 5's k_gn_solve --, or a guarded barrier part of the wave skipped) is a kernel the hardware runs out of step.  The
    emulator ends the process on it; the wave-uniform and the hoisted forms run."""
    import subprocess
    code = ("import ctypes, sys, numpy as np; sys.path.insert(0, %r); import build_emu; L = ctypes.CDLL(build_emu.build_selftest()); "
            "o = np.zeros(256, np.float32); L.emu_selftest_barrier_shapes(o.ctypes.data_as(ctypes.c_void_p), int(sys.argv[1]), int(sys.argv[2])); "
            "n = int(sys.argv[1]); e = np.arange(256, dtype=np.float32); e[:n] = (np.arange(n) + 1) %% n; assert np.array_equal(o, e); print('ran')"
            % os.path.join(ROOT, "tests", "emu"))
    for n, mode in ((112, 0), (112, 1)):
        bad = subprocess.run([sys.executable, "-c", code, str(n), str(mode)], capture_output=True, text=True, timeout=300)
        assert bad.returncode != 0 and "divergent __syncthreads" in bad.stderr and "wave 1" in bad.stderr, (n, mode, bad.returncode, bad.stderr[-600:])
    for n, mode in ((128, 0), (128, 2), (112, 3), (128, 3)):
        ok = subprocess.run([sys.executable, "-c", code, str(n), str(mode)], capture_output=True, text=True, timeout=300)
        assert ok.returncode == 0 and "ran" in ok.stdout, (n, mode, ok.stderr[-600:])


def test_barriers_lds_and_coresident_workgroups(L):
    rng = np.random.default_rng(3)
    nb = 5
    x = rng.normal(size=(nb, 32, 32)).astype(np.float32)
    y = np.zeros_like(x)
    assert L.emu_selftest_transpose(_p(x), _p(y), nb) == 0
    assert np.array_equal(y, x.transpose(0, 2, 1))
    for nblocks in (3, 200):
        s = np.zeros(nblocks, np.float32)
        assert L.emu_selftest_grid_barrier(nblocks, _p(s)) == 0
        assert np.all(s == nblocks * (nblocks + 1) / 2)


def test_kernel_waiting_for_the_host(L):
    x = np.arange(64, dtype=np.float64)
    y = np.zeros(64)
    for delay in (0, 2000):
        assert L.emu_selftest_mailbox(_p(x), _p(y), delay) == 0
        assert np.array_equal(y, 2 * x)


def test_product_sources_still_build_for_the_emulator():
    """the textual rewrites of build_emu.py on the current csrc/ and a compile of the result"""
    import build_emu
    hip, host = build_emu.build()
    assert os.path.exists(hip) and os.path.exists(host)
    lib = C.CDLL(hip, mode=C.RTLD_GLOBAL)
    from sos_slam_amd import lib as plib
    missing = [s for s in plib.SYMBOLS if not hasattr(lib, s)]
    assert not missing, missing
