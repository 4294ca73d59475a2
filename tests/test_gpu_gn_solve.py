"""k_gn_solve -- the single-workgroup solve of the device-resident Gauss-Newton loop (csrc/sos_gn_resident.inc: EnergyFunctional::solveSystemF,
OB/EnergyFunctional.cpp:1046-1148, visual part) -- against the facade's host solve (sosf_solve_system: blocked LDL^T with threshold pivoting)
and the NumPy mirror in extended precision, on every system the oracle solves at T6 / W7 / W12 / W16 and along a rolling chain: the
device factorises unpivoted, in 16 x 16 blocks (DPP row broadcasts inside the diagonal block, fp64 MFMA for the trailing tiles, the
right-hand side carried as a row of the matrix), so x may differ from the host's by solver round-off only."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc
from sos_slam_amd import host, lib, synth
from tests import rolling

pytestmark = pytest.mark.gpu


def _collect(run):
    Lo = orc.lib()
    vp = C.c_void_p
    TAP = C.CFUNCTYPE(None, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp)
    Lo.orc_set_solve_tap.argtypes = [TAP]
    systems = []

    def arr(p, shape):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape).copy()

    def tap(n, H, b, Hsc, bsc, HM, bM, delta, lam, x):
        d = 4 + 8 * n
        M = arr(HM, (d, d))
        M = np.tril(M) + np.tril(M, -1).T
        # (the oracle, as Eigen, reads lower triangles; the device and the facade read upper ones)
        systems.append((arr(H, (d, d)).T.copy(), arr(b, (d,)), arr(Hsc, (d, d)).T.copy(), arr(bsc, (d,)), M, arr(bM, (d,)), arr(delta, (d,)), float(lam)))

    cb = TAP(tap)
    Lo.orc_set_solve_tap(cb)
    try:
        run()
    finally:
        Lo.orc_set_solve_tap(C.cast(None, TAP))
    return systems


def _check(systems, ctx):
    from oracle import mirror_np as mir
    worst_host = worst_mirror = 0.0
    for (H, b, Hs, bs, M, bM, dl, lam) in systems:
        assert lam == 1e-5   # SOLVER_FIX_LAMBDA is what the kernel is built for
        xd, ph = ctx.gn_solve_system(H, b, Hs, bs, M, bM, dl)
        xh = host.solve_system(H, b, Hs, bs, M, bM, dl, lam)
        xm = np.asarray(mir.solve_system(H.T, b, Hs.T, bs, M, bM, dl, lam), dtype=np.float64)
        sc = np.abs(xm).max()
        worst_host = max(worst_host, float(np.abs(xd - xh).max() / sc))
        worst_mirror = max(worst_mirror, float(np.abs(xd - xm).max() / sc), )
        # the host's own distance from the extended-precision solve is the yardstick of solver round-off on this system
        yard = float(np.abs(xh - xm).max() / sc)
        assert np.abs(xd - xm).max() / sc <= max(1e-10, 20 * yard), (len(b), np.abs(xd - xm).max() / sc, yard)
    return worst_host, worst_mirror


def test_device_solve_on_the_systems_of_running_chains():
    def run():
        for name in ("T6", "W7"):
            orc.window_from_synth(synth.make_window(name)).optimize(6)
        sc = rolling.Scenario(n_frames=14)
        ch = rolling.OracleChain(sc)
        ch.bootstrap()
        while ch.next_frame < sc.n_frames:
            ch.step()
    systems = _collect(run)
    assert len(systems) >= 25
    ctx = lib.Context(64, 64)
    wh, wm = _check(systems, ctx)
    ctx.close()
    print(f"{len(systems)} systems (dim {sorted(set(len(s[1]) for s in systems))}): device vs host solve max {wh:.1e}, vs extended-precision mirror max {wm:.1e} (relative to max |x|)")
    assert wh < 1e-9


@pytest.mark.parametrize("name", ["T3", "W12", "W16"])
def test_device_solve_at_baseline_sizes(name):
    """dim = 28 (2 tiles, the right-hand side inside the last diagonal block), 100 (BASELINE.json's W12: 7 block columns), 132 (W16: 9)."""
    systems = _collect(lambda: orc.window_from_synth(synth.make_window(name)).optimize(2))
    assert systems
    ctx = lib.Context(64, 64)
    wh, wm = _check(systems, ctx)
    ctx.close()
    print(f"{name}: {len(systems)} systems of dimension {len(systems[0][1])}: device vs host max {wh:.1e}, vs mirror max {wm:.1e}")
    assert wh < 1e-9


def test_device_solve_reports_a_non_positive_pivot():
    """An indefinite system is not the unpivoted kernel's business: it says so (SOS_ERR_STATE) instead of returning garbage."""
    n, d = 3, 28
    rng = np.random.default_rng(5)
    A = rng.standard_normal((d, d))
    H = A @ A.T + 50 * np.eye(d)
    H[5, 5] = -1e6            # (after damping and Jacobi scaling still negative)
    z = np.zeros((d, d))
    ctx = lib.Context(64, 64)
    with pytest.raises(lib.SosError):
        ctx.gn_solve_system(np.triu(H), rng.standard_normal(d), z, np.zeros(d), z, np.zeros(d), np.zeros(d))
    x, _ = ctx.gn_solve_system(np.triu(A @ A.T + 50 * np.eye(d)), rng.standard_normal(d), z, np.zeros(d), z, np.zeros(d), np.zeros(d))
    assert np.isfinite(x).all()
    ctx.close()


def test_row_update_asm_equals_builtin_form():
    """gs_row_update<K> -- fifteen asm statements of v_fmac_f64_dpp ... row_newbcast:K with hand-placed s_nop around them (the VALU-write ->
    DPP-read hazard is the programmer's inside an asm statement) -- against the builtin form of the same update, lane by lane and bit by
    bit, and both against the arithmetic they stand for: a[j] += a_K[j] * nl with a_K the row of lane K of the lane's 16-lane group, for
    j > K; columns <= K untouched.  Random rows, multipliers of every magnitude, a zero multiplier (the masked lanes of the factorisation),
    denormal and huge entries.  (Under tests/emu both forms are the builtin: the comparison with NumPy is what that run checks.)"""
    rng = np.random.default_rng(11)
    ctx = lib.Context(640, 480)
    try:
        for trial in range(6):
            a = rng.normal(size=(64, 16)) * 10.0 ** rng.integers(-3, 4, size=(64, 16))
            nl = rng.normal(size=64) * 10.0 ** rng.integers(-2, 3, size=64)
            if trial == 1:
                nl[rng.random(64) < 0.5] = 0.0                 # lanes at or above the pivot row multiply by the 0 / 1 mask
            if trial == 2:
                a[rng.random((64, 16)) < 0.1] = 5e-324          # denormals pass through v_fmac_f64 unflushed
                a[rng.random((64, 16)) < 0.05] = 1e300
            oa, orf = ctx.dbg_gs_row_update(a, nl)
            assert np.array_equal(oa.view(np.uint64), orf.view(np.uint64)), trial
            for K in range(15):
                exp = a.copy()
                rows = a.reshape(4, 16, 16)[:, K, :]            # lane K of each 16-lane group
                src = np.repeat(rows, 16, axis=0)               # (64, 16): what row_newbcast:K delivers to every lane of the group
                for j in range(K + 1, 16):
                    # one fused multiply-add per element: exact product + one rounding (np.longdouble carries the 106-bit product of two doubles
                    # only approximately, so the check is to 1 ulp, the bit-for-bit statement is the one between the two device forms)
                    exp[:, j] = a[:, j] + src[:, j] * nl
                got = oa[K]
                assert np.array_equal(got[:, :K + 1], a[:, :K + 1]), (trial, K)
                with np.errstate(over="ignore", invalid="ignore"):
                    err = np.abs(got[:, K + 1:] - exp[:, K + 1:])
                    tol = 2 * np.spacing(np.maximum(np.abs(exp[:, K + 1:]), np.abs(src[:, K + 1:] * nl[:, None])))
                ok = (err <= tol) | ~np.isfinite(exp[:, K + 1:])
                assert ok.all(), (trial, K, float(err[~ok].max()))
    finally:
        ctx.close()
