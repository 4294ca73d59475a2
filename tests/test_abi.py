"""The C-ABI libraries load on a GPU-less box and export every symbol the headers declare; creating a
context without a GPU fails loudly (there is no CPU fallback in the product)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sos[f]?_[a-zA-Z0-9_]+)\s*\(", txt)) - {"sosf_allreduce_fn", "sosf_nth_fn"})


def test_hip_library_exports_header_symbols():
    from sos_slam_amd import lib
    L = lib.load()
    names = _declared("sos_slam.h")
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), n
    assert set(lib.SYMBOLS) <= set(names)
    assert L.sos_backend_name() == b"hip-gfx950"


def test_host_library_exports_header_symbols():
    from sos_slam_amd import host
    L = host.load()
    for n in _declared("sos_slam_host.h"):
        assert hasattr(L, n), n


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sos_slam_amd import host, lib, synth
    with pytest.raises(lib.SosError):
        lib.Context(64, 64)
    with pytest.raises(lib.SosError):
        host.System(synth.default_params(64, 64))


def test_record_layouts_match_header():
    """numpy / ctypes mirrors have the sizes the C structs have."""
    from sos_slam_amd import records, synth
    assert synth.POINT_DTYPE.itemsize == 96
    assert synth.RESID_DTYPE.itemsize == 24
    assert synth.PRECALC_DTYPE.itemsize == 112
    assert synth.RAWJAC_DTYPE.itemsize == 74 * 4
    assert synth.FRAME_INIT_DTYPE.itemsize == 12 * 8 + 10 * 8 + 16
    assert C.sizeof(records.Params) == 18 * 4
    assert C.sizeof(records.Calib) == 32


def test_record_sizes_against_the_compiled_header(tmp_path):
    """sizeof of every record of include/sos_slam.h, as gcc lays it out, against the Python mirrors."""
    import subprocess
    from sos_slam_amd import records, synth
    mirrors = {"sos_point": synth.POINT_DTYPE.itemsize, "sos_resid": synth.RESID_DTYPE.itemsize,
               "sos_precalc": synth.PRECALC_DTYPE.itemsize, "sos_rawjac": synth.RAWJAC_DTYPE.itemsize,
               "sos_params": C.sizeof(records.Params), "sos_calib": C.sizeof(records.Calib),
               "sos_trace_params": C.sizeof(records.TraceParams), "sos_immature": records.IMMATURE_DTYPE.itemsize,
               "sos_activate_params": C.sizeof(records.ActivateParams), "sos_pair_tfm": records.PAIR_TFM_DTYPE.itemsize,
               "sos_activation": records.ACTIVATION_DTYPE.itemsize, "sos_pixsel_params": C.sizeof(records.PixselParams),
               "sos_camera_model": C.sizeof(records.CameraModel), "sos_resid_final": records.RESID_FINAL_DTYPE.itemsize}
    src = tmp_path / "sizes.c"
    mirrors.update({"sosf_sequence_params": C.sizeof(records.SequenceParams), "sosf_frame_result": C.sizeof(records.FrameResult),
                    "sosf_frame_extra": C.sizeof(records.FrameExtra), "sosf_imu_frame": C.sizeof(records.ImuFrame),
                    "sosf_imu_calib": C.sizeof(records.ImuCalib), "sosf_imu_settings": C.sizeof(records.ImuSettings),
                    "sosf_imu_shell": C.sizeof(records.ImuShell)})
    src.write_text('#include <stdio.h>\n#include "sos_slam.h"\n#include "sos_slam_host.h"\nint main(void) {\n' +
                   "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in mirrors) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, size in mirrors.items():
        assert int(got[n]) == size, (n, got[n], size)


def test_poses_file_format(tmp_path):
    """poses.txt as LoopHandler::savePose writes it (src/LoopClosure/LoopHandler.cpp:62-76): default-float, 6 digits."""
    import numpy as np
    from sos_slam_amd import host
    ids = [3, 17, 250]
    t = np.array([[0.0, -1.5, 123456.789], [1e-7, 0.333333333, -2.0000004], [12.5, 1e10, -0.000123456789]])
    host.write_poses(tmp_path / "poses.txt", ids, t)
    lines = (tmp_path / "poses.txt").read_text().splitlines()
    assert lines == ["3 0 -1.5 123457", "17 1e-07 0.333333 -2", "250 12.5 1e+10 -0.000123457"]


def test_facade_ldlt_variants_agree():
    """The blocked LDL^T on the GN critical path against the unblocked reference variant and numpy, including a
    singular matrix (exact-zero pivots contribute nothing, as with Eigen's ldlt().solve) -- host code, no GPU."""
    import numpy as np
    from sos_slam_amd import host
    rng = np.random.default_rng(5)
    for n in (4, 12, 13, 37, 100, 132):
        B = rng.normal(size=(n, n + 3))
        sc = 1 + 30.0 * (np.arange(n) % 7 == 0)
        A = (B @ B.T) * sc[:, None] * sc[None, :]
        b = rng.normal(size=n)
        x0, x1 = host.ldlt_solve(A, b, 0), host.ldlt_solve(A, b, 1)
        xr = np.linalg.solve(A, b)
        assert np.abs(x0 - xr).max() < 1e-9 * np.abs(xr).max()
        assert np.abs(x0 - x1).max() < 1e-10 * np.abs(xr).max()
        if n > 6:
            A2 = A.copy()
            A2[5, :] = 0
            A2[:, 5] = 0
            y0, y1 = host.ldlt_solve(A2, b, 0), host.ldlt_solve(A2, b, 1)
            assert y0[5] == 0 and y1[5] == 0
            keep = np.arange(n) != 5
            yr = np.linalg.solve(A2[np.ix_(keep, keep)], b[keep])
            assert np.abs(y0[keep] - yr).max() < 1e-9 * np.abs(yr).max()


_ZERO_CALLER = r'''
import ctypes as C, re, sys
hip = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)
libs = {"sos_slam.h": hip, "sos_slam_host.h": C.CDLL(sys.argv[2], mode=C.RTLD_GLOBAL)}
n = 0
for hdr, L in libs.items():
    txt = re.sub(r"/\*.*?\*/", " ", open(sys.argv[3] + "/" + hdr).read(), flags=re.S)
    for m in re.finditer(r"\bint\s+(sosf?_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        vals = []
        for a in ([] if args in ("", "void") else args.split(",")):
            a = a.strip()
            if "*" in a or "[" in a: vals.append(C.c_void_p(None))
            elif re.match(r"(const\s+)?float\b", a): vals.append(C.c_float(0))
            elif re.match(r"(const\s+)?double\b", a): vals.append(C.c_double(0))
            else: vals.append(C.c_int(0))
        f = getattr(L, name)
        f.restype = C.c_int
        print(name, f(*vals), flush=True)
        n += 1
print("CALLED", n)
'''


def test_entry_points_survive_zero_arguments():
    """Every `int` entry point of the two headers called with NULL for every pointer and 0 for every scalar: a status comes back
    (SOS_ERR_ARG / SOS_ERR_STATE for all but a few queries that have an answer for 'nothing'), nothing dereferences, nothing aborts.
    Runs in a child process so that a crash is a test failure, not the end of the session; needs no GPU."""
    import subprocess
    import sys
    from sos_slam_amd import host, lib
    lib.load(), host.load()
    p = subprocess.run([sys.executable, "-c", _ZERO_CALLER, lib.lib_path(), os.path.join(os.path.dirname(lib.lib_path()), "libsos_host.so"),
                        os.path.join(ROOT, "include")], capture_output=True, text=True, timeout=300)
    lines = p.stdout.strip().splitlines()
    assert p.returncode == 0, (p.returncode, lines[-3:], p.stderr[-500:])
    assert lines[-1].startswith("CALLED") and int(lines[-1].split()[1]) >= 140, lines[-1]
    rc = {ln.split()[0]: int(ln.split()[1]) for ln in lines[:-1]}
    ok_zero = {"sos_ba_gn_resident_supported", "sos_rccl_load", "sos_comm_size", "sos_comm_rank", "sosf_get_timing",
               "sosf_imu_solve_mode", "sosf_imu_solve_stats", "sosf_get_host_threads"}   # (a getter without a handle; mode 0 = the literal form is a valid selection; the counters may go nowhere)
    bad = {k: v for k, v in rc.items() if v not in (-1, -3) and k not in ok_zero and not (k.endswith("_destroy") and v == 0)}   # destroying nothing is not an error
    assert not bad, bad
