"""tools/isa_lint.py: the two defect classes it exists for are reported, the shapes hipcc emits for sound code are not, and the
product's seven sources lint clean at HEAD (no GPU needed: hipcc cross-compiles gfx950)."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint as L  # noqa: E402

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def lint_text(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("k_test:\n" + textwrap.dedent(body) + "\n\ts_endpgm\n.Lfunc_end0:\n")
    fn = L.parse(str(p))[0]
    return L.lint_barriers(fn), L.lint_dpp(fn)


def test_barrier_in_both_arms_of_a_divergent_branch_is_reported(tmp_path):
    # round 5's k_gn_solve: `if (tid < N) { ...; __syncthreads(); ... } else { __syncthreads(); }`
    b, _ = lint_text(tmp_path, """
        v_cmp_gt_u32_e32 vcc, s4, v0
        s_and_saveexec_b64 s[0:1], vcc
        s_xor_b64 s[0:1], exec, s[0:1]
        s_cbranch_execz .LBB0_2
        v_mov_b32_e32 v1, 0
        s_barrier
    .LBB0_2:
        s_andn2_saveexec_b64 s[0:1], s[0:1]
        s_cbranch_execz .LBB0_4
        s_barrier
    .LBB0_4:
        s_or_b64 exec, exec, s[0:1]
        s_barrier
    """)
    assert len(b) == 2, b   # the two inside; the one after the reconvergence is proved


def test_structured_if_else_and_divergent_loop_reconverge(tmp_path):
    b, _ = lint_text(tmp_path, """
        v_cmp_gt_u32_e32 vcc, s4, v0
        s_and_saveexec_b64 s[0:1], vcc
        s_xor_b64 s[0:1], exec, s[0:1]
        s_cbranch_execz .LBB0_2
        v_mov_b32_e32 v1, 0
    .LBB0_2:
        s_or_saveexec_b64 s[0:1], s[0:1]
        s_xor_b64 exec, exec, s[0:1]
        s_cbranch_execz .LBB0_4
        v_mov_b32_e32 v1, 1
    .LBB0_4:
        s_or_b64 exec, exec, s[0:1]
        s_barrier
        v_cmp_gt_i32_e32 vcc, s4, v0
        s_and_saveexec_b64 s[0:1], vcc
        s_cbranch_execz .LBB0_8
        s_mov_b64 s[2:3], 0
    .LBB0_6:
        v_add_u32_e32 v1, 0x400, v1
        v_cmp_le_i32_e32 vcc, s4, v1
        s_or_b64 s[2:3], vcc, s[2:3]
        s_andn2_b64 exec, exec, s[2:3]
        s_cbranch_execnz .LBB0_6
        s_or_b64 exec, exec, s[2:3]
    .LBB0_8:
        s_or_b64 exec, exec, s[0:1]
        s_barrier
    """)
    assert b == []


def test_barrier_inside_a_divergent_loop_is_reported(tmp_path):
    b, _ = lint_text(tmp_path, """
        s_mov_b64 s[2:3], 0
    .LBB0_1:
        v_add_u32_e32 v1, 0x400, v1
        v_cmp_le_i32_e32 vcc, s4, v1
        s_barrier
        s_or_b64 s[2:3], vcc, s[2:3]
        s_andn2_b64 exec, exec, s[2:3]
        s_cbranch_execnz .LBB0_1
        s_or_b64 exec, exec, s[2:3]
        s_barrier
    """)
    assert len(b) == 1, b


def test_saved_mask_overwritten_before_the_restore_is_reported(tmp_path):
    b, _ = lint_text(tmp_path, """
        v_cmp_gt_u32_e32 vcc, s4, v0
        s_and_saveexec_b64 s[0:1], vcc
        s_mov_b32 s1, 0
        s_or_b64 exec, exec, s[0:1]
        s_barrier
    """)
    assert len(b) == 1, b


def test_dpp_source_hazard(tmp_path):
    _, d = lint_text(tmp_path, """
        v_mul_f64 v[2:3], v[4:5], v[6:7]
        v_fmac_f64_dpp v[0:1], v[2:3], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf
    """)
    assert len(d) == 1, d
    _, d = lint_text(tmp_path, """
        v_mul_f64 v[2:3], v[4:5], v[6:7]
        v_mov_b32_e32 v9, 0
        v_fmac_f64_dpp v[0:1], v[2:3], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf
    """)
    assert len(d) == 1, d   # one instruction between: one wait state, two needed
    _, d = lint_text(tmp_path, """
        v_mul_f64 v[2:3], v[4:5], v[6:7]
        s_nop 1
        v_fmac_f64_dpp v[0:1], v[2:3], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf
        v_fmac_f64_dpp v[10:11], v[10:11], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf
    """)
    assert d == [], d       # (the second reads a register nobody in front has written)
    _, d = lint_text(tmp_path, """
        v_fmac_f64_dpp v[0:1], v[0:1], v[8:9] row_newbcast:3 row_mask:0xf bank_mask:0xf
        v_fmac_f64_dpp v[0:1], v[0:1], v[8:9] row_newbcast:4 row_mask:0xf bank_mask:0xf
    """)
    assert len(d) == 1, d   # back-to-back on the same register


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_compiled_divergent_barrier_is_reported(tmp_path):
    src = tmp_path / "t.hip"
    src.write_text(textwrap.dedent("""
        #include <hip/hip_runtime.h>
        __global__ void k_bad(float *x, int n) {
          __shared__ float s[256];
          const int tid = threadIdx.x;
          if (tid < n) { s[tid] = x[tid]; __syncthreads(); x[tid] = s[(tid + 1) % n]; } else { __syncthreads(); }
        }
        __global__ void k_good(float *x, int n) {
          __shared__ float s[256];
          const int tid = threadIdx.x;
          float v = 0;
          if (tid < n) s[tid] = x[tid];
          __syncthreads();
          if (tid < n) v = s[(tid + 1) % n];
          for (int i = tid; i < n; i += 64) v += x[i];     // divergent trip count
          __syncthreads();
          if (tid < n) x[tid] = v;
        }
    """))
    out = tmp_path / "t.s"
    subprocess.check_call([HIPCC, "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out), str(src)], stderr=subprocess.DEVNULL)
    rep = {}
    for fn in L.parse(str(out)):
        if any(i[1] == "s_endpgm" for i in fn.ins):
            rep[fn.name] = L.lint_barriers(fn)
    bad = [k for k in rep if "k_bad" in k][0]
    good = [k for k in rep if "k_good" in k][0]
    assert len(rep[bad]) >= 1, rep
    assert rep[good] == [], rep


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_product_kernels_lint_clean(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), "--keep", str(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " 0 report(s)" in p.stdout
