"""tests/emu/build_emu.py -- TEST INFRASTRUCTURE: builds the lockstep CPU emulation of the device library.

    python tests/emu/build_emu.py          ->  tests/emu/_build/libsos_slam_hip.so  (+ libsos_host.so linked against it)

The product sources under sos_slam_amd/csrc are read, never changed: three purely textual rewrites make them host C++ for the
emulator's <hip/hip_runtime.h> (tests/emu/include):

  1. k<<<grid, block, lds, stream>>>(args)   ->  emu::launch("k", k, grid, block, lds, stream, args)
  2. __shared__ T name[N];                   ->  a reference to per-workgroup storage (several workgroups are resident at once)
     extern __shared__ T name[];             ->  the launch's dynamic LDS
  3. the inline assembly: seqsum8's v_add_f32_dpp chain -> emu::seqsum8, the same chain on the emulated lanes; the GS_ROW_UPDATE macro
     (gs_row_update<K>'s v_fmac_f64_dpp statements) -> gs_row_update_ref<K>, the builtin form of the same update

Everything else (kernels, device functions, the C-ABI, the host-side stream / flag logic) is compiled as written.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "sos_slam_amd", "csrc")
ASAN = os.environ.get("EMU_ASAN") == "1"   # AddressSanitizer build (run with LD_PRELOAD=<asan_runtime()> ASAN_OPTIONS=detect_leaks=0)
OUT = os.path.join(HERE, ("_build_asan" if ASAN else "_build") + os.environ.get("EMU_TAG", ""))   # EMU_TAG: a second build beside the first (e.g. EMU_OPT=-O1)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

sys.path.insert(0, ROOT)
from sos_slam_amd import build as _b  # noqa: E402  (source lists and host flags of the product build)


def _match_back_angle(s, i):
    """s[i] == '>' : index of the matching '<'."""
    depth = 0
    while i >= 0:
        if s[i] == ">":
            depth += 1
        elif s[i] == "<":
            depth -= 1
            if depth == 0:
                return i
        i -= 1
    raise ValueError("unbalanced template brackets before <<<")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x.strip() for x in out]


def rewrite_launches(s):
    out, pos = "", 0
    while True:
        i = s.find("<<<", pos)
        if i < 0:
            return out + s[pos:]
        # kernel expression before <<<
        j = i - 1
        while s[j].isspace():
            j -= 1
        if s[j] == ">":
            j = _match_back_angle(s, j) - 1
            while s[j].isspace():
                j -= 1
        k = j
        while k >= 0 and (s[k].isalnum() or s[k] == "_" or s[k] == ":"):
            k -= 1
        kern = s[k + 1:i].strip()
        e = s.find(">>>", i)
        cfg = _split_top(s[i + 3:e])
        while len(cfg) < 4:
            cfg.append("0")
        a = e + 3
        while s[a].isspace():
            a += 1
        assert s[a] == "(", "launch without an argument list near: " + s[i - 40:i + 40]
        depth, b = 0, a
        while True:
            if s[b] == "(":
                depth += 1
            elif s[b] == ")":
                depth -= 1
                if depth == 0:
                    break
            b += 1
        args = s[a + 1:b].strip()
        name = re.sub(r"\s+", "", kern)
        call = 'emu::launch("%s", [](auto... emu_a) { %s(emu_a...); }, dim3(%s), dim3(%s), (size_t)(%s), (hipStream_t)(%s)%s)' % (
            name, kern, cfg[0], cfg[1], cfg[2], cfg[3], (", " + args) if args else "")
        out += s[pos:k + 1] + call
        pos = b + 1


_KEY = "([]{ static char k; return (const void *)&k; }())"


def rewrite_shared(s):
    def repl(m):
        ext, body = m.group(1), m.group(2).strip()
        body = re.sub(r"__attribute__\(\(aligned\(\d+\)\)\)", "", body).strip()
        if ext:
            mm = re.match(r"^(.*?)(\w+)\s*\[\s*\]$", body, re.S)
            ty, name = mm.group(1).strip(), mm.group(2)
            return "%s *%s = reinterpret_cast<%s *>(emu::dyn_shared());" % (ty, name, ty)
        parts = _split_top(body)
        mm = re.match(r"^(.*?)(\w+)\s*((?:\[[^\]]*\]\s*)*)$", parts[0], re.S)
        ty = mm.group(1).strip()
        decls = [(mm.group(2), mm.group(3).strip())] + [re.match(r"^(\w+)\s*((?:\[[^\]]*\]\s*)*)$", p, re.S).groups() for p in parts[1:]]
        res = []
        for name, dims in decls:
            dims = dims.strip()
            if dims:
                res.append("%s (&%s)%s = *emu::shared_static<%s%s>(%s);" % (ty, name, dims, ty, dims, _KEY))
            else:
                res.append("%s &%s = *emu::shared_static<%s>(%s);" % (ty, name, ty, _KEY))
        return " ".join(res)

    return re.sub(r"(extern\s+)?__shared__\s+([^;]*);", repl, s)


def rewrite_asm(s, fname):
    # csrc/sos_gn_resident.inc: gs_row_update<K>'s asm statements (v_fmac_f64_dpp ... row_newbcast:K, one per column, generated by the
    # GS_ROW_UPDATE macro) -> the builtin form the product keeps beside them for exactly this comparison (gs_row_update_ref<K>)
    s = re.sub(r"#define GS_ROW_UPDATE\(K, J\) asm volatile\([^\n]*\)\n", "#define GS_ROW_UPDATE(K, J) gs_row_update_ref<K>(a, nl)\n", s)

    def repl(m):
        if "v_add_f32_dpp" in m.group(0) and "row_shr:7" in m.group(0):
            return "s = emu::seqsum8(v);"
        raise SystemExit("tests/emu: unknown inline assembly in %s -- teach build_emu.py its meaning" % fname)

    return re.sub(r"asm\s+volatile\s*\((?:[^;]|\n)*?\)\s*;", repl, s)


def preprocess(src, dst):
    s = open(src).read()
    s = rewrite_asm(s, src)
    s = rewrite_launches(s)
    s = rewrite_shared(s)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    if not os.path.exists(dst) or open(dst).read() != s:
        open(dst, "w").write(s)


def asan_runtime():
    return subprocess.check_output([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    gen = os.path.join(OUT, "gen", "sos_slam_amd", "csrc")
    os.makedirs(gen, exist_ok=True)
    srcs = []
    for f in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, f)
        if f.endswith((".hip", ".inc", ".h")):
            d = os.path.join(gen, f[:-4] + ".cpp" if f.endswith(".hip") else f)
            preprocess(p, d)
            if f in _b.HIP_SOURCES:
                srcs.append(d)
    # the generated tree mirrors the product tree so that "../../include/sos_slam.h" resolves
    inc_link = os.path.join(OUT, "gen", "include")
    if not os.path.exists(inc_link):
        os.symlink(os.path.join(ROOT, "include"), inc_link)
    hip_lib = os.path.join(OUT, "libsos_slam_hip.so")
    host_lib = os.path.join(OUT, "libsos_host.so")
    rt = os.path.join(HERE, "emu_runtime.cpp")
    deps = [os.path.join(gen, f) for f in os.listdir(gen)] + [rt, os.path.join(HERE, "include", "hip", "hip_runtime.h"),
                                                              os.path.join(HERE, "include", "rccl", "rccl.h"), os.path.abspath(__file__)]
    opt = os.environ.get("EMU_OPT", "-O1" if ASAN else "-O2")
    flags = [opt, "-g1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread", "-ftls-model=initial-exec", "-fno-omit-frame-pointer",
             "-Wno-unused-value", "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-unused-function", "-Wno-ignored-attributes",
             "-I" + os.path.join(HERE, "include")]
    if ASAN:
        flags += ["-fsanitize=address", "-shared-libasan", "-fsanitize-address-use-after-return=never"]
    if force or _newer(hip_lib, deps):
        objs, jobs = [], []
        for sfile in srcs + [rt]:
            o = os.path.join(OUT, os.path.basename(sfile) + ".o")
            if force or _newer(o, deps):
                cmd = [CLANG] + flags + ["-c", sfile, "-o", o]
                if verbose:
                    print(" ".join(cmd))
                jobs.append(subprocess.Popen(cmd))
            objs.append(o)
        if any(j.wait() != 0 for j in jobs):
            raise SystemExit("tests/emu: compilation failed")
        # -Bsymbolic: the library's calls of hipMalloc / hipMemcpyAsync / ... bind to ITS definitions, also in a process that has the real
        # HIP run time loaded (the CPU suite loads the product library for its ABI checks)
        cmd = [CLANG, "-shared", "-pthread", "-Wl,-Bsymbolic", "-o", hip_lib] + objs + ["-ldl"] + (["-fsanitize=address", "-shared-libasan"] if ASAN else [])
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    hsrcs = [os.path.join(CSRC, s) for s in _b.HOST_SOURCES]
    hdeps = hsrcs + [hip_lib] + [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))]
    if force or _newer(host_lib, hdeps):
        cmd = ["g++"] + _b.CXX_FLAGS + ["-o", host_lib] + hsrcs + ["-L" + OUT, "-lsos_slam_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return hip_lib, host_lib


def build_selftest():
    """tests/emu/selftest/emu_selftest.hip through the same rewrites and run time -> a library of kernels with known answers"""
    out = os.path.join(OUT, "selftest")
    os.makedirs(out, exist_ok=True)
    src = os.path.join(HERE, "selftest", "emu_selftest.hip")
    gen = os.path.join(out, "emu_selftest.cpp")
    preprocess(src, gen)
    lib = os.path.join(out, "libemu_selftest.so")
    rt = os.path.join(HERE, "emu_runtime.cpp")
    deps = [gen, rt, os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    if _newer(lib, deps):
        cmd = [CLANG, "-O2", "-g1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread", "-shared", "-Wno-unused-value", "-Wno-unknown-pragmas",
               "-Wno-pass-failed", "-Wno-unused-function", "-Wl,-Bsymbolic", "-I" + os.path.join(HERE, "include"), gen, rt, "-o", lib, "-ldl"]
        subprocess.check_call(cmd)
    return lib


def build_fake_rccl():
    """tests/emu/fake_rccl.cpp -> a library with the six RCCL entry points csrc/sos_comm.hip binds, for ranks that are processes of the
    emulated device library (collectives through POSIX shared memory, enqueued in stream order).  sos_rccl_load() is given its path."""
    lib = os.path.join(OUT, "libfake_rccl.so")
    src = os.path.join(HERE, "fake_rccl.cpp")
    if _newer(lib, [src, os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "include", "rccl", "rccl.h")]):
        subprocess.check_call([CLANG, "-O2", "-g1", "-std=c++17", "-fPIC", "-pthread", "-shared", "-Wno-unused-function", "-I" + os.path.join(HERE, "include"), src, "-o", lib,
                               "-ldl", "-lrt"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
