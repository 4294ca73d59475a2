#!/usr/bin/env python3
"""ISA lint over the gfx950 assembly of every kernel of libsos_slam_hip.so (no GPU needed).

    python tools/isa_lint.py                 # compiles the seven sources with sos_slam_amd/build.py's flags into a temp dir
    python tools/isa_lint.py --keep /tmp/isa # ... and keeps the .s files there (reused when newer than the sources)

Two classes of defect that neither the compiler nor tests/emu's fiber emulation reports:

B  `s_barrier` where EXEC is not provably the mask the wave started with.  s_barrier counts WAVES: a barrier under a lane-divergent
   guard is executed once by a wave that has lanes on both sides of an `if / else` with a barrier in each arm twice (the wave falls
   one barrier out of step with its siblings: round 5's k_gn_solve), and not at all by a wave whose lanes all fail the guard.
   Method: forward data-flow over the control-flow graph of each kernel with EXEC as a symbolic value -- FULL at entry; every
   narrowing (s_and_saveexec, s_and / s_andn2 / s_xor on exec) makes a fresh value that remembers what it is a subset of; scalar
   register pairs carry the exec values saved into them, the complement `s_xor sX, exec, sX` builds for the else arm, and for the
   break masks of divergent loops the invariant (sX | exec) == u; `s_or_b64 exec, exec, sX` restores exactly when those facts
   prove it.  At a join the facts of all predecessors must agree.  Anything unproved is reported -- the analysis errs towards
   reporting.
D  a DPP instruction whose DPP source VGPR (src0) is written by a VALU instruction fewer than 2 wait states in front of it (s_nop N
   = N + 1 wait states), or that follows a v_cmpx by fewer than 5: the hazard the assembler does not pad inside an `asm` block
   (sos_gn_resident.inc: gs_row_update<K>; sos_ba.hip: seqsum8).

Exit status 1 when anything is reported that ALLOW (below) does not name with its reason.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["sos_ctx", "sos_ba", "sos_tracker", "sos_comm", "sos_immature", "sos_pixsel", "sos_undistort"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-value", "-mllvm",
         "-amdgpu-kernarg-preload-count=16"]          # sos_slam_amd/build.py

# (kernel substring, class) -> why the report is not a defect.  Empty on purpose: a barrier the analysis cannot prove is rewritten.
ALLOW = {}

FULL = ("full",)
ZERO = ("zero",)


PHI_PARENT = {}   # ('phi', block) -> the value every predecessor's exec is a subset of (per kernel; reset by lint_barriers)


def parent(a):
    if a[0] == "sub":
        return a[2]
    if a[0] == "phi":
        return PHI_PARENT.get(a)
    if a[0] == "diff":
        return a[1]
    return None


def subset_of(a, u):
    n = 0
    while a is not None and n < 10000:
        if a == u or u == FULL:
            return True
        a = parent(a)
        n += 1
    return False


def v_or(a, b):
    if a is None or b is None:
        return None
    if a == ZERO:
        return b
    if b == ZERO:
        return a
    if subset_of(a, b):
        return b
    if subset_of(b, a):
        return a
    if a[0] == "diff" and a[2] == b:
        return a[1]
    if b[0] == "diff" and b[2] == a:
        return b[1]
    return None


SREG = re.compile(r"^s(\d+)$|^s\[(\d+):(\d+)\]$")
VREG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def reg_range(op, rx):
    m = rx.match(op)
    if not m:
        return None
    if m.group(1) is not None:
        return (int(m.group(1)), int(m.group(1)))
    return (int(m.group(2)), int(m.group(3)))


NO_SDST = ("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_setprio", "s_sleep", "s_sendmsg",
           "s_store", "s_buffer_store", "s_dcache", "s_icache", "s_setreg", "s_code_end", "s_trap", "s_sethalt", "s_inst_prefetch",
           "s_clause", "s_delay", "s_wakeup", "s_setvskip", "s_set_gpr", "s_scratch_store")


class Fn:
    def __init__(self, name):
        self.name = name
        self.ins = []     # (lineno, mnemonic, [operands], raw)
        self.labels = {}  # label -> instruction index


def parse(path):
    fns, cur = [], None
    for no, raw in enumerate(open(path, errors="replace"), 1):
        ln = raw.split(";")[0].rstrip()
        if not ln.strip():
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", ln)
        if m and not ln.startswith(".L"):
            cur = Fn(m.group(1))
            fns.append(cur)
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.L\w+):", ln)
        if m:
            if m.group(1).startswith(".Lfunc_end"):
                cur = None
            else:
                cur.labels[m.group(1)] = len(cur.ins)
            continue
        s = ln.strip()
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        ops = []
        if len(parts) > 1:
            # operands: comma separated; modifiers (row_newbcast:3 ...) trail the last one, space separated
            toks = [t.strip() for t in parts[1].split(",")]
            for i, t in enumerate(toks):
                ops.append(t.split()[0] if t else t)
            tail = toks[-1].split()[1:] if toks and toks[-1] else []
        else:
            tail = []
        cur.ins.append((no, mn, ops, s, tail))
    return [f for f in fns if f.ins]


def blocks_of(fn):
    n = len(fn.ins)
    leaders = {0} | set(fn.labels.values())
    for i, (_, mn, ops, _, _) in enumerate(fn.ins):
        if mn.startswith("s_cbranch") or mn in ("s_branch", "s_endpgm", "s_setpc_b64"):
            leaders.add(i + 1)
    leaders = sorted(x for x in leaders if x < n)
    start_to_b = {s: k for k, s in enumerate(leaders)}
    blocks = []
    for k, s in enumerate(leaders):
        e = leaders[k + 1] if k + 1 < len(leaders) else n
        blocks.append([s, e, []])
    for k, (s, e, succ) in enumerate(blocks):
        _, mn, ops, _, _ = fn.ins[e - 1]
        if mn == "s_endpgm" or mn == "s_setpc_b64":
            continue
        if mn == "s_branch":
            succ.append(start_to_b[fn.labels[ops[0]]])
            continue
        if mn.startswith("s_cbranch"):
            succ.append(start_to_b[fn.labels[ops[0]]])
        if k + 1 < len(blocks):
            succ.append(k + 1)
    return blocks


class State:
    __slots__ = ("exec", "holds", "rel")

    def __init__(self):
        self.exec = FULL
        self.holds = {}   # (lo, hi) -> value
        self.rel = {}     # (lo, hi) -> (e, u): (sX | e) == u, usable while exec == e

    def copy(self):
        s = State()
        s.exec, s.holds, s.rel = self.exec, dict(self.holds), dict(self.rel)
        return s

    def key(self):
        return (self.exec, tuple(sorted(self.holds.items())), tuple(sorted(self.rel.items())))

    def clobber(self, rng):
        for d in (self.holds, self.rel):
            for k in [k for k in d if not (k[1] < rng[0] or k[0] > rng[1])]:
                del d[k]


def is_exec(op):
    return op == "exec"


def step(st, idx, ins):
    """transfer function of one instruction; returns True when the instruction is an s_barrier reached with exec unproved"""
    _, mn, ops, _, _ = ins
    if mn == "s_barrier":
        return st.exec != FULL
    if mn.startswith(NO_SDST):
        return False
    if mn.startswith("v_cmpx") or (ops and ops[0] in ("exec_lo", "exec_hi")):
        st.exec = ("sub", idx, st.exec) if mn.startswith("v_cmpx") else None
        return False
    m = re.match(r"^s_(and|or|xor|andn2|orn2|nand|nor|xnor|andn1|orn1)_saveexec_b64$", mn)
    if m:
        dst, src = reg_range(ops[0], SREG), ops[1]
        old = st.exec
        sv = st.holds.get(reg_range(src, SREG)) if reg_range(src, SREG) else None
        op = m.group(1)
        if op == "and":
            new = ("sub", idx, old)
        elif op == "or":
            new = v_or(old, sv)
        elif op == "andn2":        # exec = src & ~exec
            new = sv if (sv is not None and sv[0] == "diff" and sv[2] == old) else (("sub", idx, sv) if sv is not None else None)
        else:
            new = None
        if dst:
            st.clobber(dst)
            if old is not None:
                st.holds[dst] = old
        st.exec = new
        return False
    if ops and is_exec(ops[0]):
        a = ops[1] if len(ops) > 1 else None
        b = ops[2] if len(ops) > 2 else None
        if mn == "s_mov_b64":
            r = reg_range(a, SREG)
            st.exec = st.holds.get(r) if r else (FULL if a == "-1" and False else None)
            return False
        other = b if is_exec(a) else (a if is_exec(b) else None)
        r = reg_range(other, SREG) if other else None
        if mn == "s_or_b64" and other is not None:
            rel = st.rel.get(r) if r else None
            if rel is not None and rel[0] == st.exec:
                st.exec = rel[1]
            else:
                st.exec = v_or(st.exec, st.holds.get(r) if r else None)
            return False
        if mn == "s_andn2_b64" and is_exec(a) and r is not None:
            rel = st.rel.get(r)
            new = ("sub", idx, st.exec)
            if rel is not None and rel[0] == st.exec:
                st.rel[r] = (new, rel[1])
            st.exec = new
            return False
        if mn == "s_xor_b64" and other is not None:
            h = st.holds.get(r) if r else None
            # exec ^ w with w a subset of exec: the rest;  exec ^ u with exec a subset of u: u \ exec
            if h is not None and st.exec is not None and subset_of(h, st.exec):
                st.exec = ("diff", st.exec, h)
            elif h is not None and st.exec is not None and subset_of(st.exec, h):
                st.exec = ("diff", h, st.exec)
            else:
                st.exec = None
            return False
        if mn == "s_and_b64" and other is not None:
            st.exec = ("sub", idx, st.exec) if st.exec is not None else None
            return False
        st.exec = None
        return False
    # writes to scalar registers
    if not ops:
        return False
    if mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_div_scale")) and len(ops) > 1:
        co = reg_range(ops[1], SREG)              # the carry-out / scale flag is the SECOND operand
        if co:
            st.clobber(co)
        return False
    dst = reg_range(ops[0], SREG)
    if dst is None:
        return False
    a = ops[1] if len(ops) > 1 else None
    b = ops[2] if len(ops) > 2 else None
    ra, rb = (reg_range(a, SREG) if a else None), (reg_range(b, SREG) if b else None)
    new_h, new_rel = None, None
    if mn == "s_mov_b64":
        if a == "exec":
            new_h = st.exec
        elif a == "0":
            new_h = ZERO
            if st.exec is not None:
                new_rel = (st.exec, st.exec)
        elif ra:
            new_h = st.holds.get(ra)
            new_rel = st.rel.get(ra)
    elif mn == "s_xor_b64" and (a == "exec" or b == "exec"):
        h = st.holds.get(rb if a == "exec" else ra)
        if h is not None and st.exec is not None and subset_of(st.exec, h):
            new_h = ("diff", h, st.exec)
    elif mn == "s_and_b64" and (a == "exec" or b == "exec"):
        if st.exec is not None:
            new_h = ("sub", idx, st.exec)
    elif mn.startswith("v_cmp") and not mn.startswith("v_cmpx"):
        if st.exec is not None:                   # a compare writes 0 for the inactive lanes: a subset of exec
            new_h = ("sub", idx, st.exec)
    elif mn in ("s_and_b64", "s_andn2_b64"):      # a subset of either operand (andn2: of the first)
        h = st.holds.get(ra) if ra else None
        if (h is None or h == ZERO) and mn == "s_and_b64":
            h = st.holds.get(rb) if rb else None
        if h is not None and h != ZERO:
            new_h = ("sub", idx, h)
    elif mn == "s_andn2_b64" and b == "exec" and ra:
        h = st.holds.get(ra)                      # h & ~exec
        if h is not None and st.exec is not None and subset_of(st.exec, h):
            new_h = ("diff", h, st.exec)
    elif mn == "s_or_b64" and (ra == dst or rb == dst):
        # the break mask of a divergent loop grows: (sX | exec) == u survives when what is or-ed in lies inside u
        other, ro = (b, rb) if ra == dst else (a, ra)
        rel = st.rel.get(dst)
        if rel is not None:
            ho = st.holds.get(ro) if ro else None
            if rel[1] == FULL or other == "vcc" or (ho is not None and subset_of(ho, rel[1])):
                new_rel = rel
        h, ho = st.holds.get(dst), (st.holds.get(ro) if ro else None)
        new_h = v_or(h, ho) if other != "vcc" else None
    st.clobber(dst)
    if new_h is not None:
        st.holds[dst] = new_h
    if new_rel is not None:
        st.rel[dst] = new_rel
    return False


def merge(preds, bidx, npreds=None, forced=()):
    """join of the predecessors' out-states (those computed so far; npreds = how many the block has).  Where they disagree about
    exec the block gets its own value ('phi', block); a block in `forced` gets it from its first visit on, so that the names inside a
    loop do not change between the rounds of the iteration (lint_barriers restarts with the joins the previous run ended with)."""
    st = State()
    own = ("phi", bidx)
    ex = [p.exec for p in preds]
    others = [e for e in ex if e != own]
    if any(e is None for e in ex):
        st.exec = None
    elif others and all(e == others[0] for e in others) and not (bidx in forced and npreds is not None and len(preds) < npreds):
        st.exec = others[0]      # agreement (a predecessor that carries this block's own value round a loop changes nothing)
    elif not others:
        st.exec = own
    else:
        par = others[0]          # the nearest value up the chain that contains every predecessor's exec
        n = 0
        while par is not None and n < 10000 and not all(subset_of(e, par) for e in others):
            par = parent(par)
            n += 1
        if par == own:
            par = None
        PHI_PARENT[own] = par
        st.exec = own
    keys = set(preds[0].holds)
    for p in preds[1:]:
        keys &= set(p.holds)
    for k in keys:
        v = preds[0].holds[k]
        if all(p.holds[k] == v for p in preds):
            st.holds[k] = v
    keys = set(preds[0].rel)
    for p in preds[1:]:
        keys &= set(p.rel)
    for k in keys:
        r0 = preds[0].rel[k]
        if st.exec is not None and all(p.rel[k][1] == r0[1] and p.rel[k][0] == p.exec for p in preds):
            st.rel[k] = (st.exec, r0[1])        # true of every predecessor's exec, hence of the joined one
        elif all(p.rel[k] == r0 for p in preds):
            st.rel[k] = r0                      # (about an exec value the flow returns to later: kept as it is)
    return st


def fmt(v):
    if v is None:
        return "?"
    if v[0] == "sub":
        return "sub%d<%s" % (v[1], fmt(v[2]))
    if v[0] == "phi":
        return "phi%d" % v[1]
    if v[0] == "diff":
        return "(%s \\ %s)" % (fmt(v[1]), fmt(v[2]))
    return v[0]


def lint_barriers(fn, trace=False):
    blocks = blocks_of(fn)
    preds = [[] for _ in blocks]
    for k, (_, _, succ) in enumerate(blocks):
        for s in succ:
            preds[s].append(k)
    # reverse post-order: every round visits a block after all its forward predecessors, so that a join only ever mixes this round's
    # facts with the previous round's along back edges (mixing older facts loses relations for good)
    seen, post = set(), []
    stack = [(0, iter(blocks[0][2]))]
    seen.add(0)
    while stack:
        k, it = stack[-1]
        nxt = next((t for t in it if t not in seen), None)
        if nxt is None:
            post.append(k)
            stack.pop()
        else:
            seen.add(nxt)
            stack.append((nxt, iter(blocks[nxt][2])))
    rpo = post[::-1]
    np_ = [len(preds[k]) + (1 if k == 0 else 0) for k in range(len(blocks))]
    forced = set()
    for attempt in range(12):
        PHI_PARENT.clear()
        out = [None] * len(blocks)
        flagged = {}
        joins = set()
        for rounds in range(64):
            changed = False
            for k in rpo:
                ps = [out[p] for p in preds[k] if out[p] is not None]
                if k == 0:
                    ps = ps + [State()]
                if not ps:
                    continue
                st = merge(ps, k, np_[k], forced)
                (joins.add if st.exec == ("phi", k) else joins.discard)(k)
                s, e, succ = blocks[k]
                for i in range(s, e):
                    bad = step(st, i, fn.ins[i])
                    if fn.ins[i][1] == "s_barrier":
                        flagged[i] = bad
                if out[k] is None or out[k].key() != st.key():
                    out[k] = st
                    changed = True
            if not changed:
                break
        else:
            return [(fn.ins[0][0], "analysis did not converge")]
        if joins == forced:
            break
        forced = set(joins)   # run again with these blocks named from their first visit on
    if trace:   # exec at the head and the tail of every block, for reading a report against the .s
        lab = {v: k for k, v in fn.labels.items()}
        for k in rpo:
            ps = [out[p] for p in preds[k] if out[p] is not None] + ([State()] if k == 0 else [])
            print("  block %d %s line %d: exec in %s  out %s" % (k, lab.get(blocks[k][0], ""), fn.ins[blocks[k][0]][0], fmt(merge(ps, k, np_[k], forced).exec), fmt(out[k].exec)))
    return [(fn.ins[i][0], "s_barrier with EXEC not provably the entry mask") for i, bad in sorted(flagged.items()) if bad]


def lint_dpp(fn):
    rep = []
    for i, (no, mn, ops, raw, tail) in enumerate(fn.ins):
        isdpp = mn.endswith("_dpp") or any(t.startswith(("row_", "quad_perm", "wave_", "row_newbcast")) for t in tail)
        if not isdpp or len(ops) < 2:
            continue
        src = reg_range(ops[1], VREG)
        if src is None:
            continue
        ws = 0
        j = i - 1
        while j >= 0 and ws < 5:
            _, pm, pops, _, _ = fn.ins[j]
            if pm.startswith("v_cmpx"):
                rep.append((no, "DPP %d wait states after v_cmpx (5 needed): %s" % (ws, raw)))
            if ws < 2 and pm.startswith("v_") and not pm.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and pops:
                d = reg_range(pops[0], VREG)
                if d and not (d[1] < src[0] or d[0] > src[1]):
                    rep.append((no, "DPP source %s written %d wait states earlier by `%s` (2 needed): %s" % (ops[1], ws, fn.ins[j][3], raw)))
            ws += (int(pops[0], 0) + 1) if pm == "s_nop" and pops else 1
            j -= 1
    return rep


def compile_all(outdir):
    procs = []
    for f in SRC:
        src = os.path.join(ROOT, "sos_slam_amd", "csrc", f + ".hip")
        dst = os.path.join(outdir, f + ".s")
        deps = [src] + [os.path.join(ROOT, "sos_slam_amd", "csrc", x) for x in os.listdir(os.path.join(ROOT, "sos_slam_amd", "csrc")) if x.endswith((".h", ".inc"))]
        if os.path.exists(dst) and all(os.path.getmtime(dst) > os.path.getmtime(d) for d in deps):
            continue
        procs.append((f, subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", dst, src], stderr=subprocess.PIPE, text=True)))
    for f, p in procs:
        err = p.communicate()[1]
        if p.returncode != 0:
            sys.exit("hipcc failed on %s:\n%s" % (f, err[-2000:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", help="directory for the .s files (reused when newer than the sources)")
    ap.add_argument("--trace", help="print the exec value at every block of the kernels whose name contains this")
    ap.add_argument("files", nargs="*", help="lint these .s files instead of compiling the product sources")
    a = ap.parse_args()
    tmp = None
    if a.files:
        paths = a.files
    else:
        d = a.keep
        if d is None:
            tmp = tempfile.TemporaryDirectory()
            d = tmp.name
        os.makedirs(d, exist_ok=True)
        compile_all(d)
        paths = [os.path.join(d, f + ".s") for f in SRC]
    nk = nb = nd = 0
    bad = 0
    for p in paths:
        for fn in parse(p):
            if not any(mn == "s_endpgm" for _, mn, _, _, _ in fn.ins):
                continue
            nk += 1
            nb += sum(1 for x in fn.ins if x[1] == "s_barrier")
            nd += sum(1 for x in fn.ins if x[1].endswith("_dpp"))
            for cls, rep in (("B", lint_barriers(fn, bool(a.trace) and a.trace in fn.name)), ("D", lint_dpp(fn))):
                for no, msg in rep:
                    why = next((w for (k, c), w in ALLOW.items() if c == cls and k in fn.name), None)
                    print("%s %s:%d %s: %s%s" % (cls, os.path.basename(p), no, fn.name, msg, "   [allowed: %s]" % why if why else ""))
                    bad += 0 if why else 1
    print("# %d kernels, %d s_barrier, %d DPP instructions checked; %d report(s)" % (nk, nb, nd, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
