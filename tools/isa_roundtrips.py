#!/usr/bin/env python3
"""Dependent global-memory round trips of every kernel, read off the gfx950 assembly (no GPU needed) -- the reading round 5 did by hand for
the accumulate chain and round 6 for k_track_lm, as a tool.

    python tools/isa_roundtrips.py [--root bisect/pending] [--keep /tmp/isa] [kernel-name-substring ...]
                                   (--root: another tree's sos_slam_amd/csrc, e.g. HEAD + the pending patches of tools/bisect_kit.sh)

At window sizes where a kernel is one partial round of blocks its duration is launch + the dependency chain of its slowest block, and
that chain is made of memory round trips: every `s_waitcnt vmcnt(N)` that a use sits behind ends a round.  hipcc puts such a wait
behind EVERY load that sits under a lane-divergent guard or in a loop with a run-time trip count, so source that looks batched
("request everything, then add") can be serialised again -- visible here as many rounds with one or two loads each.

Per kernel (static counts over the whole body; a round inside a loop counts once and is marked):
  loads        vector-memory loads (global_load_* / flat_load_* / buffer_load_*)
  rounds       `s_waitcnt vmcnt(..)` instructions with at least one load issued since the previous one = dependent levels, upper bound
  loads/round  smallest and median number of loads a round covers (1 = a fully serialised load)
  in-loop      rounds that sit inside a loop (label targeted by a backward branch): they repeat per trip
  polls        rounds whose loads are `sc1` / atomic loads in a loop with `s_sleep` (spin waits on another workgroup: not data loads)
"""
import os
import re
import statistics
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_lint as L  # noqa: E402


def analyse(fn):
    ins = fn.ins
    # loops: ranges [label index, branch index] of backward branches
    loops = []
    for i, (_, mn, ops, _, _) in enumerate(ins):
        if mn.startswith(("s_cbranch", "s_branch")) and ops and ops[0] in fn.labels and fn.labels[ops[0]] <= i:
            loops.append((fn.labels[ops[0]], i))
    def in_loop(i):
        return any(a <= i <= b for a, b in loops)
    def loop_sleeps(i):
        return any(a <= i <= b and any(ins[j][1] == "s_sleep" for j in range(a, b + 1)) for a, b in loops)
    loads = rounds = inl = polls = 0
    per = []
    since = 0
    since_poll = False
    for i, (_, mn, ops, raw, tail) in enumerate(ins):
        if re.match(r"^(global|flat|buffer)_load", mn) or (mn.startswith("global_atomic") and "glc" in raw or mn.startswith("global_atomic") and "sc0" in raw):
            loads += 1
            since += 1
            if ("sc1" in raw or "atomic" in mn) and loop_sleeps(i):
                since_poll = True
        elif mn == "s_waitcnt" and "vmcnt" in raw:
            if since:
                rounds += 1
                per.append(since)
                if in_loop(i):
                    inl += 1
                if since_poll:
                    polls += 1
            since = 0
            since_poll = False
    return dict(loads=loads, rounds=rounds, lo=min(per) if per else 0, med=statistics.median(per) if per else 0, inl=inl, polls=polls)


def main():
    args = sys.argv[1:]
    keep = None
    if args[:1] == ["--root"]:
        L.ROOT = os.path.abspath(args[1])
        args = args[2:]
    if args[:1] == ["--keep"]:
        keep = args[1]
        args = args[2:]
    tmp = None
    if keep is None:
        tmp = tempfile.TemporaryDirectory()
        keep = tmp.name
    os.makedirs(keep, exist_ok=True)
    L.compile_all(keep)
    rows = []
    for f in L.SRC:
        for fn in L.parse(os.path.join(keep, f + ".s")):
            if not any(i[1] == "s_endpgm" for i in fn.ins):
                continue
            name = fn.name
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            name = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", name)).split("(")[0]
            if args and not any(a in name for a in args):
                continue
            rows.append((f, name, analyse(fn)))
    print("%-13s %-34s %6s %6s %11s %8s %6s" % ("source", "kernel", "loads", "rounds", "loads/round", "in-loop", "polls"))
    for f, name, d in sorted(rows, key=lambda r: -r[2]["rounds"]):
        print("%-13s %-34s %6d %6d %5d / %-5g %8d %6d" % (f, name[:34], d["loads"], d["rounds"], d["lo"], d["med"], d["inl"], d["polls"]))


if __name__ == "__main__":
    main()
