"""Compile-time resource report of every kernel (no GPU needed): hipcc -Rpass-analysis=kernel-resource-usage for gfx950 on the
sources of libsos_slam_hip.so, one line per kernel -- VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), static LDS.

    python tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = ["sos_ctx", "sos_ba", "sos_tracker", "sos_comm", "sos_immature", "sos_pixsel", "sos_undistort"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-value", "-mllvm",
         "-amdgpu-kernarg-preload-count=16"]          # sos_slam_amd/build.py


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in SRC:
            p = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", os.path.join(ROOT, "sos_slam_amd", "csrc", f + ".hip"), "-o", os.path.join(tmp, f + ".o"),
                                "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=tmp)
            if p.returncode != 0:
                sys.exit(p.stderr[-2000:])
            cur = None
            for ln in p.stderr.splitlines():
                m = re.search(r"remark: (?:[^:]*:\d+:\d+: )?(.*?) \[-Rpass", ln)
                if not m:
                    continue
                t = m.group(1).strip()
                if t.startswith("Function Name:") or t.startswith("Name:"):
                    name = t.split(":", 1)[1].strip()
                    for filt in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
                        try:
                            name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
                            break
                        except OSError:
                            continue
                    name = re.sub(r"\(anonymous namespace\)::", "", name)
                    name = re.sub(r"^void ", "", name).split("(")[0]
                    cur = dict(name=name, file=f)
                    rows.append(cur)
                elif cur is not None and ":" in t:
                    k, v = t.split(":", 1)
                    cur[k.strip()] = v.strip()
    print("%-14s %-34s %5s %5s %8s %4s %7s" % ("source", "kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
    for r in rows:
        print("%-14s %-34s %5s %5s %8s %4s %7s" % (r["file"], r["name"][:34], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                                     r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
    bad = [r["name"] for r in rows if r.get("ScratchSize [bytes/lane]", "0") != "0"]
    print("# kernels with scratch:", ", ".join(bad) if bad else "none")


if __name__ == "__main__":
    main()
