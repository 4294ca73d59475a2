"""In-tree builds of the native libraries (explicit hipcc / g++ commands, no JIT cache).

  csrc/libsos_slam_hip.so   HIP kernels + C-ABI of include/sos_slam.h          (hipcc, gfx950 only)
  csrc/libsos_host.so       C++ host facade (EnergyFunctional / optimize / CoarseTracker)  (g++)

-ffp-contract=off pins the fp32 arithmetic convention shared with the CPU oracle.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")

HIP_SOURCES = ["sos_ctx.hip", "sos_ba.hip", "sos_tracker.hip", "sos_comm.hip", "sos_immature.hip", "sos_pixsel.hip", "sos_undistort.hip"]
HIP_LIB = os.path.join(CSRC, "libsos_slam_hip.so")
HOST_SOURCES = ["host/sos_host.cpp", "host/sos_imu.cpp", "host/sos_sequence.cpp"]
HOST_LIB = os.path.join(CSRC, "libsos_host.so")

HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
               "-fno-slp-vectorize", "-Wno-unused-value",
               # leading scalar kernel arguments are delivered in SGPRs at wave start instead of being fetched from the
               # kernel-argument segment (the kernels keep the fetching preamble for firmware without the feature)
               "-mllvm", "-amdgpu-kernarg-preload-count=16"]
CXX_FLAGS = ["-O3", "-mavx2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-pthread"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = list(sources) + [os.path.join(INCLUDE, "sos_slam.h"), os.path.join(INCLUDE, "sos_slam_host.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hostdir = os.path.join(CSRC, "host")
    if os.path.isdir(hostdir):
        deps += [os.path.join(hostdir, f) for f in os.listdir(hostdir) if f.endswith((".h", ".hpp"))]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force or _stale(HIP_LIB, srcs):
        extra = os.environ.get("SOS_HIPCC_EXTRA", "").split()  # experiment knob, e.g. -DSOS_LIN_WAVES=5
        cmd = [hipcc_path()] + HIPCC_FLAGS + extra + ["-o", HIP_LIB] + srcs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


def build_host(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not srcs:
        return None
    if force or _stale(HOST_LIB, srcs):
        cmd = ["g++"] + CXX_FLAGS + ["-o", HOST_LIB] + srcs + ["-L" + CSRC, "-lsos_slam_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
