#!/bin/bash
# The C++ facade (csrc/host/*.cpp) built with gcc AddressSanitizer against the EMULATED device library (tests/emu), then two rolling chains
# through it -- a visual one driven by the C++ frame-rate loop and a stereo-inertial one driven by the Python loop: keyframes join, points are
# activated / marginalised / dropped, keyframes leave, the system is torn down.  Every constructor / destructor path of the facade's object
# graph runs under ASan (use-after-free, overflows); no GPU needed.   tools/emu_host_asan.sh  ->  prints the chains' summary, ASan reports if any
set -eu
cd "$(dirname "$0")/.."
python tests/emu/build_emu.py > /dev/null
D=tests/emu/_build_hostasan
mkdir -p $D && cp tests/emu/_build/libsos_slam_hip.so $D/
g++ -O1 -g -mavx2 -std=c++17 -ffp-contract=off -fPIC -shared -pthread -fsanitize=address -fno-omit-frame-pointer -o $D/libsos_host.so \
  sos_slam_amd/csrc/host/sos_host.cpp sos_slam_amd/csrc/host/sos_imu.cpp sos_slam_amd/csrc/host/sos_sequence.cpp -L$D -lsos_slam_hip -Wl,-rpath,'$ORIGIN'
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from sos_slam_amd import build as b
b.HIP_LIB = os.path.abspath("tests/emu/_build_hostasan/libsos_slam_hip.so")
b.HOST_LIB = os.path.abspath("tests/emu/_build_hostasan/libsos_host.so")
b.build_all = lambda *a, **k: (b.HIP_LIB, b.HOST_LIB)
from tests import rolling
for kw in (dict(n_frames=16), dict(vio=True, stereo=True, n_frames=12)):
    sc = rolling.Scenario(**kw)
    dev = rolling.device_chain(sc)
    dev.bootstrap()
    left = 0
    while dev.next_frame < sc.n_frames:
        left += len(dev.step().marginalized)
    print(kw, "keyframes left:", left, flush=True)
    dev.close()
print("no AddressSanitizer report above = clean")
PY
