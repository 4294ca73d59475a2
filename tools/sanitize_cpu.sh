#!/bin/bash
# The CPU test-suite with the oracle (C) and the host facade (C++) built under AddressSanitizer, then under UndefinedBehaviorSanitizer.
# No GPU needed: what runs is everything the `-m "not gpu"` tests reach -- the oracle, the IMU assembly / solves, the candidate
# selection, the dense solvers, the ABI zero-argument calls.  The in-tree libraries are swapped for the instrumented ones and restored.
#   tools/sanitize_cpu.sh          (round 3: 115, round 4: 135, round 5: 152 passed under both; the emulator tests have their own sanitizer runs: tools/emu_suite.sh, tools/emu_host_asan.sh)
set -u
cd "$(dirname "$0")/.."
T=$(mktemp -d)
SRC_O="oracle/orc_backend.c oracle/orc_host.c oracle/orc_tracker.c oracle/orc_immature.c oracle/orc_pixsel.c oracle/orc_undistort.c oracle/orc_imu.c"
SRC_H="sos_slam_amd/csrc/host/sos_host.cpp sos_slam_amd/csrc/host/sos_imu.cpp sos_slam_amd/csrc/host/sos_sequence.cpp"
cp oracle/liboracle.so $T/liboracle.orig; cp sos_slam_amd/csrc/libsos_host.so $T/libsos_host.orig
restore() { cp $T/liboracle.orig oracle/liboracle.so; cp $T/libsos_host.orig sos_slam_amd/csrc/libsos_host.so; rm -rf $T; }
trap restore EXIT
for SAN in address undefined; do
  EXTRA=""; [ $SAN = undefined ] && EXTRA="-fno-sanitize-recover=undefined"
  cc -O1 -g -std=gnu11 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=$SAN $EXTRA -fno-omit-frame-pointer -pthread -shared -o $T/o.so $SRC_O -lm -lpthread || exit 1
  g++ -O1 -g -mavx2 -std=c++17 -ffp-contract=off -fPIC -shared -fsanitize=$SAN $EXTRA -fno-omit-frame-pointer -pthread -o $T/h.so $SRC_H \
      -L$PWD/sos_slam_amd/csrc -lsos_slam_hip -Wl,-rpath,$PWD/sos_slam_amd/csrc || exit 1
  cp $T/o.so oracle/liboracle.so; cp $T/h.so sos_slam_amd/csrc/libsos_host.so; touch oracle/liboracle.so
  LIB=$(gcc -print-file-name=$([ $SAN = address ] && echo libasan.so || echo libubsan.so))
  echo "== $SAN"
  LD_PRELOAD=$LIB ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 timeout 3000 python -m pytest tests -x -q -m "not gpu" \
      -p no:cacheprovider --deselect tests/test_bench_cli.py --deselect tests/test_distributed_gloo.py --deselect tests/test_emu_selfcheck.py --deselect tests/test_emulated_smoke.py 2>&1 | tail -4
done
