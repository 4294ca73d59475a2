#!/bin/bash
# The -m gpu suite against tests/emu, the lockstep CPU emulation of the device library (test infrastructure; tests/emu/README.md) -- what a
# round without GPU access can run.  NOT a GPU run: its logs go under profiles/ with "emu" in their names and are never cited as GPU evidence.
#   tools/emu_suite.sh [tag]            whole suite, 6 workers (~55 min on 8 cores)          -> /tmp/emu_<tag>_suite.log
#   EMU_ASAN=1 tools/emu_suite.sh ...   the same under AddressSanitizer (seed ensembles and exhaustive LM sweeps left out, ~60 min)
#   EMU_SCHED=1 tools/emu_suite.sh ...  descending wave / lane order (a missing barrier shows as a difference)
set -u
TAG=${1:-run}
cd "$(dirname "$0")/.."
EXTRA=()
if [ "${EMU_ASAN:-0}" = "1" ]; then
  export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
  export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=0
  EXTRA=(--deselect tests/test_gpu_rolling_ensemble.py -k "not exhaustive")
fi
SOS_EMU=1 python -X faulthandler -m pytest tests -m gpu -q -rA -p no:cacheprovider -n ${EMU_WORKERS:-6} --durations=10 "${EXTRA[@]}" > /tmp/emu_${TAG}_suite.log 2>&1
echo "rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" /tmp/emu_${TAG}_suite.log | tail -20
grep -c AddressSanitizer /tmp/emu_${TAG}_suite.log
