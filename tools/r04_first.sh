#!/bin/bash
# round-4 first call: whole GPU suite on the defaults (per-test lines), then the default bench lines
set -u
TAG=${1:-r04a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1700 python -X faulthandler -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/gputests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|Fatal|XPASS|XFAIL" $OUT/gputests.log | head -40
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 3000 $OUT/bench_default.json
timeout 300 python bench.py --imu --no-cpu-baseline --steps 10 --inner 60 > $OUT/bench_imu.json 2>> $OUT/bench.err; echo "imu rc=$?"
tail -c 1500 $OUT/bench_imu.json
tail -5 $OUT/bench.err
