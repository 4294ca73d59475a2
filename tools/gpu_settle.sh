#!/bin/bash
# First GPU call of a round: the whole GPU suite on the defaults (one line per test), the driver's bench command, the default chain's
# kernel trace.   gpurun --timeout 2400 -- 'tools/gpu_settle.sh r05a'
set -u
TAG=${1:-settle}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/gputests.log 2>&1
echo "suite rc=$?"
grep -v "$F" $OUT/gputests.log | grep -E "passed|failed|^FAILED|^ERROR|Fatal" | head -30
/usr/bin/time -v timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
grep -E "Elapsed|Maximum resident" $OUT/bench.err
python - <<PY
import json
try:
    lines = [l for l in open("$OUT/bench.json") if l.startswith("{")]
    print("json lines:", len(lines))
    d = json.loads(lines[-1])
    print("us/iter %.1f" % (d["ms_per_step"] * 1e3), "frac", d["roofline"]["frac"], "frac_of_measured_stream", d["roofline"]["frac_of_measured_stream"],
          "kf", d.get("optimize_ms"), d.get("keyframe_ms"), (d.get("keyframe") or {}).get("keyframe_with_activation_ms"))
    print("kernels", d.get("kernels_us"))
    print("host phases", d.get("host_phases_us"))
    print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if not isinstance(v, (dict, list, str))})
    print("tracker", d.get("tracker"))
    print("vio", {k: v for k, v in (d.get("visual_inertial") or {}).items() if "ms" in k or "us" in k or "error" in k})
except Exception as e:
    print("bench ERR", e)
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_def -o b -- python $OLDPWD/bench.py --no-cpu-baseline --no-sides --steps 30 --inner 50 > /dev/null 2>> $OUT/prof.err)
python tools/rocpd_summary.py kernels $OUT/prof_def/b_results.db $OUT/bench_default_kernel_stats.csv 2>> $OUT/prof.err; rm -rf $OUT/prof_def
head -14 $OUT/bench_default_kernel_stats.csv | cut -c1-60,150-260
