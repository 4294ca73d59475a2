#!/bin/bash
# One call that settles what was built while the GPU was not reachable (run from the repository root through gpurun):
#   tools/validate_pending.sh <tag>
# 1. the whole GPU suite on the defaults (incl. the loop-aligner parity at 752x480 / 1232x368 added blind)
# 2. the backend / optimize / distributed / rolling tests and the first-solve probe with the absolute-coordinate Schur path
# 3. bench A/B/C: default | SOS_ABS_SC=1 | SOS_ABS_SC=1 SOS_ABS_SIGNAL_IN_KERNEL=1, W12 and W16, and a kernel trace of the abs path
# Everything lands under gpurun_out/<tag>/.
set -u
TAG=${1:-pending}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -rxX > $OUT/gputests_default.log 2>&1
grep -v "$F" $OUT/gputests_default.log | grep -E "passed|failed|FAILED|Fatal|XPASS|XFAIL" | head -20
grep -h "idepth_hessian W" $OUT/gputests_default.log | sort -u | head -6
# the numbers that matter most first (a call cut short still leaves them): default bench lines, the visual-inertial loop, a kernel trace
timeout 300 python bench.py --window W12 > $OUT/bench_W12_full.json 2>> $OUT/bench.err
timeout 300 python bench.py --imu --no-cpu-baseline --steps 10 --inner 60 > $OUT/bench_imu_T1.json 2>> $OUT/bench.err
SOS_IMU_CACHE=0 timeout 300 python bench.py --imu --no-cpu-baseline --steps 10 --inner 60 > $OUT/bench_imu_literal.json 2>> $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_def -o b -- python $OLDPWD/bench.py --no-cpu-baseline --steps 30 --inner 50 > /dev/null 2>> $OUT/prof.err)
python tools/rocpd_summary.py kernels $OUT/prof_def/b_results.db $OUT/bench_default_kernel_stats.csv 2>> $OUT/prof.err; rm -rf $OUT/prof_def
python - <<PY
import json
for f in ("bench_W12_full", "bench_imu_T1", "bench_imu_literal"):
    try:
        d = json.load(open("$OUT/" + f + ".json"))
        print(f, "ms/step %.4f" % d["ms_per_step"], "loop", d["config"].get("gn_loop", "")[:40], "frac", (d.get("roofline") or {}).get("frac"), "kf", d.get("keyframe_ms"),
              "vio", {k: v for k, v in (d.get("visual_inertial") or {}).items() if "ms" in k or "us" in k or "solves" in k})
    except Exception as e:
        print(f, "ERR", e)
PY
SOS_ABS_SC=1 timeout 1500 python -X faulthandler -m pytest tests/test_gpu_backend.py tests/test_gpu_optimize.py tests/test_gpu_baseline_sizes.py \
  tests/test_gpu_distributed.py tests/test_gpu_edge_windows.py tests/test_gpu_variants.py tests/test_gpu_bench_rehearsal.py tests/test_golden.py \
  tests/test_golden_t6.py tests/test_gpu_imu_hook.py tests/test_gpu_rolling_window.py tests/test_gpu_rolling_vio.py tests/test_gpu_rolling_ensemble.py \
  tests/test_gpu_keyframe_pipeline.py tests/test_gpu_marginalize.py -q > $OUT/gputests_abs.log 2>&1
grep -v "$F" $OUT/gputests_abs.log | grep -E "passed|failed|FAILED|Fatal" | head -12
timeout 300 python tools/first_solve_probe.py 2>&1 | grep variant | sed 's/^/default: /'
SOS_ABS_SC=1 timeout 300 python tools/first_solve_probe.py 2>&1 | grep variant | sed 's/^/abs:     /'
for W in W12 W16; do
  timeout 300 python bench.py --window $W --no-cpu-baseline > $OUT/bench_${W}_default.json 2>> $OUT/bench.err
  SOS_ABS_SC=1 timeout 300 python bench.py --window $W --no-cpu-baseline > $OUT/bench_${W}_abs.json 2>> $OUT/bench.err
  SOS_ABS_SC=1 SOS_ABS_SIGNAL_IN_KERNEL=1 timeout 300 python bench.py --window $W --no-cpu-baseline > $OUT/bench_${W}_abs_sig.json 2>> $OUT/bench.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), "us/iter %.1f" % (d["ms_per_step"] * 1e3), "resInA", d["config"]["resInA_last_iteration"], "step", d["last_step_l2"],
              {k: d["kernels_us"].get(k) for k in ("sc_gram_prep_us", "reduce_us", "stitch_us", "sc_gram_abs_us", "abs_reduce_stitch1_us", "abs_stitch2_us")},
              "kf", d.get("optimize_ms"), d.get("keyframe_ms"), "vio", (d.get("visual_inertial") or {}).get("ms_per_iteration"))
    except Exception as e:
        print(f, "ERR", e)
PY
(cd /tmp && SOS_ABS_SC=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_abs -o b -- python $OLDPWD/bench.py --no-cpu-baseline --steps 30 --inner 50 > /dev/null 2>> $OUT/prof.err)
python tools/rocpd_summary.py kernels $OUT/prof_abs/b_results.db $OUT/bench_abs_kernel_stats.csv
rm -rf $OUT/prof_abs
head -12 $OUT/bench_abs_kernel_stats.csv | cut -c1-50,150-260
# the tracker after the scratch removal (r03n: trackNewestCoarse 0.235 ms, 9.1 us per evaluation)
timeout 300 python tools/tracker_bench.py W12 > $OUT/tracker_W12.json 2>> $OUT/bench.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/tracker_W12.json"))
    print("tracker", {k: v for k, v in d.items() if "ms" in k or "us" in k})
except Exception as e:
    print("tracker ERR", e)
PY
# the visual-inertial iteration (r03n: 0.60 ms, host KKT solve 0.50 ms) after the leaner assembly / build, and with helper threads
for T in 1 2 4; do
  SOS_SOLVE_THREADS=$T timeout 300 python bench.py --imu --no-cpu-baseline --steps 10 --inner 60 > $OUT/bench_imu_T$T.json 2>> $OUT/bench.err
done
# the kept-factor solve of the trapped-scale case (default) against the literal form, inside the loop and on its own
SOS_IMU_CACHE=0 timeout 300 python bench.py --imu --no-cpu-baseline --steps 10 --inner 60 > $OUT/bench_imu_literal.json 2>> $OUT/bench.err
timeout 120 python tools/imu_solve_bench.py W12 200 2>&1 | tail -1
timeout 300 python bench.py --side imu --window W12 2>> $OUT/bench.err | tail -1 > $OUT/bench_imu_side.json; head -c 1200 $OUT/bench_imu_side.json; echo
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_imu_*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), "ms/iter %.3f" % d["ms_per_step"], "loop", d["config"].get("gn_loop"), "resInA", d["config"]["resInA_last_iteration"])
    except Exception as e:
        print(f, "ERR", e)
PY
# the rolling-window parity tests driven by the C++ frame-rate loop (sosf_sequence) instead of the Python loop
SOS_ROLLING_CPP=1 timeout 1500 python -X faulthandler -m pytest tests/test_gpu_rolling_window.py tests/test_gpu_rolling_vio.py -q > $OUT/gputests_rolling_cpp.log 2>&1
grep -v "$F" $OUT/gputests_rolling_cpp.log | grep -E "passed|failed|FAILED|Fatal|Error" | head -12
