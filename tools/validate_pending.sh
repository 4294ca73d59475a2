#!/bin/bash
# One gpurun call that settles what was built while no GPU was reachable (rounds 3-4).  From the repository root:
#   gpurun --timeout 3000 -- 'tools/validate_pending.sh r05a'
# 1. the whole GPU suite on the defaults, one line per test          -> gpurun_out/<tag>/gputests.log      (copy to profiles/)
# 2. bench.py as the driver runs it: headline + keyframe + visual_inertial + `variants` (abs / abs-cooperative / in-kernel signal /
#    eager mirrors / resident, each in its own process)              -> gpurun_out/<tag>/bench.json
# 3. kernel traces of the default chain and of the cooperative absolute-coordinate chain -> *_kernel_stats.csv
# 4. the suites that exercise the opt-in paths: abs (incl. cooperative), the C++ frame-rate loop under the rolling oracle tests, IMU overlap
set -u
TAG=${1:-pending}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/gputests.log 2>&1
echo "suite rc=$?"
grep -v "$F" $OUT/gputests.log | grep -E "passed|failed|^FAILED|^ERROR|Fatal" | head -20
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json"))
    print("us/iter %.1f" % (d["ms_per_step"] * 1e3), "frac", d["roofline"]["frac"], "frac_of_measured_stream", d["roofline"]["frac_of_measured_stream"],
          "kf", d.get("optimize_ms"), d.get("keyframe_ms"), (d.get("keyframe") or {}).get("keyframe_with_activation_ms"))
    print("host phases", d.get("host_phases_us"))
    print("vio", {k: v for k, v in (d.get("visual_inertial") or {}).items() if "ms" in k or "us" in k}, "| overlap", (d.get("visual_inertial_overlap") or {}).get("ms_per_iteration"))
    for k, v in (d.get("variants") or {}).items():
        print("variant %-28s" % k, v.get("us_per_iteration", v.get("error")), v.get("resInA"), v.get("last_step_l2"), v.get("kernels_us"))
except Exception as e:
    print("bench ERR", e)
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_def -o b -- python $OLDPWD/bench.py --no-cpu-baseline --no-sides --steps 30 --inner 50 > /dev/null 2>> $OUT/prof.err)
python tools/rocpd_summary.py kernels $OUT/prof_def/b_results.db $OUT/bench_default_kernel_stats.csv 2>> $OUT/prof.err; rm -rf $OUT/prof_def
(cd /tmp && SOS_ABS_SC=1 SOS_ABS_COOP=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_coop -o b -- python $OLDPWD/bench.py --no-cpu-baseline --no-sides --steps 30 --inner 50 > /dev/null 2>> $OUT/prof.err)
python tools/rocpd_summary.py kernels $OUT/prof_coop/b_results.db $OUT/bench_abs_coop_kernel_stats.csv 2>> $OUT/prof.err; rm -rf $OUT/prof_coop
head -12 $OUT/bench_default_kernel_stats.csv | cut -c1-50,150-260; head -8 $OUT/bench_abs_coop_kernel_stats.csv | cut -c1-50,150-260
for V in "SOS_ABS_SC=1" "SOS_ABS_SC=1 SOS_ABS_COOP=1"; do
  env $V timeout 1200 python -X faulthandler -m pytest tests/test_gpu_backend.py tests/test_gpu_baseline_sizes.py tests/test_gpu_edge_windows.py tests/test_golden_t6.py \
    tests/test_gpu_keyframe_pipeline.py tests/test_gpu_rolling_window.py -q -m gpu -k "not T4" -p no:cacheprovider > "$OUT/gputests_$(echo $V | tr -dc 'A-Z_')".log 2>&1
  grep -v "$F" "$OUT/gputests_$(echo $V | tr -dc 'A-Z_')".log | grep -E "passed|failed|^FAILED" | head -8
done
SOS_ROLLING_CPP=1 timeout 1200 python -X faulthandler -m pytest tests/test_gpu_rolling_window.py tests/test_gpu_rolling_vio.py -q -m gpu -p no:cacheprovider > $OUT/gputests_rolling_cpp.log 2>&1
grep -v "$F" $OUT/gputests_rolling_cpp.log | grep -E "passed|failed|^FAILED" | head -8
SOS_IMU_OVERLAP=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_imu_hook.py tests/test_gpu_rolling_vio.py -q -m gpu -p no:cacheprovider > $OUT/gputests_imu_overlap.log 2>&1
grep -v "$F" $OUT/gputests_imu_overlap.log | grep -E "passed|failed|^FAILED" | head -8
timeout 300 python tools/tracker_bench.py W12 > $OUT/tracker_W12.json 2>> $OUT/bench.err; head -c 600 $OUT/tracker_W12.json; echo
