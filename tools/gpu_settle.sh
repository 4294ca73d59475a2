#!/bin/bash
# First GPU call of a round: the whole GPU suite on the defaults (one line per test), the driver's bench command, the default chain's
# kernel trace, the two PMC passes of the roofline kernel (profiles/pmc_<window>.json regenerated with the commit and the hash of the
# kernel sources it was measured on -- bench.py reports the counter value only when that hash is the tree's), and, when tools/bisect_kit.sh
# has left trees under bisect/, suite + bench of each of them in the same lease (attribution of a regression to a commit range).
#   gpurun --timeout 3000 -- "SOS_SOURCE_COMMIT=$(git rev-parse --short HEAD) tools/gpu_settle.sh r06a"      (the box has no .git)
#   SETTLE_SKIP="bisect pmc" ...   leaves phases out
set -u
TAG=${1:-settle}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl'
# the first minute: the hand-written asm of k_gn_solve's pivot chain against the builtin form, and the solve itself
timeout 300 python -X faulthandler -m pytest tests/test_gpu_gn_solve.py -m gpu -q -rA -p no:cacheprovider > $OUT/gn_solve_first.log 2>&1; echo "k_gn_solve micro-test + solve rc=$?"
grep -v "$F" $OUT/gn_solve_first.log | grep -E "passed|failed|^FAILED|^ERROR|Fatal|^PASSED" | head -12
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

case " ${SETTLE_SKIP:-} " in *" pmc "*) ;; *)
for W in W12 W16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${W}_$C -o p -- python $OLDPWD/tools/pmc_probe.py $W > $OUT/pmc_${W}_$C.log 2>> $OUT/prof.err)
    python tools/rocpd_summary.py counters $OUT/pmc_${W}_$C/p_results.db $OUT/pmc_${W}_$C.csv 2>> $OUT/prof.err
  done
  python tools/pmc_json.py $W $OUT/pmc_${W}_FETCH_SIZE.csv $OUT/pmc_${W}_WRITE_SIZE.csv $OUT/pmc_$W.json avg 2>> $OUT/prof.err
done
rm -rf $OUT/pmc_*/
;; esac
case " ${SETTLE_SKIP:-} " in *" bisect "*) ;; *)
for d in bisect/*/; do
  [ -f $d/bench.py ] || continue
  s=$(basename $d)
  (cd $d && timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/bisect_${s}_gputests.log 2>&1; echo "bisect $s suite rc=$?"
   grep -v "$F" $OUT/bisect_${s}_gputests.log | grep -E "passed|failed|^FAILED|^ERROR|Fatal" | head -12
   timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bisect_${s}_bench.json 2> $OUT/bisect_${s}_bench.err; echo "bisect $s bench rc=$?"
   python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bisect_${s}_bench.json") if l.startswith("{")][-1])
    print("bisect $s: us/iter %.1f" % (d["ms_per_step"] * 1e3), "kernels", d.get("kernels_us"))
except Exception as e:
    print("bisect $s bench ERR", e)
PY
  )
done
;; esac
