#!/bin/bash
# Collects the committed profile set of a round on the GPU box (run from the repository root through gpurun):
#   tools/collect_profiles.sh r02
# bench JSON (W12 host-solve loop = the default, W16, W12 device-resident loop), rocprofv3 kernel traces of the W12 runs
# (summarised with tools/rocpd_summary.py), PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel-trace only) for
# W12 and W16.  Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --window W16 --no-cpu-baseline > $OUT/bench_W16.json 2>> $OUT/bench.err
python bench.py --resident --no-cpu-baseline > $OUT/bench_resident.json 2>> $OUT/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o b -- python $OLDPWD/bench.py --no-cpu-baseline --steps 30 > /dev/null 2>> $OUT/prof.err)
python tools/rocpd_summary.py kernels $OUT/prof_bench/b_results.db $OUT/bench_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_res -o b -- python $OLDPWD/bench.py --resident --no-cpu-baseline --steps 30 > /dev/null 2>> $OUT/prof.err)
python tools/rocpd_summary.py kernels $OUT/prof_res/b_results.db $OUT/bench_resident_kernel_stats.csv
for W in W12 W16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${W}_$C -o p -- python $OLDPWD/tools/pmc_probe.py $W > $OUT/pmc_${W}_$C.log 2>> $OUT/prof.err)
    python tools/rocpd_summary.py counters $OUT/pmc_${W}_$C/p_results.db $OUT/pmc_${W}_$C.csv
  done
done
for W in W12 W16; do  # what bench.py's roofline.traffic reads: regenerated from THIS collection
  python tools/pmc_json.py $W $OUT/pmc_${W}_FETCH_SIZE.csv $OUT/pmc_${W}_WRITE_SIZE.csv profiles/pmc_$W.json avg > /dev/null
  cp profiles/pmc_$W.json $OUT/pmc_$W.json
done
python bench.py --imu --no-cpu-baseline > $OUT/bench_imu.json 2>> $OUT/bench.err
python tools/tracker_bench.py W12 > $OUT/tracker_W12.json 2>> $OUT/bench.err
rm -rf $OUT/prof_bench $OUT/prof_res $OUT/pmc_*/
ls -la $OUT
