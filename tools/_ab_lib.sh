#!/bin/bash
# usage: tools/_ab_lib.sh  -- alternates bench runs between libsos_slam_hip.so (new) and libsos_slam_hip_old.so
cd sos_slam_amd/csrc; cp libsos_slam_hip.so new.so; cd ../..
for i in 1 2 3 4; do
  for v in new old; do
    cp sos_slam_amd/csrc/$( [ $v = new ] && echo new.so || echo libsos_slam_hip_old.so ) sos_slam_amd/csrc/libsos_slam_hip.so
    r=$(timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_us']['linearize_fused_us'], d['kernels_us']['linearize_us'], round(d['ms_per_step']*1e3,1))")
    echo "$v $r"
  done
done
cp sos_slam_amd/csrc/new.so sos_slam_amd/csrc/libsos_slam_hip.so
