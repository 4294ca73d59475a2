#!/bin/bash
# usage: tools/_ab.sh ENVVAR  -- alternates runs with / without ENVVAR=1
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export $1=1; else unset $1; fi
    r=$(timeout 120 python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$1=$v $r"
  done
done
