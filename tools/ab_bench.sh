#!/bin/bash
# Same-box A/B of two library builds (boxes differ by +-5 %): alternates base / new, 2 rounds each.
# usage: tools/ab_bench.sh <base.so> [bench args...]
BASE=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for tag in base new; do
    if [ $tag = base ]; then export SIGNAL_HIP_LIB=$R/$BASE; else unset SIGNAL_HIP_LIB; fi
    v=$(timeout -k 10 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$tag round $round: $v ms/step"
  done
done
