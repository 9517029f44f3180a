#!/bin/bash
# usage: tools/ab_multi.sh "<lib1.so> <lib2.so> ..." [bench args]   ('-' = the shipped library)
LIBS=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
  for lib in $LIBS; do
    if [ $lib = - ]; then unset SIGNAL_HIP_LIB; else export SIGNAL_HIP_LIB=$R/$lib; fi
    v=$(timeout -k 10 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$lib round $round: $v ms/step"
  done
done
