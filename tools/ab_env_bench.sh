#!/bin/bash
# same-box A/B of the whole bench for several values of one env var (alternated, two passes).
# usage: tools/ab_env_bench.sh <VAR> "<v1> <v2> ..." [bench args]
VAR=$1; VALS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
for pass in 1 2; do
  for v in $VALS; do
    env $VAR=$v python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fwd-sim --no-other-dtype "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('$VAR=$v pass $pass: %.3f ms/step  %.1f triplets/s  roofline %s avg %.1f us frac %.3f' % (d['ms_per_step'], d['value'], r.get('kernel'), r.get('avg_us',0), r.get('frac',0)))
"
  done
done
