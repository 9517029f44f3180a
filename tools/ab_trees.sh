#!/bin/bash
# same-box A/B of two source trees on the train step (headline leg only): usage tools/ab_trees.sh <treeA> <treeB> [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq 1 $ROUNDS); do
  for t in $A $B; do
    echo -n "== $t round $r: "
    EXTRA="--no-h2d"; grep -q -- "--no-h2d" $R/$t/bench.py || EXTRA=""      # (older trees: the PCIe-inclusive leg was opt-in)
    (cd $R/$t && timeout -k 10 300 python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-fwd-sim --no-other-dtype $EXTRA "$@" 2>/dev/null) | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms/step; roofline kernel', d['roofline']['avg_us'], 'us')"
  done
done
