#!/bin/bash
# same-box A/B of environment settings on the train step: usage tools/ab_env_train.sh "VAR=a VAR=b ..." [rounds]
# (each setting is one bench.py process: headline leg only, 12 steps after 3 of warm-up; alternating rounds)
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${2:-2}
for r in $(seq 1 $ROUNDS); do
  for kv in $1; do
    echo -n "== $kv round $r: "
    env $kv timeout -k 10 300 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-fwd-sim --no-other-dtype --no-h2d 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'], 'ms/step; wgrad', d['roofline']['avg_us'], 'us')"
  done
done
