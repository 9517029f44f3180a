#!/bin/bash
# in-model time of kernels matching a pattern, for several values of one env var.
# usage: tools/ab_kernel_env.sh <pattern> <VAR> "<v1> <v2> ..." [bench args]
PAT=$1; VAR=$2; VALS=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  rm -rf /tmp/abk
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > /tmp/abk.log 2>&1 || { tail -5 /tmp/abk.log; exit 1; }
  python3 - "$PAT" "$VAR=$v" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/abk/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[1] in r["Name"]:
        print(f"{sys.argv[2]:28s} {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
done
