#!/bin/bash
# per-kernel in-model comparison of two environment settings on one box (rocprofv3 kernel trace of the train step each).
# usage: tools/ab_env_profile.sh "VAR=a" "VAR=b" [kernel name filter]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for kv in "$1" "$2"; do
  i=$((i+1)); rm -rf /tmp/abe_$i
  export $kv
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abe_$i -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fwd-sim --no-other-dtype --no-h2d > /tmp/abe_$i.log 2>&1 || { tail -5 /tmp/abe_$i.log; exit 1; }
done
python3 - "$1" "$2" "${3:-gemm}" <<'PY'
import csv, glob, re, sys
def load(i):
    f = glob.glob(f"/tmp/abe_{i}/**/*kernel_stats.csv", recursive=True)[0]
    return {re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:60]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load(1), load(2)
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1]))
print(f"{'kernel':60s} {sys.argv[1]:>28s} {sys.argv[2]:>28s}")
ta = tb = 0
for k in keys:
    ca, ma = a.get(k, (0, 0)); cb, mb = b.get(k, (0, 0))
    ta += ma; tb += mb
    if sys.argv[3] in k:
        print(f"{k:60s} {ca:6d} x {ma / max(ca, 1) * 1e3:8.2f} us = {ma:7.3f} {cb:6d} x {mb / max(cb, 1) * 1e3:8.2f} us = {mb:7.3f}")
print(f"{'TOTAL kernel time (12 steps), ms':60s} {ta:28.3f} {tb:28.3f}")
PY
