#!/bin/bash
# per-kernel comparison of two source trees on one box (rocprofv3 kernel trace of the train step each)
# usage: tools/ab_trees_profile.sh <treeA> <treeB>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for t in $1 $2; do
  i=$((i+1)); rm -rf /tmp/abt_$i
  EXTRA="--no-h2d"; grep -q -- "--no-h2d" $R/$t/bench.py || EXTRA=""
  (cd $R/$t && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abt_$i -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fwd-sim --no-other-dtype $EXTRA > /tmp/abt_$i.log 2>&1) || { tail -5 /tmp/abt_$i.log; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' /tmp/abt_$i.log | head -1
done
python3 - "$1" "$2" <<'PY'
import csv, glob, re, sys
def load(i):
    f = glob.glob(f"/tmp/abt_{i}/**/*kernel_stats.csv", recursive=True)[0]
    d = {}
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:58]
        c, m = d.get(k, (0, 0.0))
        d[k] = (c + int(r["Calls"]), m + float(r["TotalDurationNs"]) / 1e6)
    return d
a, b = load(1), load(2)
keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1]))
print(f"{'kernel (12 steps)':58s} {'A calls':>7s} {'A ms':>9s} {'B calls':>7s} {'B ms':>9s} {'delta':>8s}")
ta = tb = 0; na = nb = 0
for k in keys:
    ca, ma = a.get(k, (0, 0)); cb, mb = b.get(k, (0, 0))
    ta += ma; tb += mb; na += ca; nb += cb
for k in keys[:34]:
    ca, ma = a.get(k, (0, 0)); cb, mb = b.get(k, (0, 0))
    print(f"{k:58s} {ca:7d} {ma:9.3f} {cb:7d} {mb:9.3f} {mb - ma:+8.3f}")
print(f"{'TOTAL':58s} {na:7d} {ta:9.3f} {nb:7d} {tb:9.3f} {tb - ta:+8.3f}")
PY
