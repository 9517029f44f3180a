#!/bin/bash
# per-kernel in-model comparison of two library builds on one box. usage: tools/ab_profile.sh <base.so> [bench args]
BASE=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in base new; do
  if [ $tag = base ]; then export SIGNAL_HIP_LIB=$R/$BASE; else unset SIGNAL_HIP_LIB; fi
  rm -rf /tmp/abp_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$tag -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > /tmp/abp_$tag.log 2>&1 || { tail -5 /tmp/abp_$tag.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, re
def load(tag):
    f = glob.glob(f"/tmp/abp_{tag}/**/*kernel_stats.csv", recursive=True)[0]
    return {re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:60]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
b, n = load("base"), load("new")
keys = sorted(set(b) | set(n), key=lambda k: -(b.get(k, (0, 0))[1] + n.get(k, (0, 0))[1]))
print(f"{'kernel':60s} {'calls':>6s} {'base ms':>9s} {'new ms':>9s} {'delta':>8s}")
tb = tn = 0
for k in keys[:28]:
    cb, mb = b.get(k, (0, 0)); cn, mn = n.get(k, (0, 0))
    print(f"{k:60s} {cb:6d} {mb:9.3f} {mn:9.3f} {mn-mb:+8.3f}")
for k in keys:
    tb += b.get(k, (0, 0))[1]; tn += n.get(k, (0, 0))[1]
print(f"{'TOTAL (12 steps)':60s} {'':6s} {tb:9.3f} {tn:9.3f} {tn-tb:+8.3f}")
PY
