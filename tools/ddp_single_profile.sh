R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in plain ddp; do
  rm -rf /tmp/dp_$tag
  EXTRA=""; [ $tag = ddp ] && EXTRA="--ddp-single"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_$tag -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fwd-sim --no-other-dtype --no-h2d $EXTRA > /tmp/dp_$tag.log 2>&1 || { tail -5 /tmp/dp_$tag.log; exit 1; }
  tail -1 /tmp/dp_$tag.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d.get('ddp_single', {}).get('ms_per_step'))"
done
python3 - <<'PY'
import csv, glob, re
def load(t):
    f = glob.glob(f"/tmp/dp_{t}/**/*kernel_stats.csv", recursive=True)[0]
    return {re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:56]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("plain"), load("ddp")
keys = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]))
print(f"{'kernel':56s} {'plain calls':>11s} {'ms':>8s} {'ddp calls':>10s} {'ms':>8s} {'diff ms':>8s}")
for k in keys[:18]:
    ca, ma = a.get(k, (0, 0)); cb, mb = b.get(k, (0, 0))
    print(f"{k:56s} {ca:11d} {ma:8.3f} {cb:10d} {mb:8.3f} {mb - ma:8.3f}")
print("TOTAL", sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
PY
