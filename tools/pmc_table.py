#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass.  usage: python tools/pmc_table.py <dir> <COUNTER> [<COUNTER> ...]"""
import collections, csv, glob, re, sys
d, names = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:44]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1].get(names[0], [0])))
print(f"{'kernel':44s} launches " + " ".join(f"{n:>24s}" for n in names))
for k, v in rows[:14]:
    n = max(len(x) for x in v.values())
    print(f"{k:44s} {n:8d} " + " ".join(f"{sum(v.get(c, [0])) / max(len(v.get(c, [0])), 1):24.0f}" for c in names))
