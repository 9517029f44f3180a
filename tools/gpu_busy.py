#!/usr/bin/env python3
"""From a rocprofv3 kernel trace CSV: per-step GPU busy time (sum of kernel durations), idle gaps and their count.
usage: gpu_busy.py <kernel_trace.csv> <steps_in_trace>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady state: drop the first third of the launches (warm-up, allocation)
ev = ev[len(ev) // 3:]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
pos = [g for g in gaps if g > 0]
print(f"launches {len(ev)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({100*busy/span:.1f}%)  idle {sum(pos)/1e6:.3f} ms in {len(pos)} gaps "
      f"(median {sorted(pos)[len(pos)//2]/1e3:.2f} us, >20us: {sum(1 for g in pos if g > 20000)} totalling {sum(g for g in pos if g > 20000)/1e6:.3f} ms)")
big = sorted(((gaps[i], ev[i][2][:50], ev[i + 1][2][:50]) for i in range(len(gaps))), reverse=True)[:12]
for g, a, b in big: print(f"  {g/1e3:8.1f} us  after {a}  before {b}")
