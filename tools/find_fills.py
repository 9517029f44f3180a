#!/usr/bin/env python3
"""Diagnostic (GPU box): who calls torch.zeros / zero_ / fill_ inside a steady-state forward step?"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
cfg, model = bench.build_model(dev, sys.argv[1] if len(sys.argv) > 1 else "fwd_sim")
img, vid, cam = bench.synthetic(cfg, 64, dev, 1234)
def step():
    with torch.no_grad():
        return model(img, cam_label=cam, training=False)
for _ in range(3): step()
torch.cuda.synchronize()
sites = collections.Counter()
orig = torch.zeros
def zeros(*a, **k):
    fr = traceback.extract_stack(limit=4)[:-1]
    sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr))] += 1
    return orig(*a, **k)
torch.zeros = zeros
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
torch.zeros = orig
print("torch.zeros call sites in one step:")
for k, v in sites.most_common(20): print(f"  {v:4d}  {k}")
ev = [e for e in prof.events() if "fill" in e.name.lower() or "zero" in e.name.lower()]
c = collections.Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "signal_amd" in s or "bench" in s][:3]
    c[(e.name, " <- ".join(st))] += 1
print("fill/zero ops in one step:")
for (n, s), v in c.most_common(25): print(f"  {v:4d} {n} | {s}")
