#!/usr/bin/env python3
"""Diagnostic (GPU box): who calls torch.zeros / zero_ / fill_ inside a steady-state forward step?"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
cfg, model = bench.build_model(dev, "bf16")
img, vid, cam = bench.synthetic(cfg, 64, dev, 1234)
if len(sys.argv) > 1 and sys.argv[1] == "train":
    from signal_amd.engine.trainer import TrainStep
    ts = TrainStep(cfg, model, num_classes=171, world_size=1)
    def step():
        return ts.step(img, vid, cam)
else:
    def step():
        with torch.no_grad():
            return model(img, cam_label=cam, training=False)
for _ in range(3): step()
torch.cuda.synchronize()
sites = collections.Counter()
def wrap(name):
    orig = getattr(torch, name)
    def f(*a, **k):
        fr = traceback.extract_stack(limit=5)[:-1]
        sites[name + " " + " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in reversed(fr))] += 1
        return orig(*a, **k)
    setattr(torch, name, f)
    return orig
origs = {n: wrap(n) for n in ("zeros", "zeros_like", "empty", "empty_like", "cat", "stack")}
def twrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        fr = traceback.extract_stack(limit=5)[:-1]
        sites["Tensor." + name + " " + " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in reversed(fr))] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
    return orig
torigs = {n: twrap(n) for n in ("zero_", "fill_", "copy_", "clone", "contiguous")}
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for n, o in origs.items(): setattr(torch, n, o)
for n, o in torigs.items(): setattr(torch.Tensor, n, o)
print("allocation / fill / copy call sites in one step:")
for k, v in sites.most_common(60): print(f"  {v:4d}  {k}")
ev = [e for e in prof.events() if any(w in e.name.lower() for w in ("fill", "zero", "copy", "memcpy", "memset"))]
c = collections.Counter()
for e in ev:
    st = [s for s in (e.stack or []) if "signal_amd" in s or "bench" in s][:3]
    c[(e.name, " <- ".join(st))] += 1
print("fill/zero ops in one step:")
for (n, s), v in c.most_common(25): print(f"  {v:4d} {n} | {s}")
