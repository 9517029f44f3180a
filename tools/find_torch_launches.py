#!/usr/bin/env python3
"""Diagnostic (GPU box): every aten op that touches device memory inside ONE steady-state train step, with the signal_amd call site
that issued it (TorchDispatchMode; the backward runs on the calling thread for the trace).  Views / metadata ops are skipped."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
dev = torch.device("cuda:0")
cfg, model = bench.build_model(dev, "bf16")
img, vid, cam = bench.synthetic(cfg, 64, dev, 1234)
from signal_amd.engine.trainer import TrainStep
ts = TrainStep(cfg, model, num_classes=171, world_size=1)
for _ in range(4): ts.step(img, vid, cam)
torch.cuda.synchronize()
SKIP = ("view", "reshape", "expand", "slice", "select", "as_strided", "detach", "alias", "unbind", "t.default", "transpose", "permute",
        "squeeze", "unsqueeze", "empty", "size", "stride", "is_", "numel", "storage_offset", "sym_", "_local_scalar", "split", "narrow",
        "unsafe_view", "lift_fresh", "set_", "resize_", "item", "contiguous", "_unsafe_view", "chunk", "unfold")
sites = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            fr = [f for f in traceback.extract_stack() if ("signal_amd" in f.filename or "bench.py" in f.filename) and "tools/" not in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:]))
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))
torch.autograd.set_multithreading_enabled(False)
with Spy():
    ts.step(img, vid, cam)
torch.cuda.synchronize()
print(f"{sum(sites.values())} data-touching aten calls in one train step")
for (n, w), c in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c:4d}  {n:34s} {w}")
