#!/usr/bin/env python3
"""Diagnostic (GPU box): every kernel / copy of ONE steady-state train step (B = 64, bf16), counted by name, and for the torch
(non-signal_amd) ones the Python call sites that issued them.  usage: tools/step_launches.py [train|infer]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
cfg, model = bench.build_model(dev, "bf16")
img, vid, cam = bench.synthetic(cfg, int(os.environ.get("BATCH", "64")), dev, 1234)
if len(sys.argv) < 2 or sys.argv[1] == "train":
    from signal_amd.engine.trainer import TrainStep
    ts = TrainStep(cfg, model, num_classes=171, world_size=1)
    step = lambda: ts.step(img, vid, cam)
else:
    def step():
        with torch.no_grad():
            return model(img, cam_label=cam, training=False)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
cnt, tot = collections.Counter(), collections.Counter()
for e in ev:
    cnt[e.name[:90]] += 1; tot[e.name[:90]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print(f"{len(ev)} device activities in one step, {sum(tot.values()) / 1e3:.2f} ms summed")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{cnt[k]:5d} x {v / max(cnt[k], 1):9.1f} us  = {v / 1e3:8.3f} ms  {k}")
# call sites of the torch-issued ones: CPU ops (aten::*) that launched something, with their Python stacks
print("\n--- aten ops that launched device work, by Python call site")
sites = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and getattr(e, "kernels", None):
        if not e.kernels: continue
        st = [s for s in (e.stack or []) if "signal_amd" in s or "bench.py" in s or "tools/" in s][:2]
        sites[(e.name, " <- ".join(s.split("/")[-1] for s in st))] += len(e.kernels)
for (n, s), c in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d}  {n:28s} {s}")
