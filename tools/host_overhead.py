#!/usr/bin/env python3
"""Is the step host-bound?  Compares host enqueue time per step with the synchronised wall time. GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "train"
cfg, model = bench.build_model(dev, wl)
img, vid, cam = bench.synthetic(cfg, 64, dev, 1234)
if wl == "train":
    from signal_amd.engine.trainer import TrainStep
    ts = TrainStep(cfg, model, num_classes=171)
    step = lambda: ts.step(img, vid, cam)
else:
    def step():
        with torch.no_grad():
            return model(img, cam_label=cam, training=False)
for _ in range(3): step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{wl}: host enqueue {1e3*(t1-t0)/n:.2f} ms/step, wall {1e3*(t2-t0)/n:.2f} ms/step")
if wl == "train":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
