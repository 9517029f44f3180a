#!/usr/bin/env python3
"""Soak: N train steps at the benched size on ONE fixed synthetic batch (it must be memorised: the loss has to fall), both operand
types; checks finiteness of loss / parameters every 20 steps and, for fp16, that the loss scaler neither overflows repeatedly nor stalls.
usage: python tools/train_soak.py [--steps 120] [--batch 64]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from signal_amd.engine.trainer import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda:0")
for dtype in ("bf16", "fp16"):
    cfg, model = bench.build_model(dev, dtype)
    img, vid, cam = bench.synthetic(cfg, a.batch, dev, 1234)
    ts = TrainStep(cfg, model, num_classes=171, world_size=1)
    for g in ts.optimizer.param_groups:            # the schedule's epoch-1 value (warm-up) is tiny; use the base rates
        g["lr"] = g.get("initial_lr", g["lr"])
    t0 = time.time()
    hist = []
    for i in range(a.steps):
        loss = ts.step(img, vid, cam)
        if i % 20 == 0 or i == a.steps - 1:
            v = float(loss)
            ok = bool(torch.isfinite(model.hip.flat.data).all())
            hist.append(v)
            sc = ts.scaler.state.cpu().tolist() if ts.scaler is not None else None
            print(f"[{dtype}] step {i:4d} loss {v:9.4f} params finite {ok}" + (f" scale {sc[0]:.0f} applied {sc[4]:.0f}" if sc else ""), flush=True)
            assert ok and v == v
    torch.cuda.synchronize()
    print(f"[{dtype}] {a.steps} steps in {time.time() - t0:.1f} s; loss {hist[0]:.4f} -> {hist[-1]:.4f}")
    assert hist[-1] < hist[0], "the fixed batch must be learnt"
    if ts.scaler is not None:
        st = ts.scaler.state.cpu().tolist()
        assert st[4] >= 0.9 * a.steps, f"too many skipped steps: {st}"
    del ts, model
print("soak ok")
