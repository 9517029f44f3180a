#!/usr/bin/env python3
"""G8: learning-rate schedules.  Runs the reference's own solver/scheduler_factory.py, cosine_lr.py, scheduler.py and
lr_scheduler310.py (imported by path from /root/reference; plain torch) on a three-group dummy optimizer and records the
lr of every group after each scheduler step.  Output: tests/golden/g8_lr_schedule.npz (a few KB of numbers, no code).

    python tests/golden/make_golden_sched.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SIGNAL_REFERENCE", "/root/reference")


def main():
    sys.path.insert(0, REF)
    pkg = types.ModuleType("solver")
    pkg.__path__ = [os.path.join(REF, "solver")]
    sys.modules["solver"] = pkg
    fac = importlib.import_module("solver.scheduler_factory")
    ms = importlib.import_module("solver.lr_scheduler310")
    out = {}
    # (tag, BASE_LR, WARMUP_ITERS, MAX_EPOCHS) of configs/RGBNT201 and configs/RGBNT100; groups = weight, bias (2x), backbone
    for tag, base, warm, epochs in [("rgbnt201", 0.00035, 10, 50), ("rgbnt100", 0.0007, 5, 30), ("nowarm", 0.001, 0, 12)]:
        cfg = types.SimpleNamespace(SOLVER=types.SimpleNamespace(MAX_EPOCHS=epochs, BASE_LR=base, WARMUP_ITERS=warm))
        ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(3)]
        opt = torch.optim.Adam([{"params": [ps[0]], "lr": base}, {"params": [ps[1]], "lr": 2 * base}, {"params": [ps[2]], "lr": 5e-6}])
        sch = fac.create_scheduler(cfg, opt)
        rows = [[g["lr"] for g in opt.param_groups]]            # state right after construction
        clean = []
        for epoch in range(1, epochs + 4):                       # processor.py:135 steps with epoch = 1..MAX (and beyond)
            sch.step(epoch)
            rows.append([g["lr"] for g in opt.param_groups])
            clean.append(sch._get_lr(epoch))
        out[f"{tag}_after_step"] = np.array(rows, dtype=np.float64)
        out[f"{tag}_noise_free"] = np.array(clean, dtype=np.float64)
        out[f"{tag}_cfg"] = np.array([base, warm, epochs], dtype=np.float64)
    # MSVR310: WarmupMultiStepLR(STEPS [20, 40], GAMMA 0.1, WARMUP_FACTOR 0.01, WARMUP_ITERS 0 and 10, linear)
    for tag, wit in [("msvr310", 0), ("msvr310_warm10", 10)]:
        ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
        opt = torch.optim.Adam([{"params": [ps[0]], "lr": 5e-6}, {"params": [ps[1]], "lr": 5e-4}])
        sch = ms.WarmupMultiStepLR(opt, [20, 40], 0.1, 0.01, wit, "linear")
        rows = [[g["lr"] for g in opt.param_groups]]
        for _ in range(50):
            opt.step()
            sch.step()
            rows.append([g["lr"] for g in opt.param_groups])
        out[f"{tag}_lr"] = np.array(rows, dtype=np.float64)
    path = os.path.join(HERE, "g8_lr_schedule.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
