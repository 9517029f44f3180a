#!/usr/bin/env python3
"""Wall-clock check of the grouped weight-gradient launch (kernel + reduce) at the benched shapes: N back-to-back launches
between two torch events, next to the library's own per-launch HIP-event timing (sig_prof_*), with and without rocprofv3
around the process.  usage: python tools/tn_group_probe.py [--iters 30] [--split S]"""
import argparse, ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from signal_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rows", type=int, default=24832)
a = ap.parse_args()
dev = torch.device("cuda:0")
D, F, mr = 768, 3072, a.rows
g = torch.Generator().manual_seed(0)
mk = lambda c: (torch.randn(mr, c, generator=g) * 0.1).bfloat16().to(dev)
pairs = [(mk(3 * D), mk(D)), (mk(D), mk(D)), (mk(F), mk(D)), (mk(D), mk(F))]
outs = [torch.zeros(p.shape[1], q.shape[1], device=dev) for p, q in pairs]
jobs = [(p, q, o) for (p, q), o in zip(pairs, outs)]
flops = sum(2.0 * mr * p.shape[1] * q.shape[1] for p, q in pairs)
lib = _lib.load()
for label, fn in (("grouped (1 + 1 launches)", lambda: ops.gemm_tn_grouped(jobs)),
                  ("per weight (4 + 4 launches)", lambda: [ops.gemm_tn(p, q, o) for p, q, o in jobs])):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    cls = 101 if label.startswith("grouped") else 100
    _lib.call("sig_prof_begin", cls, 0, 0, 8 * a.iters)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    _lib.call("sig_prof_end", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
    wall = e0.elapsed_time(e1) / a.iters * 1e3
    print(f"{label:30s} wall per iteration (kernels + reduces + launch gaps) {wall:8.1f} us = {flops / wall / 1e6:7.1f} TFLOP/s | "
          f"library HIP events: {n.value} main-kernel launches, sum per iteration {ms.value / a.iters * 1e3:8.1f} us")
