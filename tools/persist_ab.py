#!/usr/bin/env python3
"""Same-process A/B of the persistent 192x256 NT kernel against the per-tile kernels (sig_tune_nt_persist 1 / 0), interleaved
rounds, at the B = 64 shapes it is legal for.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("AB_M", "24768"))
Mp = ops.pad_rows(M)
dt = torch.float16 if os.environ.get("AB_DT") == "fp16" else torch.bfloat16


def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


cases = [("qkv", 2304, 768, ops.BIAS_BF16, False, False), ("c_fc_infer", 3072, 768, ops.BIAS_GELU_BF16, False, False),
         ("c_fc_train", 3072, 768, ops.BIAS_GELU_BF16, True, False), ("dgelu", 3072, 768, ops.DGELU_BF16, True, True)]
res = {}
for rnd in range(3):
    for name, n, k, epi, has_aux, aux_in in cases:
        a = torch.randn(Mp, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * 0.02).to(dt)
        bias = torch.randn(n, device=dev)
        out = torch.zeros(Mp, n, device=dev, dtype=dt)
        aux = (torch.randn(Mp, n, device=dev).to(dt) if aux_in else torch.zeros(Mp, n, device=dev, dtype=dt)) if has_aux else None
        for persist in (0, 3):
            prev = lib.sig_tune_nt_persist(persist)
            us = timeit(lambda: ops.gemm_nt(a, w, M, epi, out, bias=None if epi == ops.DGELU_BF16 else bias, aux=aux))
            lib.sig_tune_nt_persist(prev)
            res.setdefault((name, persist), []).append(us)
        del a, w, out, aux
for (name, persist), v in sorted(res.items()):
    n, k = {"qkv": (2304, 768)}.get(name, (3072, 768))
    best = min(v)
    print(f"{name:11s} persist={persist}  us per launch: " + " ".join(f"{x:7.1f}" for x in v) + f"   best {best:7.1f} = {2 * M * n * k / best / 1e6:6.0f} TFLOP/s")
