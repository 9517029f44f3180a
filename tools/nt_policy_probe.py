#!/usr/bin/env python3
"""GPU box: time the wide NT shapes under the tile choice given by the environment (SIG_GEMM_TILE / SIG_GEMM_BM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0"); M = 24768; Mp = ops.pad_rows(M)
def timeit(fn, it=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
res = []
for name, n, k, epi, aux in [("qkv", 2304, 768, ops.BIAS_BF16, False), ("c_fc_infer", 3072, 768, ops.BIAS_GELU_BF16, False),
                             ("c_fc_train", 3072, 768, ops.BIAS_GELU_BF16, True), ("dgelu", 3072, 768, ops.DGELU_BF16, True)]:
    a = torch.randn(Mp, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(n, device=dev); out = torch.zeros(Mp, n, device=dev, dtype=torch.bfloat16)
    ax = torch.randn(Mp, n, device=dev).to(torch.bfloat16) if aux else None
    us = timeit(lambda: ops.gemm_nt(a, w, M, epi, out, bias=None if epi == ops.DGELU_BF16 else bias, aux=ax))
    res.append(f"{name} {us:.1f}us")
print("  ".join(res))
