#!/usr/bin/env python3
"""The f32 + residual NT GEMMs of a block (out_proj K = 768, c_proj K = 3072; N = 768, M = 24768) in the inference form (residual
read from the buffer that is written) and the training form (separate residual buffer); a few launches each, for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0")
M = 24768; Mp = ops.pad_rows(M)
g = torch.Generator().manual_seed(0)
def run(k, inplace, iters=6):
    a = (torch.randn(Mp, k, generator=g) * 0.5).bfloat16().to(dev)
    w = (torch.randn(768, k, generator=g) * 0.02).bfloat16().to(dev)
    bias = torch.randn(768, generator=g).to(dev)
    x = torch.randn(Mp, 768, generator=g).to(dev)
    out = x if inplace else torch.zeros(Mp, 768, device=dev)
    for _ in range(2): ops.gemm_nt(a, w, M, ops.BIAS_RES_F32, out, bias=bias, res=x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_nt(a, w, M, ops.BIAS_RES_F32, out, bias=bias, res=x)
    e1.record(); torch.cuda.synchronize()
    print(f"K={k} {'in place' if inplace else 'separate residual'}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us")
order = sys.argv[1] if len(sys.argv) > 1 else "768i,768s,3072i,3072s"
for tok in order.split(","):
    run(int(tok[:-1]), tok[-1] == "i")
