#!/usr/bin/env python3
"""GPU box: do a dgrad (NT) and a wgrad (TN) of the same layer overlap when issued on two streams?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0"); M = 24768; Mp = ops.pad_rows(M)
def mk(r, c): return torch.randn(r, c, device=dev).to(torch.bfloat16)
cases = {"c_fc (dgrad N768 K3072 | wgrad 3072x768)": (768, 3072), "qkv (dgrad N768 K2304 | wgrad 2304x768)": (768, 2304),
         "out_proj (dgrad N768 K768 | wgrad 768x768)": (768, 768)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, (n, k) in cases.items():
    dy, w, x = mk(Mp, k), (mk(n, k) * 0.02), mk(Mp, n)
    dx = torch.zeros(Mp, n, device=dev, dtype=torch.bfloat16); dw = torch.zeros(k, n, device=dev)
    def seq():
        ops.gemm_nt(dy, w, M, ops.BF16, dx); ops.gemm_tn(dy, x, dw)
    def par():
        e = torch.cuda.Event(); e.record()
        s1.wait_event(e); s2.wait_event(e)
        with torch.cuda.stream(s1): ops.gemm_nt(dy, w, M, ops.BF16, dx)
        with torch.cuda.stream(s2): ops.gemm_tn(dy, x, dw)
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    def t(fn, it=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3
    print(f"{name}: sequential {t(seq):.0f} us, two streams {t(par):.0f} us")
