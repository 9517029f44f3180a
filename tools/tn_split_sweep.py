#!/usr/bin/env python3
"""GPU box: sweep the row-chunk count of the 128x128 TN kernel on the small weight-gradient shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0"); M = 24768; Mp = ops.pad_rows(M)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (i, j) in [(768, 768), (512, 768), (768, 512), (2304, 768), (768, 3072)]:
    p = torch.randn(Mp, i, device=dev).to(torch.bfloat16); q = torch.randn(Mp, j, device=dev).to(torch.bfloat16)
    out = torch.zeros(i, j, device=dev)
    res = []
    for split in (0, 4, 5, 6, 7, 8, 10, 12, 14):
        ms = timeit(lambda: ops.gemm_tn(p, q, out, split=split))
        res.append(f"{split}:{ms*1e3:.0f}us")
    print(i, j, " ".join(res))
