#!/usr/bin/env python3
"""attn_fwd / attn_bwd at the benched shape (S = 192 sequences x 12 heads, L = 129): time per launch. Env knobs are read by
the library once per process, so run one process per setting (tools/ab: SIG_ATTN_STAGGER)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0")
S, L, H = 192, 129, 12
Mp = ops.pad_rows(S * L)
qkv = torch.randn(Mp, 2304, device=dev).to(torch.bfloat16); o = torch.zeros(Mp, 768, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(S, H, L, device=dev)
do = torch.randn(Mp, 768, device=dev).to(torch.bfloat16); dqkv = torch.zeros_like(qkv)
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
f = timeit(lambda: ops.attn_fwd(qkv, o, lse, S, L, H))
b = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, S, L, H))
print(f"SIG_ATTN_STAGGER={os.environ.get('SIG_ATTN_STAGGER', 'default')} SIG_ATTN_FWD_STAGGER={os.environ.get('SIG_ATTN_FWD_STAGGER', 'default')}: attn_fwd {f:.1f} us  attn_bwd {b:.1f} us")
