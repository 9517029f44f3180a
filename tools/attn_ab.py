#!/usr/bin/env python3
"""Timing of the attention kernels at the B = 64 shapes (S = 192 sequences, L = 129, 12 heads). GPU box only.
SIG_ATTN_BWD_X1=0 selects the nine-tile backward for an A/B (one setting per process: the switch is read once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0")
S, L, H = int(os.environ.get("AB_S", "192")), 129, int(os.environ.get("AB_H", "12"))
# (AB_S=2304 AB_H=1: the same bytes and flops with every (sequence, head) operand block CONTIGUOUS in memory -- what a head-major
#  qkv layout would give the kernels)
Mp = ops.pad_rows(S * L)
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
qkv = torch.randn(Mp, 3 * 64 * H, device=dev).to(torch.bfloat16)
o = torch.zeros(Mp, 64 * H, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(S, H, L, device=dev)
do = torch.randn(Mp, 64 * H, device=dev).to(torch.bfloat16)
dqkv = torch.zeros_like(qkv)
res = []
for _ in range(3):
    f = timeit(lambda: ops.attn_fwd(qkv, o, lse, S, L, H))
    b = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, S, L, H))
    res.append((f, b))
print(f"S={S} H={H} " + "SIG_ATTN_BWD_X1=%s  fwd us: %s   bwd us: %s" % (os.environ.get("SIG_ATTN_BWD_X1", "default"),
      " ".join(f"{f:.1f}" for f, _ in res), " ".join(f"{b:.1f}" for _, b in res)))
