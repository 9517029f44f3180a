#!/usr/bin/env python3
"""Diagnostic (GPU box): how much of the gap between a GEMM's time in the kernel micro-benchmark (same buffers every launch) and
inside the train step is the state of L2 / Infinity Cache?  Each GEMM is timed (HIP events around the GEMM only) in four states:
  warm      the same operands launch after launch (tools/bench_kernels.py's state)
  produced  a copy kernel rewrites the A operand (and the residual) right before every launch -- the train step's state for an
            operand the previous kernel produced
  cold      512 MB of unrelated writes + reads between launches (everything evicted)
  rotate    four operand sets used round-robin (each launch reads data last touched three launches ago)
usage: tools/cache_state_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0")
M = 24768; Mp = ops.pad_rows(M); dt = torch.bfloat16
junk_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev); junk_b = torch.empty_like(junk_a)

def mk(n, k, epi):
    a = torch.randn(Mp, k, device=dev).to(dt); w = (torch.randn(n, k, device=dev) * 0.02).to(dt)
    f32 = epi in (ops.BIAS_RES_F32,)
    out = torch.zeros(Mp, n, device=dev, dtype=torch.float32 if f32 else dt)
    res = torch.randn(Mp, n, device=dev) if f32 else None
    bias = torch.randn(n, device=dev)
    return dict(a=a, w=w, out=out, res=res, bias=bias, a_src=a.clone(), res_src=None if res is None else res.clone())

def launch(s, epi):
    kw = {}
    if epi != ops.BF16: kw["bias"] = s["bias"]
    if s["res"] is not None: kw["res"] = s["res"]
    ops.gemm_nt(s["a"], s["w"], M, epi, s["out"], **kw)

def timed(sets, epi, pre, iters=24):
    tot = 0.0
    for i in range(iters + 4):
        s = sets[i % len(sets)]
        pre(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(s, epi); e1.record(); torch.cuda.synchronize()
        if i >= 4: tot += e0.elapsed_time(e1)
    return tot / iters * 1e3

def produce(s):
    s["a"].copy_(s["a_src"])
    if s["res"] is not None: s["res"].copy_(s["res_src"])
def evict(s):
    junk_b.copy_(junk_a)

print(f"{'GEMM':34s} {'warm':>8s} {'produced':>9s} {'cold':>8s} {'rotate':>8s}   us per launch")
for name, n, k, epi in [("qkv (persistent, N=2304 K=768)", 2304, 768, ops.BIAS_BF16), ("dgrad c_fc (320-row, N=768 K=3072)", 768, 3072, ops.BF16),
                        ("dgrad qkv (N=768 K=2304)", 768, 2304, ops.BF16), ("out_proj f32+res (N=768 K=768)", 768, 768, ops.BIAS_RES_F32),
                        ("c_proj f32+res (N=768 K=3072)", 768, 3072, ops.BIAS_RES_F32)]:
    sets = [mk(n, k, epi) for _ in range(4)]
    r = [timed(sets[:1], epi, lambda s: None), timed(sets[:1], epi, produce), timed(sets[:1], epi, evict), timed(sets, epi, lambda s: None)]
    print(f"{name:34s} " + " ".join(f"{x:8.1f}" for x in r), flush=True)
    del sets
