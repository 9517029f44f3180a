#!/usr/bin/env python3
"""Micro-benchmark of the hot kernels at the B=64 shapes (M = 3*64*129 = 24768 tokens). GPU box only."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops

dev = torch.device("cuda:0")
M = 24768
Mp = ops.pad_rows(M)

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

res = {}
for name, n, k, epi in [("qkv", 2304, 768, ops.BIAS_BF16), ("out_proj", 768, 768, ops.BIAS_RES_F32),
                        ("c_fc", 3072, 768, ops.BIAS_GELU_BF16), ("c_proj", 768, 3072, ops.BIAS_RES_F32)]:
    a = torch.randn(Mp, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    f32 = epi in (ops.BIAS_RES_F32,)
    out = torch.zeros(Mp, n, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    aux = torch.zeros(Mp, n, device=dev, dtype=torch.bfloat16) if epi == ops.BIAS_GELU_BF16 else None
    ms = timeit(lambda: ops.gemm_nt(a, w, M, epi, out, bias=bias, res=out if f32 else None, aux=aux))
    res["nt_" + name] = dict(ms=round(ms, 4), tflops=round(2 * M * n * k / ms / 1e9, 1))
    if f32:     # as in the model: the residual comes from ANOTHER buffer (x_in -> x_mid -> x_out), not in place
        r2 = torch.randn(Mp, n, device=dev)
        ms = timeit(lambda: ops.gemm_nt(a, w, M, epi, out, bias=bias, res=r2))
        res["nt_" + name + "_sep"] = dict(ms=round(ms, 4), tflops=round(2 * M * n * k / ms / 1e9, 1))
        del r2
    # wgrad of the same layer: dW[n,k] = dY^T X
    dy = torch.randn(Mp, n, device=dev).to(torch.bfloat16); dy[M:] = 0
    dw = torch.zeros(n, k, device=dev)
    ms = timeit(lambda: ops.gemm_tn(dy, a, dw))
    res["tn_" + name] = dict(ms=round(ms, 4), tflops=round(2 * M * n * k / ms / 1e9, 1))
    del a, w, out, dy, dw

# dgrad shapes (plain bf16 output; the GELU' one reads the saved pre-activation)
for name, n, k, epi in [("dgrad_qkv", 768, 2304, ops.BF16), ("dgrad_out_proj", 768, 768, ops.BF16), ("dgrad_c_fc", 768, 3072, ops.BF16),
                        ("dgrad_c_proj", 3072, 768, ops.DGELU_BF16)]:
    a = torch.randn(Mp, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.zeros(Mp, n, device=dev, dtype=torch.bfloat16)
    aux = torch.randn(Mp, n, device=dev).to(torch.bfloat16) if epi == ops.DGELU_BF16 else None
    cs = torch.zeros(n, device=dev)
    ms = timeit(lambda: ops.gemm_nt(a, w, M, epi, out, aux=aux))
    res["nt_" + name] = dict(ms=round(ms, 4), tflops=round(2 * M * n * k / ms / 1e9, 1))
    del a, w, out

S, L, H = 192, 129, 12
qkv = torch.randn(Mp, 2304, device=dev).to(torch.bfloat16)
o = torch.zeros(Mp, 768, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(S, H, L, device=dev)
ms = timeit(lambda: ops.attn_fwd(qkv, o, lse, S, L, H))
res["attn_fwd"] = dict(ms=round(ms, 4), gbps=round((Mp * 2304 * 2 + Mp * 768 * 2) / ms / 1e6, 1))
do = torch.randn(Mp, 768, device=dev).to(torch.bfloat16)
dqkv = torch.zeros_like(qkv)
ms = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, S, L, H))
res["attn_bwd"] = dict(ms=round(ms, 4))
x = torch.randn(M, 768, device=dev); g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
y = torch.zeros(M, 768, device=dev, dtype=torch.bfloat16); mean = torch.zeros(M, device=dev); rstd = torch.zeros(M, device=dev)
ms = timeit(lambda: ops.layernorm_fwd(x, g, b, M, y_bf16=y, mean=mean, rstd=rstd))
res["ln_fwd"] = dict(ms=round(ms, 4), gbps=round(M * 768 * 6 / ms / 1e6, 1))
dx = torch.zeros(M, 768, device=dev); dxb = torch.zeros(M, 768, device=dev, dtype=torch.bfloat16)
dg = torch.zeros(768, device=dev); db = torch.zeros(768, device=dev)
ms = timeit(lambda: ops.layernorm_bwd(y, x, g, mean, rstd, M, dres=x, dx_f32=dx, dx_bf16=dxb, dgamma=dg, dbeta=db))
res["ln_bwd"] = dict(ms=round(ms, 4), gbps=round(M * 768 * (2 + 4 + 4 + 4 + 2) / ms / 1e6, 1))
for k, v in res.items(): print(k, v)
json.dump(res, open(os.path.join("gpurun_out", "bench_kernels.json"), "w"), indent=1)
