#!/usr/bin/env python3
"""Diagnostic (GPU box): each deterministic-by-design kernel is run N times on identical inputs; any bitwise difference
between runs is a race."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0")
M = 24768; Mp = ops.pad_rows(M); N_RUNS = int(os.environ.get("RUNS", "12"))
def mk(r, c, s=1.0): return (torch.randn(r, c, device=dev) * s).to(torch.bfloat16)
def check(name, fn, outs):
    fn(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    bad = 0; worst = 0.0
    for _ in range(N_RUNS):
        for o in outs: o.zero_()
        fn(); torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            if not torch.equal(o, r):
                bad += 1
                worst = max(worst, float((o.float() - r.float()).abs().max()))
    print(f"{name:46s} {'DETERMINISTIC' if bad == 0 else f'DIFFERS in {bad} comparisons, max abs diff {worst:.3e}'}")
for name, n, k, epi in [("nt qkv (256, BIAS_BF16)", 2304, 768, ops.BIAS_BF16), ("nt c_fc (256, BIAS_GELU_BF16 + u)", 3072, 768, ops.BIAS_GELU_BF16),
                        ("nt c_proj dgrad (256, DGELU)", 3072, 768, ops.DGELU_BF16), ("nt out_proj (128, BIAS_RES_F32)", 768, 768, ops.BIAS_RES_F32),
                        ("nt c_proj (128, BIAS_RES_F32)", 768, 3072, ops.BIAS_RES_F32), ("nt dgrad qkv (128, BF16)", 768, 2304, ops.BF16),
                        ("nt dgrad c_fc (128, BF16)", 768, 3072, ops.BF16)]:
    a, w, bias = mk(Mp, k), mk(n, k, 0.02), torch.randn(n, device=dev)
    f32 = epi == ops.BIAS_RES_F32
    out = torch.zeros(Mp, n, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    res = torch.randn(Mp, n, device=dev) if f32 else None
    aux = None
    outs = [out]
    if epi == ops.BIAS_GELU_BF16: aux = torch.zeros(Mp, n, device=dev, dtype=torch.bfloat16); outs.append(aux)
    if epi == ops.DGELU_BF16: aux = mk(Mp, n)
    check(name, lambda: ops.gemm_nt(a, w, M, epi, out, bias=bias if epi not in (ops.BF16, ops.DGELU_BF16) else None, res=res, aux=aux), [out] if epi == ops.DGELU_BF16 else outs)
    del a, w, out
for name, i, j in [("tn c_fc wgrad (256 + workspace reduce)", 3072, 768), ("tn qkv wgrad (256)", 2304, 768)]:
    p_, q_ = mk(Mp, i), mk(Mp, j); out = torch.zeros(i, j, device=dev)
    check(name, lambda: ops.gemm_tn(p_, q_, out), [out])
S, L, H = 192, 129, 12
qkv = mk(Mp, 2304); o = torch.zeros(Mp, 768, device=dev, dtype=torch.bfloat16); lse = torch.zeros(S, H, L, device=dev)
check("attn_fwd", lambda: ops.attn_fwd(qkv, o, lse, S, L, H), [o, lse])
ops.attn_fwd(qkv, o, lse, S, L, H)
o2, lse2 = o.clone(), lse.clone()
do = mk(Mp, 768); dqkv = torch.zeros_like(qkv)
check("attn_bwd", lambda: ops.attn_bwd(qkv, o2, do, lse2, dqkv, S, L, H), [dqkv])
x = torch.randn(M, 768, device=dev); g = torch.randn(768, device=dev); b = torch.randn(768, device=dev)
y = torch.zeros(M, 768, device=dev, dtype=torch.bfloat16); mean = torch.zeros(M, device=dev); rstd = torch.zeros(M, device=dev)
check("layernorm_fwd", lambda: ops.layernorm_fwd(x, g, b, M, y_bf16=y, mean=mean, rstd=rstd), [y, mean, rstd])
ops.layernorm_fwd(x, g, b, M, y_bf16=y, mean=mean, rstd=rstd)
y2, m2, r2 = y.clone(), mean.clone(), rstd.clone()
dx = torch.zeros(M, 768, device=dev); dxb = torch.zeros(M, 768, device=dev, dtype=torch.bfloat16)
dy = mk(M, 768); dres = torch.randn(M, 768, device=dev)
check("layernorm_bwd (dx only)", lambda: ops.layernorm_bwd(dy, x, g, m2, r2, M, dres=dres, dx_f32=dx, dx_bf16=dxb), [dx, dxb])
dg, db = torch.zeros(768, device=dev), torch.zeros(768, device=dev)
check("layernorm_bwd (+ dgamma/dbeta, partial rows + reduce)", lambda: ops.layernorm_bwd(dy, x, g, m2, r2, M, dres=dres, dx_f32=dx, dx_bf16=dxb, dgamma=dg, dbeta=db), [dx, dxb, dg, db])
# fp16 operands through the same kernels (template instantiations of their own)
ah, wh = torch.randn(Mp, 768, device=dev).half(), (torch.randn(3072, 768, device=dev) * 0.02).half()
uh, oh = torch.randn(Mp, 3072, device=dev).half(), torch.zeros(Mp, 3072, device=dev, dtype=torch.float16)
check("nt c_proj dgrad (256, DGELU, fp16)", lambda: ops.gemm_nt(ah, wh, M, ops.DGELU_BF16, oh, aux=uh), [oh])
ph, qh = torch.randn(Mp, 3072, device=dev).half(), torch.randn(Mp, 768, device=dev).half(); outh = torch.zeros(3072, 768, device=dev)
check("tn c_fc wgrad (256x16, fp16)", lambda: ops.gemm_tn(ph, qh, outh), [outh])
