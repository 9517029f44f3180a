#!/usr/bin/env python3
"""Diagnostic: LayerNorm forward on the model's shapes against a torch fp32 evaluation (max abs error per output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for m, d in [(3096, 768), (3097, 768), (24768, 768), (24, 512), (25, 512), (1, 768), (7, 768), (2064, 512)]:
    x = (torch.randn(m, d, generator=g) * 3 + 1.5).to(dev)
    gam, bet = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    Mp = ops.pad_rows(m)
    xb = torch.zeros(Mp, d, device=dev); xb[:m] = x
    yb = torch.zeros(Mp, d, device=dev, dtype=torch.bfloat16); yf = torch.zeros(Mp, d, device=dev)
    mean, rstd = torch.zeros(Mp, device=dev), torch.zeros(Mp, device=dev)
    ops.layernorm_fwd(xb, gam, bet, m, y_bf16=yb, y_f32=yf, mean=mean, rstd=rstd)
    ref = torch.nn.functional.layer_norm(x, (d,), gam, bet, 1e-5)
    mu = x.mean(1); rs = 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
    print(f"M={m:6d} D={d}: y_f32 {float((yf[:m]-ref).abs().max()):.2e}  y_bf16 {float((yb[:m].float()-ref).abs().max()):.2e}  mean {float((mean[:m]-mu).abs().max()):.2e}  rstd {float((rstd[:m]-rs).abs().max()):.2e}  pad rows touched: {bool(yf[m:].abs().any()) or bool(mean[m:].abs().any())}")
    yb2 = torch.zeros(Mp, d, device=dev, dtype=torch.bfloat16); mean2, rstd2 = torch.zeros(Mp, device=dev), torch.zeros(Mp, device=dev)
    ops.layernorm_fwd(xb, gam, bet, m, y_bf16=yb2, mean=mean2, rstd=rstd2)        # the ViT block's call: 16-bit output + statistics
    print(f"           16-bit only: y {float((yb2[:m].float()-ref).abs().max()):.2e} (== two-output call: {bool(torch.equal(yb2, yb))})  mean {float((mean2[:m]-mu).abs().max()):.2e}  rstd {float((rstd2[:m]-rs).abs().max()):.2e}")
    # in place (y_f32 = x)
    xc = xb.clone()
    ops.layernorm_fwd(xc, gam, bet, m, y_f32=xc)
    print(f"           in place: {float((xc[:m]-ref).abs().max()):.2e}")
