#!/usr/bin/env python3
"""GPU box: do the three modality streams of the ViT forward overlap usefully when each runs on its own HIP stream
(S = 64 sequences, 78-tile GEMMs on a third of the chip each) instead of batched in M (S = 192, one stream)?
The single-round f32-output GEMMs end in an HBM burst nothing overlaps; independent streams would
de-phase those bursts.  Times LAYERS blocks of the inference forward (LN, qkv, attention, out_proj, LN, c_fc, c_proj)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops

dev = torch.device("cuda:0")
B, L, H, D, F = 64, 129, 12, 768, 3072
LAYERS = int(os.environ.get("LAYERS", "6"))
BF = torch.bfloat16


def weights():
    g = lambda r, c: (torch.randn(r, c, device=dev) * 0.02).to(BF)
    return dict(w_in=g(3 * D, D), b_in=torch.zeros(3 * D, device=dev), w_out=g(D, D), b_out=torch.zeros(D, device=dev),
                w_fc=g(F, D), b_fc=torch.zeros(F, device=dev), w_proj=g(D, F), b_proj=torch.zeros(D, device=dev),
                ln_w=torch.ones(D, device=dev), ln_b=torch.zeros(D, device=dev))


W = weights()


def bufs(S):
    M = S * L
    Mp = ops.pad_rows(M)
    z = lambda c, dt: torch.zeros(Mp, c, device=dev, dtype=dt)
    return dict(S=S, M=M, x=torch.randn(Mp, D, device=dev), h=z(D, BF), qkv=z(3 * D, BF), attn=z(D, BF), g=z(F, BF),
                lse=torch.zeros(S * H * L, device=dev), mean=torch.zeros(Mp, device=dev), rstd=torch.zeros(Mp, device=dev))


def block(b):
    M, S = b["M"], b["S"]
    ops.layernorm_fwd(b["x"], W["ln_w"], W["ln_b"], M, y_bf16=b["h"], mean=b["mean"], rstd=b["rstd"])
    ops.gemm_nt(b["h"], W["w_in"], M, ops.BIAS_BF16, b["qkv"], bias=W["b_in"])
    ops.attn_fwd(b["qkv"], b["attn"], b["lse"], S, L, H)
    ops.gemm_nt(b["attn"], W["w_out"], M, ops.BIAS_RES_F32, b["x"], bias=W["b_out"], res=b["x"])
    ops.layernorm_fwd(b["x"], W["ln_w"], W["ln_b"], M, y_bf16=b["h"], mean=b["mean"], rstd=b["rstd"])
    ops.gemm_nt(b["h"], W["w_fc"], M, ops.BIAS_GELU_BF16, b["g"], bias=W["b_fc"])
    ops.gemm_nt(b["g"], W["w_proj"], M, ops.BIAS_RES_F32, b["x"], bias=W["b_proj"], res=b["x"])


def timed(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / it * 1e3 / LAYERS


big = bufs(3 * B)
def batched():
    for _ in range(LAYERS): block(big)

parts = [bufs(B) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
def split():
    ev = torch.cuda.Event(); ev.record()
    for s in streams: s.wait_event(ev)
    for _ in range(LAYERS):                      # layer-major issue order: the three streams' kernels interleave in the queues
        for p, s in zip(parts, streams):
            with torch.cuda.stream(s): block(p)
    for s in streams: torch.cuda.current_stream().wait_stream(s)
def split_seq():                                 # the same three M = 8256 problems one after another on one stream
    for _ in range(LAYERS):
        for p in parts: block(p)

print(f"per block: batched M=24768 {timed(batched):.0f} us | 3 x M=8256 on one stream {timed(split_seq):.0f} us | on three streams {timed(split):.0f} us")
