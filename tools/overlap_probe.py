#!/usr/bin/env python3
"""Feasibility probe: does the MFMA-bound grouped weight-gradient launch overlap with the HBM-bound kernels of the backward
chain (LayerNorm backward, attention backward) when the two run on separate HIP streams -- plain streams, and streams with
disjoint CU masks (hipExtStreamCreateWithCUMask)?  Prints serial vs concurrent time of [wgrad] || [2 x LN bwd + attn bwd + 3 dgrad GEMMs]."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
hip = ctypes.CDLL("libamdhip64.so")
S, L, H, D, F = 192, 129, 12, 768, 3072
M = S * L; Mp = ops.pad_rows(M)
g = torch.Generator().manual_seed(0)
bf = lambda r, c, s=0.1: (torch.randn(r, c, generator=g) * s).bfloat16().to(dev)
# wgrad operands
pairs = [(bf(Mp, 3 * D), bf(Mp, D, 1.0)), (bf(Mp, D), bf(Mp, D, 1.0)), (bf(Mp, F), bf(Mp, D, 1.0)), (bf(Mp, D), bf(Mp, F, 1.0))]
outs = [torch.zeros(p.shape[1], q.shape[1], device=dev) for p, q in pairs]
jobs = [(p, q, o) for (p, q), o in zip(pairs, outs)]
# chain operands
x = torch.randn(Mp, D, device=dev); gamma = torch.ones(D, device=dev); mean = torch.zeros(Mp, device=dev); rstd = torch.ones(Mp, device=dev)
dy = bf(Mp, D); dres = torch.randn(Mp, D, device=dev); dxf = torch.zeros(Mp, D, device=dev); dxb = torch.zeros(Mp, D, device=dev, dtype=torch.bfloat16)
dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
qkv = bf(Mp, 3 * D, 1.0); o = torch.zeros(Mp, D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(S, H, L, device=dev)
ops.attn_fwd(qkv, o, lse, S, L, H)
do = bf(Mp, D); dqkv = torch.zeros_like(qkv)
w_in_t = bf(D, 3 * D, 0.02); w_o_t = bf(D, D, 0.02); w_fc_t = bf(D, F, 0.02)
du = bf(Mp, F); dh = torch.zeros(Mp, D, device=dev, dtype=torch.bfloat16)

def chain():
    ops.gemm_nt(du, w_fc_t, M, ops.BF16, dh)                      # dh2
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, M, dres=dres, dx_f32=dxf, dx_bf16=dxb, dgamma=dgam, dbeta=dbet)
    ops.gemm_nt(dxb, w_o_t, M, ops.BF16, dh)                      # d attn
    ops.attn_bwd(qkv, o, do, lse, dqkv, S, L, H)
    ops.gemm_nt(dqkv, w_in_t, M, ops.BF16, dh)                    # dh1
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, M, dres=dres, dx_f32=dxf, dx_bf16=dxb, dgamma=dgam, dbeta=dbet)

def wgrad():
    ops.gemm_tn_grouped(jobs)

def masked_stream(pred):
    words = (ctypes.c_uint32 * 8)()
    for bit in range(256):
        if pred(bit):
            words[bit // 32] |= 1 << (bit % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

t_chain, t_w = timeit(chain), timeit(wgrad)
print(f"serial: chain {t_chain:.0f} us + wgrad {t_w:.0f} us = {t_chain + t_w:.0f} us")

def concurrent(sa, sb, reserve):
    cur = torch.cuda.current_stream()
    def run():
        sa.wait_stream(cur); sb.wait_stream(cur)
        prev = lib.sig_tune_reserved_cus(reserve[0])
        with torch.cuda.stream(sa):
            chain()
        lib.sig_tune_reserved_cus(reserve[1])
        with torch.cuda.stream(sb):
            wgrad()
        lib.sig_tune_reserved_cus(prev)
        cur.wait_stream(sa); cur.wait_stream(sb)
    return timeit(run)

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print(f"two plain streams: {concurrent(sa, sb, (0, 0)):.0f} us")
for ncu_w in (64, 96, 112, 128):
    per_xcd = ncu_w // 8
    # mask bit b -> XCD b % 8, CU b // 8 (assumed); wgrad takes the first per_xcd CUs of every XCD, the chain the rest
    sw = masked_stream(lambda b: (b // 8) < per_xcd)
    sc = masked_stream(lambda b: (b // 8) >= per_xcd)
    t = concurrent(sc, sw, (ncu_w, 256 - ncu_w))
    with torch.cuda.stream(sw):
        lib.sig_tune_reserved_cus(256 - ncu_w); tw = timeit(wgrad); lib.sig_tune_reserved_cus(0)
    with torch.cuda.stream(sc):
        lib.sig_tune_reserved_cus(ncu_w); tc = timeit(chain); lib.sig_tune_reserved_cus(0)
    print(f"CU masks, wgrad on {ncu_w} CUs / chain on {256 - ncu_w}: concurrent {t:.0f} us  (alone: wgrad {tw:.0f} us, chain {tc:.0f} us)")
