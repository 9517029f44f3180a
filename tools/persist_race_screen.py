#!/usr/bin/env python3
"""Race screen for the persistent NT kernel (hand-counted s_waitcnt, parked store units): fresh operands every round, the persistent
result must equal the per-tile kernels' bit for bit, alone AND while a second stream saturates HBM with copies (which stretches every
DMA / store latency the counted waits cover).  GPU box only.  usage: tools/persist_race_screen.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signal_amd import ops, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
big_b = torch.empty_like(big_a)
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for (m, n, k) in [(24768, 2304, 768), (24768, 3072, 768), (8256, 3072, 1536)]:
        mp = ops.pad_rows(m)
        for r in range(rounds):
            g = torch.Generator(device=dev).manual_seed(1000 * r + n + k)
            a = torch.zeros(mp, k, device=dev, dtype=dt); a[:m] = torch.randn(m, k, device=dev, generator=g).to(dt)
            w = (torch.randn(n, k, device=dev, generator=g) * 0.05).to(dt)
            bias = torch.randn(n, device=dev, generator=g)
            aux = torch.zeros(mp, n, device=dev, dtype=dt); aux[:m] = torch.randn(m, n, device=dev, generator=g).to(dt)
            epis = [(ops.BIAS_BF16, dict(bias=bias)), (ops.BF16, dict(bias=None))]
            if k == 768:
                epis.append((ops.DGELU_BF16, dict(bias=None, aux=aux)))
            for epi, kw in epis:
                prev = lib.sig_tune_nt_persist(0)
                ref = torch.zeros(mp, n, device=dev, dtype=dt)
                ops.gemm_nt(a, w, m, epi, ref, **kw)
                lib.sig_tune_nt_persist(1)
                for load in (False, True):
                    if load:
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            for _ in range(6): big_b.copy_(big_a); big_a.copy_(big_b)
                    out = torch.zeros(mp, n, device=dev, dtype=dt)
                    ops.gemm_nt(a, w, m, epi, out, **kw)
                    ops.gemm_nt(a, w, m, epi, out, **kw)      # (second launch over a warm L2)
                    torch.cuda.synchronize()
                    if not torch.equal(out, ref):
                        bad += 1
                        d = (out.float() - ref.float()).abs()
                        print(f"MISMATCH dt={dt} {m}x{n}x{k} epi={epi} round={r} load={load}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}", flush=True)
                lib.sig_tune_nt_persist(prev)
        print(f"{dt} {m}x{n}x{k}: {rounds} rounds x (alone, under HBM load) done, mismatches so far {bad}", flush=True)
print("RACE SCREEN", "FAILED" if bad else "clean", f"({bad} mismatching launches)")
sys.exit(1 if bad else 0)
