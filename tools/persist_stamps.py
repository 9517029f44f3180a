#!/usr/bin/env python3
"""Diagnostic: builds a PRIVATE copy of the library with -DSIG_GEMM_STAMPS and prints where a workgroup of the persistent
192x256 NT kernel spends its cycles (prologue, each tile's twelve K-steps, each conversion, the final flush). GPU box only; the
shipped library never contains the stamps."""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "signal_amd", "csrc")
tmp = tempfile.mkdtemp()
lib = os.path.join(tmp, "libsignal_hip_stamps.so")
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip") and f != "sim.hip"]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffast-math",
       "-fno-finite-math-only", "-DSIG_GEMM_STAMPS", "-shared", "-o", lib, *srcs, os.path.join(csrc, "sim.hip"), "-ldl"]
subprocess.run(cmd, check=True, capture_output=True)
import torch
from signal_amd import _lib, ops
_lib.LIB_PATH = lib
_lib._lib = None
L = _lib.load()
L.sig_tune_nt_persist(2)
dev = torch.device("cuda:0")
M = 24768; Mp = ops.pad_rows(M)
for name, n, k, epi in [("qkv", 2304, 768, ops.BIAS_BF16), ("c_fc_infer", 3072, 768, ops.BIAS_GELU_BF16), ("dgelu", 3072, 768, ops.DGELU_BF16)]:
    a = torch.randn(Mp, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) * .02).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    out = torch.zeros(Mp, n, device=dev, dtype=torch.bfloat16)
    aux = torch.randn(Mp, n, device=dev).to(torch.bfloat16) if epi == ops.DGELU_BF16 else None
    for _ in range(3):
        ops.gemm_nt(a, w, M, epi, out, bias=None if epi == ops.DGELU_BF16 else bias, aux=aux)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 17))()
    assert L.sig_debug_read_pstamps(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(256, 17)
    ntl = ((t[:, 2:16:2] > 0).sum(1)).long()                 # tiles per workgroup
    pro = (t[:, 1] - t[:, 0]).median()
    tile = []; conv = []
    for b in range(256):
        prev = t[b, 1]
        for i in range(int(ntl[b])):
            tile.append(float(t[b, 2 + 2 * i] - prev)); conv.append(float(t[b, 3 + 2 * i] - t[b, 2 + 2 * i])); prev = t[b, 3 + 2 * i]
    tile, conv = torch.tensor(tile), torch.tensor(conv)
    last = torch.stack([t[b, 3 + 2 * (int(ntl[b]) - 1)] for b in range(256)])
    flush = (t[:, 16] - last).median()
    span = float(t[:, 16].max() - t[:, 0].min())
    print(f"{name:11s} tiles/WG {int(ntl.min())}-{int(ntl.max())}  prologue {pro:7.0f}  tile main loop median {tile.median():7.0f} = {tile.median() / (k // 64):6.0f} per K-step "
          f"(MFMA issue 1536)  conversion {conv.median():6.0f}  final flush {flush:6.0f}  whole launch {span:8.0f} cycles")
shutil.rmtree(tmp, ignore_errors=True)
