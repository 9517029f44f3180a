#!/usr/bin/env python3
"""Diagnostic: builds a PRIVATE copy of the library with -DSIG_GEMM_STAMPS (s_memtime stamps at kernel start, first data
landed, main loop end, stores drained) and prints the per-tile time split of the qkv / c_fc / c_proj GEMMs. GPU box only;
the shipped library never contains the stamps."""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "signal_amd", "csrc")
tmp = tempfile.mkdtemp()
lib = os.path.join(tmp, "libsignal_hip_stamps.so")
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip") and f != "sim.hip"]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffast-math",
       "-fno-finite-math-only", "-DSIG_GEMM_STAMPS", "-shared", "-o", lib, *srcs, os.path.join(csrc, "sim.hip")]
subprocess.run(cmd, check=True, capture_output=True)
import torch
from signal_amd import _lib, ops
_lib.LIB_PATH = lib
_lib._lib = None
L = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get('STAMP_M', '24768')); Mp = ops.pad_rows(M)   # STAMP_M=512: a few tiles only -> a tile's cost with the chip idle
for name, n, k, epi in [("qkv", 2304, 768, ops.BIAS_BF16), ("c_fc", 3072, 768, ops.BIAS_GELU_BF16), ("c_fc_inf", 3072, 768, ops.BIAS_GELU_BF16),
                        ("dgelu", 3072, 768, ops.DGELU_BF16), ("c_proj", 768, 3072, ops.BIAS_RES_F32)]:
    a = torch.randn(Mp, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) * .02).to(torch.bfloat16)
    bias = torch.randn(n, device=dev); f32 = epi == ops.BIAS_RES_F32
    out = torch.zeros(Mp, n, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    aux = torch.randn(Mp, n, device=dev).to(torch.bfloat16) if (epi == ops.BIAS_GELU_BF16 and name != "c_fc_inf") or epi == ops.DGELU_BF16 else None
    for _ in range(3):
        ops.gemm_nt(a, w, M, epi, out, bias=None if epi == ops.DGELU_BF16 else bias, res=out if f32 else None, aux=aux)
    torch.cuda.synchronize()
    T = int(os.environ.get('SIG_GEMM_TILE', '128'))
    nb = min(8192, (Mp // T) * (n // T))
    big = (ctypes.c_ulonglong * (5 * 8192))()
    assert L.sig_debug_read_stamps(big, 0) == 0
    allv = torch.tensor(list(big), dtype=torch.float64)
    t = allv[:4 * nb].view(nb, 4)
    issued = allv[4 * 8192:4 * 8192 + nb]
    ok = t[:, 0] > 0
    issued = issued[ok]
    t = t[ok]
    extra = f"  [stores issued after {(issued - t[:, 2]).median():.0f}, acknowledged {(t[:, 3] - issued).median():.0f} later]" if float(issued.max()) > 0 else ""
    pro, loop, epi_t = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
    span = (t[:, 3].max() - t[:, 0].min())
    print(f"{name:7s} tiles {nb:5d}  per-tile cycles: prologue {pro.median():8.0f}  main loop {loop.median():8.0f} ({loop.median()/(k//64):6.0f}/k-step)"
          f"  epilogue+drain {epi_t.median():8.0f}  total {(t[:,3]-t[:,0]).median():8.0f}" + extra)
shutil.rmtree(tmp, ignore_errors=True)
