#!/usr/bin/env python3
"""Diagnostic (GPU box): time gemm_tn256_kernel with and without its atomic flush (private -DSIG_TN_NOFLUSH build)."""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "signal_amd", "csrc")
tmp = tempfile.mkdtemp(); lib = os.path.join(tmp, "libsignal_hip_noflush.so")
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffast-math",
                "-fno-finite-math-only", "-DSIG_TN_NOFLUSH", "-I" + os.path.join(ROOT, "include"), "-shared", "-o", lib, *srcs], check=True, capture_output=True)
import torch
from signal_amd import _lib, ops
dev = torch.device("cuda:0"); M = 24768; Mp = ops.pad_rows(M)
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
shapes = [(2304, 768), (3072, 768), (768, 3072)]
ten = {s: (torch.randn(Mp, s[0], device=dev).to(torch.bfloat16), torch.randn(Mp, s[1], device=dev).to(torch.bfloat16), torch.zeros(*s, device=dev)) for s in shapes}
full = {s: timeit(lambda: ops.gemm_tn(*ten[s])) for s in shapes}
_lib.LIB_PATH = lib; _lib._lib = None; _lib.load()
for s in shapes:
    nf = timeit(lambda: ops.gemm_tn(*ten[s]))
    print(f"wgrad {s[0]}x{s[1]}: with flush {full[s]:.0f} us, without {nf:.0f} us  (flush {s[0]*s[1]*4*0+0:.0f})")
shutil.rmtree(tmp, ignore_errors=True)
