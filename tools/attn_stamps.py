#!/usr/bin/env python3
"""Diagnostic (GPU box): private -DSIG_ATTN_STAMPS build; per-block cycle split of attn_bwd_kernel (wave 0's view):
staging (global -> LDS, delta), pass A (dQ), pass B (dK, dV) + store drain."""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "signal_amd", "csrc")
tmp = tempfile.mkdtemp(); lib = os.path.join(tmp, "libsignal_hip_astamps.so")
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffast-math",
                "-fno-finite-math-only", "-DSIG_ATTN_STAMPS=" + os.environ.get("STAMPS_VARIANT", "1"), "-I" + os.path.join(ROOT, "include"), "-shared", "-o", lib, *srcs], check=True, capture_output=True)
import torch
from signal_amd import _lib, ops
_lib.LIB_PATH = lib; _lib._lib = None; L_ = _lib.load()
dev = torch.device("cuda:0")
S, L, H = 192, 129, 12
Mp = ops.pad_rows(S * L)
qkv = torch.randn(Mp, 2304, device=dev).to(torch.bfloat16); o = torch.zeros(Mp, 768, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(S, H, L, device=dev); ops.attn_fwd(qkv, o, lse, S, L, H)
do = torch.randn(Mp, 768, device=dev).to(torch.bfloat16); dqkv = torch.zeros_like(qkv)
nb = S * H
def read():
    buf = (ctypes.c_ulonglong * (4 * nb))()
    assert L_.sig_debug_read_attn_stamps(buf, nb) == 0
    return torch.tensor(list(buf), dtype=torch.float64).view(nb, 4)
for _ in range(3): ops.attn_fwd(qkv, o, lse, S, L, H)
torch.cuda.synchronize()
t = read()
print(f"attn_fwd per block (cycles, median): staging {(t[:,1]-t[:,0]).median():.0f}  query tiles {(t[:,2]-t[:,1]).median():.0f}  store drain {(t[:,3]-t[:,2]).median():.0f}  total {(t[:,3]-t[:,0]).median():.0f}")
for _ in range(3): ops.attn_bwd(qkv, o, do, lse, dqkv, S, L, H)
torch.cuda.synchronize()
t = read()
st, pa, pb = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print(f"attn_bwd per block (cycles, median): staging {st.median():.0f}  pass A {pa.median():.0f}  pass B + drain {pb.median():.0f}  total {(t[:,3]-t[:,0]).median():.0f}")
shutil.rmtree(tmp, ignore_errors=True)
