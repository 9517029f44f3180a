#!/usr/bin/env python3
"""Diagnostic (GPU box): per-CU timeline of attn_bwd_kernel from in-kernel timestamps (private -DSIG_ATTN_STAMPS -DSIG_ATTN_HWID
build): how many blocks a CU holds at a time, and how the staging (HBM) phases of co-resident blocks line up."""
import ctypes, os, shutil, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "signal_amd", "csrc")
tmp = tempfile.mkdtemp(); lib = os.path.join(tmp, "libsignal_hip_astamps.so")
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffast-math",
                "-fno-finite-math-only", "-DSIG_ATTN_STAMPS", "-DSIG_ATTN_HWID", "-I" + os.path.join(ROOT, "include"), "-shared", "-o", lib, *srcs, "-ldl"],
               check=True, capture_output=True)
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
for _ in range(3): ops.attn_bwd(qkv, o, do, lse, dqkv, S, L, H)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * nb))()
assert L_.sig_debug_read_attn_stamps(buf, nb) == 0
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(nb)]
t0 = min(r[0] for r in rows if r[0])
percu = collections.defaultdict(list)
for i, (a, b, hw, d) in enumerate(rows):
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    cu, sh, se, tg = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 0x7, (hwid >> 16) & 0xf
    percu[(xcc, se, sh, cu)].append((a - t0, b - t0, d - t0, tg, i))
span = max(r[3] for r in rows) - t0
print(f"blocks {nb}, distinct CUs {len(percu)}, kernel span {span} ticks; blocks per CU min/median/max "
      f"{min(map(len, percu.values()))}/{sorted(map(len, percu.values()))[len(percu)//2]}/{max(map(len, percu.values()))}")
tick = (rows[0][3] - rows[0][0])
# concurrency per CU: sweep events
both_stage = one_stage = tot2 = tot1 = tot0 = 0
for key, lst in percu.items():
    ev = []
    for a, b, d, tg, i in lst:
        ev += [(a, 0, +1), (b, 0, -1), (a, 1, +1), (d, 1, -1)]     # kind 0: staging interval, kind 1: whole block
    ev.sort()
    st = res = 0; last = 0
    for t, kind, dlt in ev:
        dt = t - last
        if res == 2: tot2 += dt
        elif res == 1: tot1 += dt
        else: tot0 += dt
        if st == 2: both_stage += dt
        elif st == 1: one_stage += dt
        last = t
        if kind == 0: st += dlt
        else: res += dlt
tot = tot0 + tot1 + tot2
print(f"CU time with 2 / 1 / 0 resident blocks: {tot2 / tot:.2%} / {tot1 / tot:.2%} / {tot0 / tot:.2%} (of the per-CU span up to its last block)")
print(f"CU time with both resident blocks staging {both_stage / tot:.2%}, exactly one staging {one_stage / tot:.2%}")
k = sorted(percu)[0]
print("one CU's blocks (start, staging end, end, TG_ID, blockIdx):")
for r in sorted(percu[k]): print("  ", r)
shutil.rmtree(tmp, ignore_errors=True)
