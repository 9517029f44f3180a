#!/usr/bin/env python3
"""Builds signal_amd/lib/libsignal_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python signal_amd/csrc/build.py [--force] [--verbose]

Each .hip translation unit is compiled to an object (skipped when newer than its sources), then linked.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libsignal_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-ffast-math",
         "-fno-finite-math-only", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]


# translation units whose float results decide INTEGER outputs (top-k masks) are built without fast-math
PRECISE = {"sim.hip"}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _newest_header():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "signal_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdr = _newest_header()
    jobs, objs = [], []
    for src in sources():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            flags = [f for f in FLAGS if not (src in PRECISE and f in ("-ffast-math", "-fno-finite-math-only"))]
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
