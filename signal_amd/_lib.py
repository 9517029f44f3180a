"""ctypes binding of libsignal_hip.so (include/signal_hip.h).  There is no fallback: if the shared
library is missing or a call fails, the product path raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsignal_hip.so")

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> argtypes; mirrors include/signal_hip.h one to one (tests check the export list against the header)
SIGNATURES = {
    "sig_gemm_nt": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp],
    "sig_gemm_tn": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "sig_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "sig_layernorm_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "sig_attn_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "sig_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sig_cast_bf16": [_vp, _vp, _sz, _vp],
    "sig_transpose_cast_bf16": [_vp, _vp, _i, _i, _vp],
    "sig_colsum_bf16": [_vp, _i, _i, _i, _vp, _vp],
    "sig_colsum_f32": [_vp, _i, _i, _i, _vp, _vp],
    "sig_im2col": [_vp, _vp, _i, _i, _i, _i, _vp],
    "sig_embed_assemble": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "sig_embed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp],
}

_lib = None


class SignalHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SignalHipError(
            f"{LIB_PATH} is missing: build it with `python signal_amd/csrc/build.py` "
            "(or __graft_entry__.build()); signal_amd has no non-HIP fallback")
    lib = C.CDLL(LIB_PATH)
    lib.sig_last_error.restype = C.c_char_p
    lib.sig_last_error.argtypes = []
    lib.sig_version.restype = _i
    lib.sig_version.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    _lib = lib
    return lib


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SignalHipError(f"{name} failed (rc={rc}): {lib.sig_last_error().decode()}")
