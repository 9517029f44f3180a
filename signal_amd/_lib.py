"""ctypes binding of libsignal_hip.so (include/signal_hip.h).  There is no fallback: if the shared
library is missing or a call fails, the product path raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIGNAL_HIP_LIB: developer override for same-box A/B timing of two builds (tools/); the product always ships lib/libsignal_hip.so
LIB_PATH = os.environ.get("SIGNAL_HIP_LIB") or os.path.join(_HERE, "lib", "libsignal_hip.so")

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

_TUNING_ONLY = {"sig_tune_gemm_tile", "sig_tune_nt_persist", "sig_tune_ln_defer", "sig_ln_flush", "sig_tune_tn_overwrite", "sig_zero_ranges", "sig_tune_attn_bwd_waves", "sig_tune_attn_fwd_waves", "sig_tune_reserved_cus", "sig_tune_tn_path", "sig_debug_tn_plan"}

# name -> argtypes; mirrors include/signal_hip.h one to one (tests check the export list against the header)
SIGNATURES = {
    "sig_prof_begin": [_i, _i, _i, _i],
    "sig_prof_end": [_vp, _vp, _vp],
    "sig_tune_gemm_tile": [_i],
    "sig_tune_nt_persist": [_i],
    "sig_tune_ln_defer": [_i],
    "sig_tune_tn_overwrite": [_i],
    "sig_tune_attn_bwd_waves": [_i],
    "sig_tune_attn_fwd_waves": [_i],
    "sig_ln_flush": [_vp],
    "sig_zero_ranges": [_vp, _vp, _vp, _i, _i, _vp],
    "sig_tune_reserved_cus": [_i],
    "sig_tune_tn_path": [_i],
    "sig_debug_tn_plan": [_i, _i, _i, _i, _vp],
    "sig_gemm_nt": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp],
    "sig_gemm_tn": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp],
    "sig_gemm_tn_grouped": [_vp, _i, _i, _i, _vp],
    "sig_comm_unique_id": [_vp],
    "sig_comm_init": [_vp, _i, _i, _vp],
    "sig_comm_allreduce_async": [_vp, _vp, _sz, _vp],
    "sig_comm_wait": [_vp, _vp],
    "sig_comm_destroy": [_vp],
    "sig_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "sig_layernorm_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sig_attn_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sig_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sig_cast_bf16": [_vp, _vp, _sz, _i, _vp],
    "sig_transpose_cast_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "sig_transpose_cast_multi": [_vp, _vp, _i, _i, _i, _vp],
    "sig_transpose16_multi": [_vp, _vp, _i, _i, _vp],
    "sig_colsum_bf16": [_vp, _i, _i, _i, _vp, _i, _vp],
    "sig_colsum_f32": [_vp, _i, _i, _i, _vp, _vp],
    "sig_im2col": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sig_embed_assemble": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "sig_embed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp],
}



# ---- struct mirrors of include/signal_hip.h -----------------------------------------------------------
def _struct(name, ptr_fields, tail=()):
    fields = [(f, _vp) for f in ptr_fields] + list(tail)
    return type(name, (C.Structure,), {"_fields_": fields})


class SigVitDims(C.Structure):
    _fields_ = [(n, _i) for n in ("S", "B", "L", "D", "H", "F", "out_dim", "dtype")]


class SigTnJobDesc(C.Structure):
    _fields_ = [("P", _vp), ("Q", _vp), ("out", _vp)] + [(n, _i) for n in ("ldp", "ldq", "ldo", "I", "J")] + [("colsum", _vp)]


SigEmbedParams = _struct("SigEmbedParams", ["w_conv", "class_embedding", "positional_embedding", "cv_embed", "ln_w", "ln_b"],
                         [("sie_coe", _f)])
SigEmbedActs = _struct("SigEmbedActs", ["patches", "tok", "pre_ln", "mean", "rstd", "x0"])
SigEmbedGrads = _struct("SigEmbedGrads", ["w_conv", "class_embedding", "positional_embedding", "cv_embed", "ln_w", "ln_b"])
SigBlockParams = _struct("SigBlockParams", ["w_in", "w_out", "w_fc", "w_proj", "wt_in", "wt_out", "wt_fc", "wt_proj",
                                             "b_in", "b_out", "b_fc", "b_proj", "ln1_w", "ln1_b", "ln2_w", "ln2_b"])
SigBlockActs = _struct("SigBlockActs", ["x_in", "h1", "mean1", "rstd1", "qkv", "lse", "attn", "x_mid", "h2", "mean2", "rstd2",
                                         "u", "g", "x_out"])
SigBlockGrads = _struct("SigBlockGrads", ["w_in", "w_out", "w_fc", "w_proj", "b_in", "b_out", "b_fc", "b_proj",
                                           "ln1_w", "ln1_b", "ln2_w", "ln2_b"])
SigBlockScratch = _struct("SigBlockScratch", ["du", "dh", "dqkv", "dx_mid", "dx_mid_b"])
SigHeadParams = _struct("SigHeadParams", ["proj_t", "proj", "ln_w", "ln_b"])
SigHeadActs = _struct("SigHeadActs", ["x", "hp", "mean", "rstd", "tokens"])
SigHeadGrads = _struct("SigHeadGrads", ["proj", "ln_w", "ln_b"])
SigSimParams = _struct("SigSimParams", ["sel_wq", "sel_bq", "sel_wk", "sel_bk", "w_q", "w_kv", "w_o", "w_f1", "w_f2",
                                         "wt_q", "wt_kv", "wt_o", "wt_f1", "wt_f2", "b_q", "b_kv", "b_o", "b_f1", "b_f2",
                                         "n1_w", "n1_b", "n2_w", "n2_b"], [("topk", _i), ("dtype", _i), ("max_keep", _i)])
SigSimActs = _struct("SigSimActs", ["qprime", "cconst", "intra", "inter", "mask_f", "mask_u8", "sel", "cls_b", "cls_f", "qh",
                                     "kv", "probs", "ao", "y", "z1", "z1_b", "mean1", "rstd1", "f1_pre", "f1", "y2", "mean2",
                                     "rstd2", "out"])
SigSimGrads = _struct("SigSimGrads", ["w_q", "w_kv", "w_o", "w_f1", "w_f2", "b_q", "b_kv", "b_o", "b_f1", "b_f2",
                                       "n1_w", "n1_b", "n2_w", "n2_b"])
SigSimScratch = _struct("SigSimScratch", ["dy2", "dy2_b", "df1", "dz1", "dy", "dy_b", "dao", "dqh", "dqh_b", "dkv", "dsel", "dcls"])

SigGamActs = _struct("SigGamActs", ["fh", "nrm", "lv", "la", "vec", "coef", "loss"])
SigDasParams = _struct("SigDasParams", ["w_q", "w_0", "wt_q", "wt_0", "b_q", "b_0", "wd", "bd", "w4"])
SigDasGrads = _struct("SigDasGrads", ["w_q", "w_0", "b_q", "b_0", "wd", "bd", "w4"])
SigLamActs = _struct("SigLamActs", ["xb", "q", "a1", "a1pre", "a2pre", "offs", "samp", "loss"])
SigLamScratch = _struct("SigLamScratch", ["da1pre", "dq", "dx"])

SIGNATURES.update({
    "sig_bnneck_fwd": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sig_bnneck_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sig_reid_loss": [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sig_adam_step": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _i, _f, _vp, _sz, _vp],
    "sig_grad_check": [_vp, _sz, _vp, _vp],
    "sig_loss_scale_update": [_vp, _f, _f, _i, _vp],
    "sig_gam_fwd": [_vp, _i, _i, _vp, _vp, _vp],
    "sig_gam_bwd": [_i, _i, _vp, _vp, _vp, _vp, _vp],
    "sig_lam_fwd": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "sig_lam_bwd": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sig_embed_assemble_bwd": SIGNATURES.pop("sig_embed_bwd"),
    "sig_embed_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "sig_embed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sig_block_fwd": [_vp, _vp, _vp, _vp],
    "sig_block_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sig_head_fwd": [_vp, _vp, _vp, _vp],
    "sig_head_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sig_sim_select": [_vp, _i, _i, _vp, _vp, _vp],
    "sig_sim_fwd": [_vp, _i, _i, _vp, _vp, _vp],
    "sig_sim_bwd": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sig_xattn_fwd": [_vp, _vp, _i, _i, _vp, _vp, _i, _vp],
    "sig_xattn_bwd": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp],
})


def ref(struct):
    """pointer argument for a struct"""
    return C.cast(C.pointer(struct), _vp)


def fill(struct_cls, **tensors):
    """Build a struct from name -> tensor / None / number; unknown names are an error."""
    s = struct_cls()
    names = {f[0] for f in struct_cls._fields_}
    for k, v in tensors.items():
        if k not in names:
            raise KeyError(f"{struct_cls.__name__} has no field {k}")
        if v is None:
            setattr(s, k, None)
        elif hasattr(v, "data_ptr"):
            setattr(s, k, v.data_ptr())
        else:
            setattr(s, k, v)
    return s


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
        if name in _TUNING_ONLY and not hasattr(lib, name):
            continue          # an older build loaded through SIGNAL_HIP_LIB for an A/B run
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
