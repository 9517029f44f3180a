"""Tensor-level wrappers over the C ABI: shape/dtype validation happens here, before any launch
(SURVEY.md section 8(b) 'Errors'); kernels run on PyTorch's CURRENT stream and never synchronise."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

# epilogue ids of sig_gemm_nt (include/signal_hip.h)
F32, BF16, BIAS_F32, BIAS_BF16, BIAS_RES_F32, BIAS_GELU_BF16, DGELU_BF16, BIAS_GELUERF_BF16, DGELUERF_BF16, RES_F32 = range(10)

ROW_PAD = 128


def pad_rows(m: int) -> int:
    return (m + ROW_PAD - 1) // ROW_PAD * ROW_PAD


def zeros_rows(m: int, n: int, dtype, device) -> torch.Tensor:
    """[pad_rows(m), n] zero buffer; kernels write only the first m rows, so the pad stays zero."""
    return torch.zeros(pad_rows(m), n, dtype=dtype, device=device)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


OP16 = (torch.bfloat16, torch.float16)   # the two MFMA operand types; the C ABI's dtype code is the index
DT_BF16, DT_F16 = 0, 1


def dt_code(dtype) -> int:
    """torch dtype -> SIG_DT_* of include/signal_hip.h"""
    if dtype not in OP16:
        raise TypeError(f"expected a 16-bit operand type (bfloat16 or float16), got {dtype}")
    return OP16.index(dtype)


def _chk16(t: torch.Tensor, name: str, like=None, ndim: int = 2) -> int:
    """16-bit operand check; with `like` the type must match that tensor's (one operand type per call)."""
    want = like.dtype if like is not None else (t.dtype if t.dtype in OP16 else torch.bfloat16)
    _chk(t, want, name, ndim)
    return dt_code(t.dtype)


def _chk(t: torch.Tensor, dtype, name: str, ndim: int = 2):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a device tensor (signal_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def gemm_nt(a: torch.Tensor, bt: torch.Tensor, m: int, epilogue: int, out: torch.Tensor,
            bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
            aux: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[:m] = a[:m] @ bt.T (+ epilogue).  a: bf16 [>=pad_rows(m), K]; bt: bf16 [N, K]."""
    dt = _chk16(a, "gemm_nt.a")
    _chk16(bt, "gemm_nt.bt", like=a)
    n, k = bt.shape
    if a.shape[1] != k:
        raise ValueError(f"gemm_nt: K mismatch {a.shape[1]} vs {k}")
    if a.shape[0] < pad_rows(m):
        raise ValueError(f"gemm_nt: a has {a.shape[0]} rows, needs {pad_rows(m)} (row padding)")
    want = torch.float32 if epilogue in (F32, BIAS_F32, BIAS_RES_F32, RES_F32) else a.dtype
    _chk(out, want, "gemm_nt.out")
    if out.shape[0] < m or out.shape[1] != n:
        raise ValueError(f"gemm_nt: out shape {tuple(out.shape)} does not hold [{m},{n}]")
    if bias is not None:
        _chk(bias, torch.float32, "gemm_nt.bias", 1)
        if bias.numel() != n:
            raise ValueError("gemm_nt: bias length")
    if res is not None:
        _chk(res, torch.float32, "gemm_nt.res")
        if res.shape[0] < m or res.shape[1] != n:
            raise ValueError("gemm_nt: residual shape")
    if aux is not None:
        _chk16(aux, "gemm_nt.aux", like=a)
        if aux.shape[0] < m or aux.shape[1] != n:
            raise ValueError("gemm_nt: aux shape")
    _lib.call("sig_gemm_nt", a.data_ptr(), a.stride(0), bt.data_ptr(), bt.stride(0), m, n, k, epilogue,
              out.data_ptr(), out.stride(0), _ptr(bias), _ptr(res), 0 if res is None else res.stride(0),
              _ptr(aux), 0 if aux is None else aux.stride(0), dt, _stream())
    return out


def gemm_tn(p: torch.Tensor, q: torch.Tensor, out: torch.Tensor, split: int = 0) -> torch.Tensor:
    """out += p.T @ q, f32 accumulate (atomics).  p: bf16 [Mr, I], q: bf16 [Mr, J], Mr % 64 == 0, zero pad rows."""
    dt = _chk16(p, "gemm_tn.p")
    _chk16(q, "gemm_tn.q", like=p)
    _chk(out, torch.float32, "gemm_tn.out")
    if p.shape[0] != q.shape[0]:
        raise ValueError("gemm_tn: row mismatch")
    if tuple(out.shape) != (p.shape[1], q.shape[1]):
        raise ValueError(f"gemm_tn: out {tuple(out.shape)} vs [{p.shape[1]},{q.shape[1]}]")
    _lib.call("sig_gemm_tn", p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0), p.shape[0], p.shape[1], q.shape[1],
              out.data_ptr(), out.stride(0), split, dt, _stream())
    return out


def gemm_tn_grouped(jobs) -> None:
    """out_k += p_k.T @ q_k for up to 4 (p, q, out[, colsum]) jobs with the same row count, as one launch (a transformer
    block's four weight gradients); colsum (optional, f32 [I]) += column sums of p (the bias gradient)."""
    import ctypes
    if not 1 <= len(jobs) <= 4:
        raise ValueError("gemm_tn_grouped: 1..4 jobs")
    arr = (_lib.SigTnJobDesc * len(jobs))()
    dt = rows = None
    for k, job in enumerate(jobs):
        p, q, out = job[:3]
        cs = job[3] if len(job) > 3 else None          # optional [I] f32: += column sums of p
        d = _chk16(p, "gemm_tn_grouped.p")
        _chk16(q, "gemm_tn_grouped.q", like=p)
        _chk(out, torch.float32, "gemm_tn_grouped.out")
        if dt is None:
            dt, rows = d, p.shape[0]
        if d != dt or p.shape[0] != rows or q.shape[0] != rows:
            raise ValueError("gemm_tn_grouped: the jobs must share operand type and row count")
        if tuple(out.shape) != (p.shape[1], q.shape[1]):
            raise ValueError(f"gemm_tn_grouped: out {tuple(out.shape)} vs [{p.shape[1]},{q.shape[1]}]")
        if cs is not None:
            _chk(cs, torch.float32, "gemm_tn_grouped.colsum", ndim=1)
            if cs.numel() != p.shape[1]:
                raise ValueError("gemm_tn_grouped: colsum must have one entry per column of p")
        arr[k] = _lib.SigTnJobDesc(p.data_ptr(), q.data_ptr(), out.data_ptr(), p.stride(0), q.stride(0), out.stride(0),
                                   p.shape[1], q.shape[1], _ptr(cs))
    _lib.call("sig_gemm_tn_grouped", ctypes.cast(arr, ctypes.c_void_p), len(jobs), rows, dt, _stream())


def layernorm_fwd(x, gamma, beta, m, y_bf16=None, y_f32=None, mean=None, rstd=None, eps=1e-5):
    _chk(x, torch.float32, "layernorm_fwd.x")
    d = x.shape[1]
    for t, nm in ((gamma, "gamma"), (beta, "beta")):
        _chk(t, torch.float32, "layernorm_fwd." + nm, 1)
    if not x.is_contiguous():
        raise ValueError("layernorm_fwd: x must be contiguous")
    for t in (y_bf16, y_f32):
        if t is not None and (not t.is_contiguous() or t.shape[1] != d or t.shape[0] < m):
            raise ValueError("layernorm_fwd: output shape")
    _lib.call("sig_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(y_bf16), _ptr(y_f32),
              _ptr(mean), _ptr(rstd), m, d, float(eps), 0 if y_bf16 is None else dt_code(y_bf16.dtype), _stream())


def layernorm_bwd(dy, x, gamma, mean, rstd, m, dres=None, dx_f32=None, dx_bf16=None, dgamma=None, dbeta=None):
    if dy.dtype not in OP16 + (torch.float32,):
        raise TypeError("layernorm_bwd: dy must be bf16 / f16 or f32")
    t16 = [t for t in (dy, dx_bf16) if t is not None and t.dtype in OP16]
    if len({t.dtype for t in t16}) > 1:
        raise TypeError("layernorm_bwd: dy and dx_bf16 must share one 16-bit type")
    dt = dt_code(t16[0].dtype) if t16 else 0
    _chk(x, torch.float32, "layernorm_bwd.x")
    d = x.shape[1]
    for t in (dy, x, dres, dx_f32, dx_bf16):
        if t is not None and (not t.is_contiguous() or t.shape[1] != d or t.shape[0] < m):
            raise ValueError("layernorm_bwd: tensor shape")
    _lib.call("sig_layernorm_bwd", dy.data_ptr(), int(dy.dtype in OP16), x.data_ptr(), gamma.data_ptr(),
              mean.data_ptr(), rstd.data_ptr(), _ptr(dres), _ptr(dx_f32), _ptr(dx_bf16), _ptr(dgamma), _ptr(dbeta),
              m, d, dt, _stream())


def attn_fwd(qkv, out, lse, s, l, h):
    dt = _chk16(qkv, "attn_fwd.qkv")
    _chk16(out, "attn_fwd.out", like=qkv)
    if qkv.shape[1] != 3 * h * 64 or out.shape[1] != h * 64 or qkv.shape[0] < s * l or out.shape[0] < s * l:
        raise ValueError("attn_fwd: shapes")
    if not (qkv.is_contiguous() and out.is_contiguous()):
        raise ValueError("attn_fwd: tensors must be contiguous")
    if lse is not None and (lse.dtype != torch.float32 or lse.numel() != s * h * l):
        raise ValueError("attn_fwd: lse must be f32 [S,H,L]")
    _lib.call("sig_attn_fwd", qkv.data_ptr(), out.data_ptr(), _ptr(lse), s, l, h, dt, _stream())


def attn_bwd(qkv, out, dout, lse, dqkv, s, l, h):
    for t, nm, w in ((qkv, "qkv", 3 * h * 64), (out, "out", h * 64), (dout, "dout", h * 64), (dqkv, "dqkv", 3 * h * 64)):
        dt = _chk16(t, "attn_bwd." + nm, like=qkv)
        if t.shape[1] != w or t.shape[0] < s * l or not t.is_contiguous():
            raise ValueError(f"attn_bwd: {nm} shape")
    if lse.dtype != torch.float32 or lse.numel() != s * h * l:
        raise ValueError("attn_bwd: lse must be f32 [S,H,L]")
    _lib.call("sig_attn_bwd", qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), s, l, h,
              dt, _stream())


def cast_bf16(src: torch.Tensor, dst: torch.Tensor):
    if src.dtype != torch.float32 or dst.dtype not in OP16 or src.numel() != dst.numel():
        raise ValueError("cast_bf16: f32 -> bf16 / f16 of equal size")
    if not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("cast_bf16: contiguous tensors only")
    _lib.call("sig_cast_bf16", src.data_ptr(), dst.data_ptr(), src.numel(), dt_code(dst.dtype), _stream())
    return dst


def transpose_cast_bf16(src: torch.Tensor, dst: torch.Tensor):
    _chk(src, torch.float32, "transpose_cast.src")
    dt = _chk16(dst, "transpose_cast.dst")
    r, c = src.shape
    if tuple(dst.shape) != (c, r) or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("transpose_cast: dst must be contiguous [cols, rows]")
    _lib.call("sig_transpose_cast_bf16", src.data_ptr(), dst.data_ptr(), r, c, dt, _stream())
    return dst


def colsum(a: torch.Tensor, m: int, out: torch.Tensor):
    """out[n] += sum_{r<m} a[r, n]"""
    _chk(out, torch.float32, "colsum.out", 1)
    if a.dim() != 2 or out.numel() != a.shape[1] or a.shape[0] < m:
        raise ValueError("colsum: shapes")
    if a.dtype in OP16:
        _lib.call("sig_colsum_bf16", a.data_ptr(), a.stride(0), m, a.shape[1], out.data_ptr(), dt_code(a.dtype), _stream())
    elif a.dtype == torch.float32:
        _lib.call("sig_colsum_f32", a.data_ptr(), a.stride(0), m, a.shape[1], out.data_ptr(), _stream())
    else:
        raise TypeError("colsum: bf16 / f16 or f32")
    return out


def im2col(img: torch.Tensor, out: torch.Tensor, patch: int):
    _chk(img, torch.float32, "im2col.img", 4)
    n, c, h, w = img.shape
    if c != 3 or not img.is_contiguous():
        raise ValueError("im2col: contiguous [N,3,H,W] expected")
    rows = n * (h // patch) * (w // patch)
    dt = _chk16(out, "im2col.out")
    if out.shape[0] < rows or out.shape[1] != 3 * patch * patch or not out.is_contiguous():
        raise ValueError("im2col: out shape")
    _lib.call("sig_im2col", img.data_ptr(), out.data_ptr(), n, h, w, patch, dt, _stream())
    return out


def embed_assemble(tok, cls_emb, pos, cv_embed, cam, sie_coe, ln_w, ln_b, x, pre_ln, mean, rstd, s, b, l, d, eps=1e-5):
    if cam is not None and cam.dtype != torch.int64:
        raise TypeError("embed_assemble: cam labels must be int64")
    _lib.call("sig_embed_assemble", tok.data_ptr(), cls_emb.data_ptr(), pos.data_ptr(), _ptr(cv_embed), _ptr(cam),
              float(sie_coe), ln_w.data_ptr(), ln_b.data_ptr(), x.data_ptr(), _ptr(pre_ln), _ptr(mean), _ptr(rstd),
              s, b, l, d, float(eps), _stream())


def embed_bwd(d_pre, dtok_f32, dtok_bf16, dcls, dpos, dcv, cam, sie_coe, s, b, l, d):
    _lib.call("sig_embed_assemble_bwd", d_pre.data_ptr(), _ptr(dtok_f32), _ptr(dtok_bf16), dcls.data_ptr(), dpos.data_ptr(),
              _ptr(dcv), _ptr(cam), float(sie_coe), s, b, l, d, 0 if dtok_bf16 is None else dt_code(dtok_bf16.dtype), _stream())
