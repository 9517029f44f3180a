"""Host side of the HIP path: owns the flat parameter / packed-weight storage and the activation
workspaces, and exposes the backbone (three modalities batched as one [3B,L,D] problem) and SIM as
torch.autograd.Functions whose forward/backward are sequences of C-ABI calls on the current stream.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every FLOP of these stages runs
in signal_amd/csrc.  There is no fallback: without a GPU or without the built library this raises."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import _lib
from .._lib import fill, ref
from ..ops import pad_rows

F32 = torch.float32
OPERAND_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16,
                  "float16": torch.float16, "half": torch.float16}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class FlatParams:
    """All parameters the HIP stages read live in ONE f32 buffer (views are re-pointed into it), their
    gradients in one f32 buffer of the same layout, and the 16-bit operands of the GEMMs (bf16 or f16, `op16`) in
    one buffer of that type: packing is one cast kernel over the whole buffer plus one transpose per matrix."""

    def __init__(self, named: Dict[str, torch.nn.Parameter], device, op_dtype=torch.bfloat16):
        self.names = list(named)
        self.params = [named[n] for n in self.names]
        self.offsets, off = {}, 0
        for n, p in zip(self.names, self.params):
            self.offsets[n] = off
            off += (p.numel() + 63) // 64 * 64          # keep every tensor 256-B aligned
        self.total = off
        self.device = device
        self.data = torch.zeros(off, dtype=F32, device=device)
        self.grad = torch.zeros(off, dtype=F32, device=device)
        self.op16 = torch.zeros(off, dtype=op_dtype, device=device)
        with torch.no_grad():
            for n, p in zip(self.names, self.params):
                v = self.view(self.data, n, p.shape)
                v.copy_(p.data)
                p.data = v
        self._versions = None
        self._ptrs = [p.data_ptr() for p in self.params]
        self.byname = dict(zip(self.names, self.params))

    def view(self, flat, name, shape=None):
        o = self.offsets[name]
        shape = self.byname[name].shape if shape is None else shape
        n = 1
        for s in shape:
            n *= s
        return flat[o:o + n].view(shape)

    def span(self, prefix):
        """[lo, hi) element range of the (contiguous) group of parameters whose name starts with prefix."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        if not idx:
            return None
        if idx != list(range(idx[0], idx[-1] + 1)):
            raise RuntimeError(f"parameters under {prefix} are not contiguous in the flat buffer")
        last = self.names[idx[-1]]
        hi = self.offsets[last] + (self.byname[last].numel() + 63) // 64 * 64
        return self.offsets[self.names[idx[0]]], hi

    def intact(self) -> bool:
        return all(p.data_ptr() == q for p, q in zip(self.params, self._ptrs))

    def stale(self) -> bool:
        v = tuple(p._version for p in self.params)
        if v != self._versions:
            self._versions = v
            return True
        return False


class _Ws:
    """Zero-initialised, row-padded buffers (pad rows are never written, so they stay zero)."""

    def __init__(self, device):
        self.device = device
        self.bytes = 0

    def z(self, rows, cols, dtype=F32):
        t = torch.zeros(pad_rows(rows), cols, dtype=dtype, device=self.device)
        self.bytes += t.numel() * t.element_size()
        return t

    def v(self, n, dtype=F32):
        t = torch.zeros(n, dtype=dtype, device=self.device)
        self.bytes += t.numel() * t.element_size()
        return t


class _WsLease:
    """Hands a workspace back to its pool exactly once: explicitly when the backward has consumed it, or when the autograd
    node holding the lease is dropped without a backward (a forward under grad mode that is never differentiated, an
    exception between forward and backward) -- such a call used to strand a multi-GB training workspace per occurrence."""
    __slots__ = ("ws",)

    def __init__(self, ws):
        self.ws = ws          # ws["_free"] is the pool's free list (no reference cycle: the workspace does not know the lease)

    def release(self):
        ws, self.ws = self.ws, None
        if ws is not None:
            ws["_free"].append(ws)

    __del__ = release


class HipPath:
    def __init__(self, model):
        self.model = model
        self.flat: Optional[FlatParams] = None
        self._vit_ws: Dict[tuple, list] = {}
        self._sim_ws: Dict[tuple, list] = {}
        self._gam_ws: Dict[tuple, list] = {}
        self._lam_ws: Dict[tuple, list] = {}
        enc = model.clip_vision_encoder
        b = enc.base
        self.D, self.H, self.layers, self.out_dim, self.patch = b.width, b.heads, b.layers, b.output_dim, b.patch
        self.F = 4 * self.D
        import os
        name = os.environ.get("SIGNAL_HIP_DTYPE") or getattr(model.cfg.MODEL, "OPERAND_DTYPE", "bf16")
        if str(name).lower() not in OPERAND_DTYPES:
            raise ValueError(f"MODEL.OPERAND_DTYPE={name!r}: expected 'bf16' or 'fp16'")
        # MFMA operand type of every GEMM / attention (f32 accumulate, residual stream, LayerNorm, softmax either way):
        # bf16, or fp16 = the reference's CUDA autocast type (engine/processor.py:165), trained with loss scaling
        self.operand_dtype = OPERAND_DTYPES[str(name).lower()]
        self.dt = 1 if self.operand_dtype == torch.float16 else 0
        self.L = b.h_resolution * b.w_resolution + 1
        self.img_hw = (b.h_resolution * b.patch, b.w_resolution * b.patch)

    # ------------------------------------------------------------------ parameters
    def _hip_named_params(self) -> Dict[str, torch.nn.Parameter]:
        # every parameter lives in the flat buffer (the fused optimizer and the gradient reducer walk it);
        # the HIP stages read the clip_vision_encoder. / SIM. / AlignM. groups
        return dict(self.model.named_parameters())

    def prepare(self, device):
        """(Re)build flat storage if needed and refresh the bf16 operands when a parameter changed."""
        if device.type != "cuda":
            raise _lib.SignalHipError("signal_amd runs on an MI355X only: move the model and inputs to 'cuda' "
                                      "(there is no CPU path; the CPU oracle lives in oracle/ and is test-only)")
        _lib.load()
        if self.flat is None or self.flat.device != device or not self.flat.intact():
            named = self._hip_named_params()
            if any(p.device != device for p in named.values()):
                raise _lib.SignalHipError("model parameters are not on the input's device: call model.to(device) first")
            self.flat = FlatParams(named, device, self.operand_dtype)
            self._build_structs()
            if self.direct_grads:
                self.enable_direct_grads()
        if self.flat.stale():
            self._pack()

    direct_grads = False
    # Grad mode of the CALLER.  Inside autograd.Function.forward grad mode is always off and ctx.needs_input_grad is
    # True for parameters even under torch.no_grad(), so the callers (make_model / align) record torch.is_grad_enabled()
    # here right before .apply(); without it an inference forward took (and never returned) a training workspace.
    grad_mode = True

    _grad_skip = None

    def enable_direct_grads(self, skip=None):
        """Training-engine mode: every Parameter's .grad is a persistent view of flat.grad; the HIP backward
        stages accumulate straight into it (no per-step clone), zeroing is one memset, the optimizer and the
        data-parallel reducer work on the flat buffer.  Parameters that can never receive a gradient (skip(name) true,
        or requires_grad False) keep .grad = None -- what autograd leaves them with in the reference -- so that torch
        optimizers skip them instead of applying weight decay to a zero gradient."""
        self.direct_grads = True
        if skip is not None:
            self._grad_skip = skip
        skip = self._grad_skip or (lambda n: False)
        for n, p in self.flat.byname.items():
            p.grad = None if (skip(n) or not p.requires_grad) else self.flat.view(self.flat.grad, n)

    # Training-engine mode on top of direct_grads: the transformer blocks' weight gradients (340 of the 345 MB of flat.grad) are
    # written exactly once per step, by the grouped weight-gradient launch of sig_block_bwd.  With sig_tune_tn_overwrite around the
    # backward that launch WRITES them (no read of the old contents), and the top of the step zeroes only everything else --
    # the gradients that really accumulate (biases, LayerNorm, embeddings, heads) -- with one sig_zero_ranges launch.
    wgrad_overwrite = False
    _zero_tab = None

    def enable_wgrad_overwrite(self) -> bool:
        import os
        lib = _lib.load()
        if not (self.direct_grads and hasattr(lib, "sig_tune_tn_overwrite") and hasattr(lib, "sig_zero_ranges")):
            return False
        if os.environ.get("SIGNAL_WGRAD_OVERWRITE", "1") == "0":       # A/B runs: the one-memset form
            return False
        fl = self.flat
        taken = sorted((fl.offsets[n], fl.byname[n].numel()) for n in self.block_weight_names)
        ranges, pos = [], 0
        for o, n in taken:            # (the alignment padding behind a tensor stays in the zeroed complement)
            if o > pos:
                ranges.append((pos, o - pos))
            pos = o + n
        if pos < fl.total:
            ranges.append((pos, fl.total - pos))
        starts, tot = [], 0
        for _, ln in ranges:
            starts.append(tot)
            tot += (ln + 4095) // 4096
        starts.append(tot)
        self._zero_tab = (torch.tensor(ranges, dtype=torch.int64, device=fl.device), torch.tensor(starts, dtype=torch.int32, device=fl.device),
                          len(ranges), tot)
        self.wgrad_overwrite = True
        return True

    def zero_grads(self):
        """Top of a training step: every gradient that accumulates during the step starts from zero."""
        if self.wgrad_overwrite:
            table, starts, n, tot = self._zero_tab
            _lib.call("sig_zero_ranges", self.flat.grad.data_ptr(), table.data_ptr(), starts.data_ptr(), n, tot, _stream())
        else:
            self.flat.grad.zero_()

    def after_fused_step(self):
        """The fused optimizer already refreshed flat.op16; redo the transposed copies and mark versions."""
        self._pack(cast=False)
        self.flat._versions = tuple(p._version for p in self.flat.params)

    def _pk(self, name):      # 16-bit operand view of a parameter (same layout as f32)
        return self.flat.view(self.flat.op16, name)

    def _g(self, name):       # f32 gradient view
        return self.flat.view(self.flat.grad, name)

    def _p(self, name):
        return self.flat.byname[name].data

    def grads_of(self, prefix, names):
        """One copy of the group's gradient range, returned as per-parameter views (autograd adopts them)."""
        lo, hi = self.flat.span(prefix)
        chunk = self.flat.grad[lo:hi].clone()
        out = []
        for n in names:
            o = self.flat.offsets[n] - lo
            p = self.flat.byname[n]
            out.append(chunk[o:o + p.numel()].view(p.shape))
        return tuple(out)

    def zero_grads_of(self, prefix):
        lo, hi = self.flat.span(prefix)
        self.flat.grad[lo:hi].zero_()

    def _build_structs(self):
        fl, dev = self.flat, self.flat.device
        base = "clip_vision_encoder.base."
        self._byname = fl.byname
        self._transposed: Dict[str, torch.Tensor] = {}
        self._tt = None

        def T(name, rows, cols):   # storage for the transposed bf16 copy of a [rows, cols] matrix
            t = torch.zeros(cols, rows, dtype=self.operand_dtype, device=dev)
            self._transposed[name] = t
            return t

        D, Fd, O = self.D, self.F, self.out_dim
        m = self.model
        cv = "clip_vision_encoder.cv_embed" if "clip_vision_encoder.cv_embed" in self._byname else None
        self.embed_p = fill(_lib.SigEmbedParams, w_conv=self._pk(base + "conv1.weight"),
                            class_embedding=self._p(base + "class_embedding"),
                            positional_embedding=self._p(base + "positional_embedding"),
                            cv_embed=self._p(cv) if cv else None, ln_w=self._p(base + "ln_pre.weight"),
                            ln_b=self._p(base + "ln_pre.bias"), sie_coe=float(m.clip_vision_encoder.sie_xishu))
        self.embed_g = fill(_lib.SigEmbedGrads, w_conv=self._g(base + "conv1.weight"),
                            class_embedding=self._g(base + "class_embedding"),
                            positional_embedding=self._g(base + "positional_embedding"),
                            cv_embed=self._g(cv) if cv else None, ln_w=self._g(base + "ln_pre.weight"),
                            ln_b=self._g(base + "ln_pre.bias"))
        # what sig_embed_bwd writes: the only gradients that are not final before the last stage of the backward (the
        # data-parallel reducer sends everything else earlier)
        self.embed_param_names = [base + "conv1.weight", base + "class_embedding", base + "positional_embedding",
                                  base + "ln_pre.weight", base + "ln_pre.bias"] + ([cv] if cv else [])
        self.block_p, self.block_g = [], []
        self.block_weight_names = []        # the four big matrices of every block: what the grouped weight-gradient launch writes
        for i in range(self.layers):
            p = f"{base}transformer.resblocks.{i}."
            names = dict(w_in=p + "attn.in_proj_weight", w_out=p + "attn.out_proj.weight", w_fc=p + "mlp.c_fc.weight",
                         w_proj=p + "mlp.c_proj.weight", b_in=p + "attn.in_proj_bias", b_out=p + "attn.out_proj.bias",
                         b_fc=p + "mlp.c_fc.bias", b_proj=p + "mlp.c_proj.bias", ln1_w=p + "ln_1.weight",
                         ln1_b=p + "ln_1.bias", ln2_w=p + "ln_2.weight", ln2_b=p + "ln_2.bias")
            shapes = dict(w_in=(3 * D, D), w_out=(D, D), w_fc=(Fd, D), w_proj=(D, Fd))
            self.block_weight_names += [names[k] for k in shapes]
            kw = {}
            for k, n in names.items():
                kw[k] = self._pk(n) if k.startswith("w_") else self._p(n)
            for k, (r, c) in shapes.items():
                kw["wt_" + k[2:]] = T(names[k], r, c)
            self.b_proj_grad = getattr(self, "b_proj_grad", []) if i else []
            self.b_proj_grad.append(self._g(names["b_proj"]))
            self.block_p.append(fill(_lib.SigBlockParams, **kw))
            self.block_g.append(fill(_lib.SigBlockGrads, **{k: self._g(n) for k, n in names.items()}))
        self.proj_t = T(base + "proj", D, O)      # forward operand proj^T [O, D]
        self.head_p = fill(_lib.SigHeadParams, proj_t=self.proj_t, proj=self._pk(base + "proj"),
                           ln_w=self._p(base + "ln_post.weight"), ln_b=self._p(base + "ln_post.bias"))
        self.head_g = fill(_lib.SigHeadGrads, proj=self._g(base + "proj"), ln_w=self._g(base + "ln_post.weight"),
                           ln_b=self._g(base + "ln_post.bias"))
        self.vit_param_names = [n for n in fl.names if n.startswith("clip_vision_encoder.")]

        self.sim_p = self.sim_g = None
        if "SIM.token_selection.W_q.weight" in self._byname:
            s, mi = "SIM.token_selection.", "SIM.modal_interactive."
            d = 512
            inw, inb = mi + "cross_attn.in_proj_weight", mi + "cross_attn.in_proj_bias"
            w_in_bf, g_in = self._pk(inw), self._g(inw)
            b_in, gb_in = self._p(inb), self._g(inb)
            self._sim_T = dict(q=torch.zeros(d, d, dtype=self.operand_dtype, device=dev),
                               kv=torch.zeros(d, 2 * d, dtype=self.operand_dtype, device=dev))
            kr = m.SIM.token_selection.keep_ratio                 # MODEL.KEEP_RATIO when MODEL.FIXED_KEEP_RATIO, else None
            max_keep = 0 if kr is None else int((self.L - 1) * kr)  # useA.py:255-256
            if kr is not None and not 1 <= max_keep <= self.L - 1:
                raise ValueError(f"MODEL.KEEP_RATIO={kr}: int(Lp * ratio) must lie in 1..{self.L - 1}")
            self.sim_p = fill(
                _lib.SigSimParams, sel_wq=self._p(s + "W_q.weight"), sel_bq=self._p(s + "W_q.bias"),
                sel_wk=self._p(s + "W_k.weight"), sel_bk=self._p(s + "W_k.bias"),
                w_q=w_in_bf[:d], w_kv=w_in_bf[d:], w_o=self._pk(mi + "cross_attn.out_proj.weight"),
                w_f1=self._pk(mi + "ffn.0.weight"), w_f2=self._pk(mi + "ffn.2.weight"),
                wt_q=self._sim_T["q"], wt_kv=self._sim_T["kv"], wt_o=T(mi + "cross_attn.out_proj.weight", d, d),
                wt_f1=T(mi + "ffn.0.weight", 2 * d, d), wt_f2=T(mi + "ffn.2.weight", d, 2 * d),
                b_q=b_in[:d], b_kv=b_in[d:], b_o=self._p(mi + "cross_attn.out_proj.bias"),
                b_f1=self._p(mi + "ffn.0.bias"), b_f2=self._p(mi + "ffn.2.bias"),
                n1_w=self._p(mi + "norm1.weight"), n1_b=self._p(mi + "norm1.bias"),
                n2_w=self._p(mi + "norm2.weight"), n2_b=self._p(mi + "norm2.bias"), topk=int(m.SIM.token_selection.k1),
                dtype=self.dt, max_keep=max_keep)
            self.sim_g = fill(
                _lib.SigSimGrads, w_q=g_in[:d], w_kv=g_in[d:], w_o=self._g(mi + "cross_attn.out_proj.weight"),
                w_f1=self._g(mi + "ffn.0.weight"), w_f2=self._g(mi + "ffn.2.weight"), b_q=gb_in[:d], b_kv=gb_in[d:],
                b_o=self._g(mi + "cross_attn.out_proj.bias"), b_f1=self._g(mi + "ffn.0.bias"),
                b_f2=self._g(mi + "ffn.2.bias"), n1_w=self._g(mi + "norm1.weight"), n1_b=self._g(mi + "norm1.bias"),
                n2_w=self._g(mi + "norm2.weight"), n2_b=self._g(mi + "norm2.bias"))
            self.sim_param_names = [n for n in fl.names if n.startswith(mi)]
            self._sim_in_w = inw

        self.das_p = self.das_g = None
        if "AlignM.contra_temp" in self._byname:
            DP, DG = _lib.SigDasParams * 3, _lib.SigDasGrads * 3
            ps, gs = [], []
            for mch in "rnt":
                a = f"AlignM.DAS_{mch}."
                ps.append(fill(_lib.SigDasParams, w_q=self._pk(a + "proj_q.weight"), w_0=self._pk(a + "conv_offset.0.weight"),
                               wt_q=T(a + "proj_q.weight", 512, 512), wt_0=T(a + "conv_offset.0.weight", 512, 512),
                               b_q=self._p(a + "proj_q.bias"), b_0=self._p(a + "conv_offset.0.bias"),
                               wd=self._p(a + "conv_offset.2.weight"), bd=self._p(a + "conv_offset.2.bias"),
                               w4=self._p(a + "conv_offset.4.weight")))
                gs.append(fill(_lib.SigDasGrads, w_q=self._g(a + "proj_q.weight"), w_0=self._g(a + "conv_offset.0.weight"),
                               b_q=self._g(a + "proj_q.bias"), b_0=self._g(a + "conv_offset.0.bias"),
                               wd=self._g(a + "conv_offset.2.weight"), bd=self._g(a + "conv_offset.2.bias"),
                               w4=self._g(a + "conv_offset.4.weight")))
            self.das_p, self.das_g = DP(*ps), DG(*gs)
            self.das_param_names = [n for n in fl.names if n.startswith("AlignM.DAS_")]

    def _pack(self, cast=True):
        """f32 -> 16-bit operand type for every parameter in one kernel, then the transposed copies."""
        fl, st = self.flat, _stream()
        if cast:
            _lib.call("sig_cast_bf16", fl.data.data_ptr(), fl.op16.data_ptr(), fl.total, self.dt, st)
        if self._tt is None:   # descriptor table of every transposed copy (pointers are stable: flat buffers persist)
            # sources are the 16-bit operand mirror (just refreshed by the cast above or by the fused Adam): a pure 16-bit
            # transpose moves half the bytes of cast-and-transpose from the f32 master and gives the same bits
            rows = []
            for name, t in self._transposed.items():
                src = self._pk(name)
                src = src.view(src.shape[0], -1)
                rows.append((src.data_ptr(), t.data_ptr(), src.shape[0], src.shape[1]))
            if self.sim_p is not None:
                w = self._pk(self._sim_in_w)
                rows.append((w[:512].data_ptr(), self._sim_T["q"].data_ptr(), 512, 512))
                rows.append((w[512:].data_ptr(), self._sim_T["kv"].data_ptr(), 1024, 512))
            fast = all(r % 64 == 0 and c % 64 == 0 for _, _, r, c in rows)
            starts, tot = [], 0
            for _, _, r, c in rows:
                starts.append(tot)
                tot += ((r + 63) // 64) * ((c + 63) // 64)
            starts.append(tot)
            if not fast:       # a matrix that is not a multiple of 64 x 64: the general kernel, from the f32 master
                rows = [(self._byname[name].data.data_ptr(), t.data_ptr(), self._byname[name].shape[0],
                         self._byname[name].numel() // self._byname[name].shape[0]) for name, t in self._transposed.items()]
                if self.sim_p is not None:
                    w = self._byname[self._sim_in_w].data
                    rows.append((w[:512].data_ptr(), self._sim_T["q"].data_ptr(), 512, 512))
                    rows.append((w[512:].data_ptr(), self._sim_T["kv"].data_ptr(), 1024, 512))
            self._tt = (torch.tensor(rows, dtype=torch.int64, device=fl.device),
                        torch.tensor(starts, dtype=torch.int32, device=fl.device), len(rows), tot, fast)
        table, starts, n, tot, fast = self._tt
        if fast:
            _lib.call("sig_transpose16_multi", table.data_ptr(), starts.data_ptr(), n, tot, st)
        else:
            _lib.call("sig_transpose_cast_multi", table.data_ptr(), starts.data_ptr(), n, tot, self.dt, st)

    # ------------------------------------------------------------------ backbone
    def _alloc_vit(self, S, B, train):
        dev, D, Fd, O, L, H = self.flat.device, self.D, self.F, self.out_dim, self.L, self.H
        BF = self.operand_dtype
        M, Mt, K = S * L, S * (L - 1), 3 * self.patch * self.patch
        w = _Ws(dev)
        ws = {"S": S, "B": B, "M": M, "train": train}
        ws["dims"] = _lib.SigVitDims(S, B, L, D, H, Fd, O, self.dt)
        ws["patches"], ws["tok"] = w.z(Mt, K, BF), w.z(Mt, D)
        ws["pre_ln"] = w.z(M, D) if train else None
        ws["emean"], ws["erstd"] = (w.v(M), w.v(M)) if train else (None, None)
        # training keeps every block's x_in / x_mid for the LayerNorm backward; inference updates ONE residual stream in place
        # (out_proj and c_proj read the residual from the buffer they write: 134 -> 118 us for c_proj at B = 64, the output
        # lines are already in L2 when they are written)
        nx = self.layers + 1 if train else 1
        ws["x"] = [w.z(M, D) for _ in range(nx)]
        nset = self.layers if train else 1
        acts = []
        for _ in range(nset):
            acts.append(dict(h1=w.z(M, D, BF), mean1=w.v(M), rstd1=w.v(M), qkv=w.z(M, 3 * D, BF), lse=w.v(S * H * L),
                             attn=w.z(M, D, BF), x_mid=w.z(M, D) if train else ws["x"][0], h2=w.z(M, D, BF), mean2=w.v(M), rstd2=w.v(M),
                             u=w.z(M, Fd, BF) if train else None, g=w.z(M, Fd, BF)))
        ws["acts"] = acts
        ws["hp"], ws["hmean"], ws["hrstd"], ws["tokens"] = w.z(M, D, BF), w.v(M), w.v(M), w.z(M, O)
        ws["embed_a"] = fill(_lib.SigEmbedActs, patches=ws["patches"], tok=ws["tok"], pre_ln=ws["pre_ln"], mean=ws["emean"],
                             rstd=ws["erstd"], x0=ws["x"][0])
        ws["block_a"] = []
        for i in range(self.layers):
            a = acts[i if train else 0]
            xin, xout = (ws["x"][i], ws["x"][i + 1]) if train else (ws["x"][0], ws["x"][0])
            ws["block_a"].append(fill(_lib.SigBlockActs, x_in=xin, x_out=xout, **a))
        xl = ws["x"][self.layers] if train else ws["x"][0]
        ws["head_a"] = fill(_lib.SigHeadActs, x=xl, hp=ws["hp"], mean=ws["hmean"], rstd=ws["hrstd"], tokens=ws["tokens"])
        if train:
            ws["du"], ws["dh"], ws["dqkv"] = w.z(M, Fd, BF), w.z(M, D, BF), w.z(M, 3 * D, BF)
            ws["dx_mid"], ws["dx_mid_b"], ws["dx"], ws["dx_b"] = w.z(M, D), w.z(M, D, BF), w.z(M, D), w.z(M, D, BF)
            ws["dtokens"], ws["dtok_b"] = w.z(M, O), w.z(M, O, BF)
            ws["dpre"], ws["dtok_e"] = w.z(M, D), w.z(Mt, D, BF)
            ws["scratch"] = fill(_lib.SigBlockScratch, du=ws["du"], dh=ws["dh"], dqkv=ws["dqkv"], dx_mid=ws["dx_mid"],
                                 dx_mid_b=ws["dx_mid_b"])
        ws["bytes"] = w.bytes
        return ws

    def _get_ws(self, pool, key, make):
        """-> workspace; wrap it in a _WsLease and release() that when done."""
        free = pool.setdefault(key, [])
        ws = free.pop() if free else make()
        ws["_free"] = free
        return ws

    def vit_forward(self, imgs: List[torch.Tensor], cam: Optional[torch.Tensor], train: bool):
        B = imgs[0].shape[0]
        S = len(imgs) * B
        ws = self._get_ws(self._vit_ws, (S, B, train), lambda: self._alloc_vit(S, B, train))
        for im in imgs:
            if tuple(im.shape) != (B, 3, *self.img_hw) or im.dtype != F32 or not im.is_contiguous():
                raise ValueError(f"expected contiguous f32 images [{B},3,{self.img_hw[0]},{self.img_hw[1]}], got {tuple(im.shape)} {im.dtype}")
        ws["cam"] = cam
        st = _stream()
        d = ref(ws["dims"])
        import ctypes
        parts = (ctypes.c_void_p * len(imgs))(*[im.data_ptr() for im in imgs])     # read in place: no staging copy
        _lib.call("sig_embed_fwd", d, ref(self.embed_p), ref(ws["embed_a"]), ctypes.cast(parts, ctypes.c_void_p), len(imgs),
                  None if cam is None else cam.data_ptr(), self.img_hw[0], self.img_hw[1], self.patch, st)
        if train:
            ws["dtokens"].zero_()       # the heads' backward stages accumulate the token gradient here (shared_dtokens)
        for i in range(self.layers):
            _lib.call("sig_block_fwd", d, ref(self.block_p[i]), ref(ws["block_a"][i]), st)
        _lib.call("sig_head_fwd", d, ref(self.head_p), ref(ws["head_a"]), st)
        return ws

    def vit_backward(self, ws):
        """ws["dtokens"] (the gradient of the token tensor, accumulated by the consumers) -> accumulates every ViT parameter
        gradient into flat.grad."""
        st, d = _stream(), ref(ws["dims"])
        # The 25 LayerNorm backwards of this pass chain their column reduces (each launch adds up the previous one's partial rows
        # in its last workgroups; one flush at the end) instead of 25 small reduce launches.  With a data-parallel reducer hooked
        # in, a stage's gradients are final one LayerNorm launch later than its backward is enqueued, so its hook fires after the
        # NEXT stage's backward (see _vit_backward_stages).
        lib = _lib.load()
        chain = hasattr(lib, "sig_tune_ln_defer")
        prev_defer = lib.sig_tune_ln_defer(1) if chain else 0
        self._ln_chained = chain
        prev_ow = lib.sig_tune_tn_overwrite(1) if self.wgrad_overwrite else 0
        try:
            self._vit_backward_stages(ws, st, d)
        finally:
            if self.wgrad_overwrite:
                lib.sig_tune_tn_overwrite(prev_ow)
            if chain:
                lib.sig_tune_ln_defer(prev_defer)
                _lib.call("sig_ln_flush", st)

    def _vit_backward_stages(self, ws, st, d):
        _lib.call("sig_head_bwd", d, ref(self.head_p), ref(ws["head_a"]), ref(self.head_g), ws["dtokens"].data_ptr(),
                  ws["dtok_b"].data_ptr(), ws["dh"].data_ptr(), ws["dx"].data_ptr(), ws["dx_b"].data_ptr(),
                  self.b_proj_grad[self.layers - 1].data_ptr(), st)
        # Hooks of the data-parallel reducer.  With the chained LayerNorm reduce a stage's last LayerNorm launch leaves its column
        # sums (ln weights / biases, a c_proj bias gradient) parked as partial rows; the FIRST LayerNorm launch of the next stage
        # adds them up.  So "stage s is final" holds once the stage after it is enqueued: the head's hook fires after block
        # L-1's backward, block i's after block i-1's, block 0's after an explicit flush.  Without chaining: right away.
        chained = getattr(self, "_ln_chained", False)
        if self.on_head_grads_ready is not None and not chained:     # every head-side gradient (heads ran before this backward) is final now
            self.on_head_grads_ready()
        for i in reversed(range(self.layers)):
            below = self.b_proj_grad[i - 1].data_ptr() if i > 0 else None   # column sums of dx_in = block i-1's c_proj bias grad
            _lib.call("sig_block_bwd", d, ref(self.block_p[i]), ref(ws["block_a"][i]), ref(self.block_g[i]),
                      ref(ws["scratch"]), ws["dx"].data_ptr(), ws["dx_b"].data_ptr(), ws["dx"].data_ptr(),
                      ws["dx_b"].data_ptr(), below, 1, st)
            if chained:
                if i == self.layers - 1:
                    if self.on_head_grads_ready is not None:
                        self.on_head_grads_ready()
                elif self.on_block_grads_ready is not None:
                    self.on_block_grads_ready(i + 1)
            elif self.on_block_grads_ready is not None:
                self.on_block_grads_ready(i)
        if chained and self.on_block_grads_ready is not None:
            _lib.call("sig_ln_flush", st)            # block 0's last LayerNorm launch has no successor yet
            self.on_block_grads_ready(0)
        cam = ws["cam"]
        _lib.call("sig_embed_bwd", d, ref(self.embed_p), ref(ws["embed_a"]), ref(self.embed_g), ws["dx"].data_ptr(),
                  ws["dpre"].data_ptr(), ws["dtok_e"].data_ptr(), None if cam is None else cam.data_ptr(), self.patch, st)

    # ------------------------------------------------------------------ per-step arena of small zero-initialised buffers
    # The head stages (BNNeck, ReID loss) need a dozen small zero-filled accumulators per training step.  torch.zeros launches a
    # fill kernel for each (~5 us + a kernel boundary; ~30 per step, tools/find_fills.py).  The training engine arms an arena at
    # the top of every step: ONE fill over the part that was used, then the buffers are views handed out by a bump pointer.
    # Outside an armed step (tests, inference, plain autograd use) zeros() is torch.zeros.  Arena views are valid until the next
    # armed step begins: only per-step scratch may come from here, nothing a caller keeps.
    _arena = None
    _arena_pos = 0
    _arena_hw = 0
    _arena_armed = False
    _ids_checked = None       # the label tensor whose two-identity check already ran in this step (reid_head.reid_loss)

    def arena_begin(self, device, floats=1 << 21):
        self._ids_checked = None
        if self._arena is None or self._arena.device != device:
            self._arena = torch.zeros(floats, dtype=F32, device=device)
            self._arena_hw = 0
        elif self._arena_hw:
            self._arena[:self._arena_hw].zero_()
        self._arena_pos, self._arena_hw, self._arena_armed = 0, 0, True

    def arena_end(self):
        self._arena_armed = False
        self._ids_checked = None

    def zeros(self, *shape, device=None):
        n = 1
        for d in shape:
            n *= d
        n64 = (n + 63) // 64 * 64
        if self._arena_armed and self._arena_pos + n64 <= self._arena.numel():
            v = self._arena[self._arena_pos:self._arena_pos + n].view(*shape)
            self._arena_pos += n64
            self._arena_hw = max(self._arena_hw, self._arena_pos)
            return v
        return torch.zeros(*shape, dtype=F32, device=device if device is not None else self.flat.device)

    on_block_grads_ready = None  # hooks for the data-parallel reducer (signal_amd/parallel)
    on_head_grads_ready = None

    # The token tensor handed out by a TRAINING forward and the buffer that collects its gradient.  SIM / GAM / LAM
    # backward accumulate straight into that buffer when their input IS this tensor (they return no gradient to autograd),
    # instead of each zero-filling a 50 MB tensor that autograd then adds up and the backbone copies once more.
    _live_tokens = None

    def shared_dtokens(self, tokens: torch.Tensor):
        lt = self._live_tokens
        if lt is None or lt[0] != tokens.data_ptr() or tuple(tokens.shape) != lt[1]:
            return None
        # the SAME tensor (or a full view of it), not merely a tensor that happens to start at the same address
        base = tokens._base if tokens._base is not None else tokens
        live = lt[3]._base if lt[3]._base is not None else lt[3]
        return lt[2] if base is live or tokens is lt[3] else None

    # ------------------------------------------------------------------ SIM
    def _alloc_sim(self, B, train):
        dev, L = self.flat.device, self.L
        BF = self.operand_dtype
        Lp, d = L - 1, 512
        Mq, Mk = 3 * B, 3 * B * Lp
        w = _Ws(dev)
        ws = {"B": B, "train": train}
        a = dict(qprime=w.v(B * 3 * d), cconst=w.v(B * 3), intra=w.v(B * 3 * Lp), inter=w.v(B * 9 * Lp),
                 mask_f=w.v(3 * B * Lp), mask_u8=None, sel=w.z(Mk, d, BF), cls_b=w.z(Mq, d, BF), cls_f=w.z(Mq, d),
                 qh=w.z(Mq, d), kv=w.z(Mk, 2 * d, BF), probs=w.v(B * 24 * 3 * Lp), ao=w.z(Mq, d, BF), y=w.z(Mq, d),
                 z1=w.z(Mq, d), z1_b=w.z(Mq, d, BF), mean1=w.v(Mq), rstd1=w.v(Mq), f1_pre=w.z(Mq, 2 * d, BF),
                 f1=w.z(Mq, 2 * d, BF), y2=w.z(Mq, d), mean2=w.v(Mq), rstd2=w.v(Mq), out=w.z(Mq, d))
        ws["t"] = a
        ws["acts"] = fill(_lib.SigSimActs, **a)
        if train:
            s = dict(dy2=w.z(Mq, d), dy2_b=w.z(Mq, d, BF), df1=w.z(Mq, 2 * d, BF), dz1=w.z(Mq, d), dy=w.z(Mq, d),
                     dy_b=w.z(Mq, d, BF), dao=w.z(Mq, d), dqh=w.z(Mq, d), dqh_b=w.z(Mq, d, BF), dkv=w.z(Mk, 2 * d, BF),
                     dsel=w.z(Mk, d, BF), dcls=w.z(Mq, d))
            ws["s"] = s
            ws["scratch"] = fill(_lib.SigSimScratch, **s)
            ws["dout"] = w.z(Mq, d)
        return ws

    def sim_forward(self, tokens: torch.Tensor, B: int, train: bool, select_only: bool = False):
        ws = self._get_ws(self._sim_ws, (B, train), lambda: self._alloc_sim(B, train))
        name = "sig_sim_select" if select_only else "sig_sim_fwd"
        _lib.call(name, tokens.data_ptr(), B, self.L, ref(self.sim_p), ref(ws["acts"]), _stream())
        return ws

    def sim_backward(self, ws, dout: torch.Tensor, dtokens: torch.Tensor):
        B = ws["B"]
        ws["dout"][:3 * B].copy_(dout.reshape(3 * B, 512))
        _lib.call("sig_sim_bwd", ws["dout"].data_ptr(), B, self.L, ref(self.sim_p), ref(ws["acts"]), ref(self.sim_g),
                  ref(ws["scratch"]), dtokens.data_ptr(), _stream())


    # ------------------------------------------------------------------ GAM / LAM
    def gam_forward(self, tokens, B):
        def make():
            w = _Ws(self.flat.device)
            t = dict(fh=w.v(3 * B * 512), nrm=w.v(3 * B), lv=w.v(B * B), la=w.v(B * B), vec=w.v(4 * B),
                     coef=w.v(2 * B * B + 4 * B + 1), loss=w.v(1))
            return {"B": B, "t": t, "acts": fill(_lib.SigGamActs, **t)}
        ws = self._get_ws(self._gam_ws, (B,), make)
        temp = self.flat.byname["AlignM.contra_temp"].data
        _lib.call("sig_gam_fwd", tokens.data_ptr(), B, self.L, temp.data_ptr(), ref(ws["acts"]), _stream())
        return ws

    def gam_backward(self, ws, dloss, dtokens):
        _lib.call("sig_gam_bwd", ws["B"], self.L, ref(ws["acts"]), dloss.data_ptr(), dtokens.data_ptr(),
                  self._g("AlignM.contra_temp").data_ptr(), _stream())

    def lam_forward(self, tokens, B, train):
        h, wd_ = self.model.h, self.model.w
        P, R = (h // 4) * (wd_ // 4), B * (self.L - 1)

        BF = self.operand_dtype

        def make():
            w = _Ws(self.flat.device)
            Rp = pad_rows(R)
            t = dict(xb=w.z(3 * Rp, 512, BF), q=w.z(3 * Rp, 512, BF), a1=w.z(3 * Rp, 512, BF), a1pre=w.z(3 * Rp, 512, BF),
                     a2pre=w.v(3 * B * P * 512), offs=w.v(3 * B * P * 3), samp=w.v(3 * B * P * 512), loss=w.v(1 + 64))
            ws = {"B": B, "train": train, "t": t, "acts": fill(_lib.SigLamActs, **t)}
            if train:
                s = dict(da1pre=w.z(3 * Rp, 512, BF), dq=w.z(R, 512, BF), dx=w.z(R, 512))
                ws["s"], ws["scratch"] = s, fill(_lib.SigLamScratch, **s)
            return ws
        ws = self._get_ws(self._lam_ws, (B, train), make)
        _lib.call("sig_lam_fwd", tokens.data_ptr(), B, self.L, h, wd_, self.dt, C_ptr(self.das_p), ref(ws["acts"]), _stream())
        return ws

    def lam_backward(self, ws, tokens, dloss, dtokens):
        _lib.call("sig_lam_bwd", tokens.data_ptr(), ws["B"], self.L, self.model.h, self.model.w, self.dt, C_ptr(self.das_p),
                  C_ptr(self.das_g), ref(ws["acts"]), ref(ws["scratch"]), dloss.data_ptr(), dtokens.data_ptr(), _stream())


def C_ptr(arr):
    import ctypes
    return ctypes.cast(arr, ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------------------
# autograd plumbing
# ---------------------------------------------------------------------------------------------------------
class BackboneFn(torch.autograd.Function):
    """tokens[S,L,out] = ViT(images); parameters are inputs so autograd routes the gradients the HIP
    backward accumulated in FlatParams.grad to each Parameter's .grad."""

    @staticmethod
    def forward(ctx, hip: HipPath, cam, n_img, *args):
        imgs, params = args[:n_img], args[n_img:]
        train = hip.grad_mode and any(ctx.needs_input_grad)   # see HipPath.grad_mode
        ws = hip.vit_forward(list(imgs), cam, train)
        ctx.hip, ctx.ws, ctx.n_img, ctx.train, ctx.lease = hip, ws, n_img, train, _WsLease(ws)
        ctx.set_materialize_grads(False)
        M = ws["M"]
        view = ws["tokens"][:M].view(ws["S"], hip.L, hip.out_dim)
        cls = view[:, 0].clone()                     # [S, out]: the ReID heads read only these rows
        if not train:
            out = view.clone()                       # the inference workspace goes back to the pool right away
            ctx.lease.release()
            return out, cls
        import os
        if os.environ.get("SIGNAL_CLONE_TOKENS") == "1":
            # debug aid ("Lifetime of the token tensor"): a private copy that outlives the workspace; the heads'
            # backward stages then return their gradients through autograd instead of accumulating in place
            hip._live_tokens = None
            return view.clone(), cls
        # training: the workspace is held until backward, so the tokens are handed out in place (no 50 MB clone); valid until
        # that backward or the next training forward
        hip._live_tokens = (view.data_ptr(), tuple(view.shape), ws["dtokens"][:M].view(ws["S"], hip.L, hip.out_dim), view)
        return view, cls

    @staticmethod
    def backward(ctx, dtokens, dcls):
        hip, ws = ctx.hip, ctx.ws
        acc = ws["dtokens"][:ws["M"]].view(ws["S"], hip.L, hip.out_dim)
        if dtokens is not None:                      # a consumer that went through plain autograd
            acc.add_(dtokens)
        if dcls is not None:
            acc[:, 0].add_(dcls)
        if hip._live_tokens is not None and hip._live_tokens[2].data_ptr() == acc.data_ptr():
            hip._live_tokens = None
        if hip.direct_grads:
            hip.vit_backward(ws)
            grads = (None,) * len(hip.vit_param_names)
        else:
            hip.zero_grads_of("clip_vision_encoder.")
            hip.vit_backward(ws)
            grads = hip.grads_of("clip_vision_encoder.", hip.vit_param_names)
        ctx.lease.release()
        return (None, None, None) + (None,) * ctx.n_img + grads


class SimFn(torch.autograd.Function):
    """vars_total[B,1536] = SIM(tokens) (useA.py:454-476); masks are constants (no gradient path)."""

    @staticmethod
    def forward(ctx, hip: HipPath, B, tokens, *params):
        train = hip.grad_mode and any(ctx.needs_input_grad)
        tok = tokens.contiguous()   # read row-wise only, so it needs no row padding
        ws = hip.sim_forward(tok, B, train)
        ctx.hip, ctx.ws, ctx.B, ctx.shape, ctx.lease = hip, ws, B, tokens.shape, _WsLease(ws)
        ctx.acc = hip.shared_dtokens(tokens) if train else None
        out = ws["t"]["out"][:3 * B].reshape(B, 3 * 512)
        # training: handed out in place -- the workspace stays leased until SimFn.backward, which autograd runs after the backward
        # of everything that consumed `out`; inference returns the workspace at once, so the caller gets a copy
        if not train:
            out = out.clone()
        mask = ws["t"]["mask_f"].view(3, B, hip.L - 1).clone()      # (kept by the module as last_masks: the caller's own copy)
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)             # (no zero tensor built for the mask's "gradient")
        if not train:
            ctx.lease.release()
        return out, mask

    @staticmethod
    def backward(ctx, dout, _dmask):
        hip, ws = ctx.hip, ctx.ws
        if dout is None:                             # nothing consumed the features: no gradient path through SIM this step
            ctx.lease.release()
            return (None,) * (3 + len(hip.sim_param_names))
        shared = ctx.acc is not None
        dtokens = ctx.acc if shared else torch.zeros(ctx.shape, dtype=F32, device=dout.device)
        if hip.direct_grads:
            hip.sim_backward(ws, dout.contiguous(), dtokens)
            grads = (None,) * len(hip.sim_param_names)
        else:
            hip.zero_grads_of("SIM.modal_interactive.")
            hip.sim_backward(ws, dout.contiguous(), dtokens)
            grads = hip.grads_of("SIM.modal_interactive.", hip.sim_param_names)
        ctx.lease.release()
        return (None, None, None if shared else dtokens) + grads


class GamFn(torch.autograd.Function):
    """loss_area = AlignmentM.Cls_Align(patches) (useB.py:76-126)."""

    @staticmethod
    def forward(ctx, hip: HipPath, B, tokens, contra_temp):
        tok = tokens.contiguous()
        ws = hip.gam_forward(tok, B)
        ctx.hip, ctx.ws, ctx.shape, ctx.lease = hip, ws, tokens.shape, _WsLease(ws)
        ctx.acc = hip.shared_dtokens(tokens) if hip.grad_mode else None
        keep = hip.grad_mode and any(ctx.needs_input_grad)
        # (with a backward to come the workspace stays leased until then: the scalar is handed out in place)
        out = ws["t"]["loss"][0] if keep else ws["t"]["loss"][0].clone()
        if not keep:                                            # no backward will come: hand the workspace back now
            ctx.lease.release()
        return out

    @staticmethod
    def backward(ctx, dloss):
        hip, ws = ctx.hip, ctx.ws
        shared = ctx.acc is not None
        dtokens = ctx.acc if shared else torch.zeros(ctx.shape, dtype=F32, device=dloss.device)
        if hip.direct_grads:
            hip.gam_backward(ws, dloss.contiguous().reshape(1), dtokens)
            dtemp = None
        else:
            hip._g("AlignM.contra_temp").zero_()
            hip.gam_backward(ws, dloss.contiguous().reshape(1), dtokens)
            dtemp = hip._g("AlignM.contra_temp").clone()
        ctx.lease.release()
        return None, None, None if shared else dtokens, dtemp


class LamFn(torch.autograd.Function):
    """patch_loss = AlignmentM.patch_Align(patches) (useB.py:128-167)."""

    @staticmethod
    def forward(ctx, hip: HipPath, B, tokens, *params):
        train = hip.grad_mode and any(ctx.needs_input_grad)
        tok = tokens.contiguous()
        ws = hip.lam_forward(tok, B, train)
        ctx.hip, ctx.ws, ctx.tok, ctx.lease = hip, ws, tok, _WsLease(ws)
        ctx.acc = hip.shared_dtokens(tok) if train else None
        out = ws["t"]["loss"][0] if train else ws["t"]["loss"][0].clone()
        if not train:
            ctx.lease.release()
        return out

    @staticmethod
    def backward(ctx, dloss):
        hip, ws = ctx.hip, ctx.ws
        shared = ctx.acc is not None
        dtokens = ctx.acc if shared else torch.zeros(ctx.tok.shape, dtype=F32, device=dloss.device)
        if hip.direct_grads:
            hip.lam_backward(ws, ctx.tok, dloss.contiguous().reshape(1), dtokens)
            grads = (None,) * len(hip.das_param_names)
        else:
            hip.zero_grads_of("AlignM.DAS_")
            hip.lam_backward(ws, ctx.tok, dloss.contiguous().reshape(1), dtokens)
            grads = hip.grads_of("AlignM.DAS_", hip.das_param_names)
        ctx.lease.release()
        return (None, None, None if shared else dtokens) + grads
