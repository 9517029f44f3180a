"""HIP ReID head: BNNeck + classifier (make_model.py:194-219) and the ID + triplet loss (layers/make_loss.py)
as autograd Functions over the C ABI (sig_bnneck_*, sig_reid_loss).  fp32; device tensors only."""
from __future__ import annotations

import torch

from .. import _lib


def _st():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


class BnneckClassifierFn(torch.autograd.Function):
    """score = classifier(bottleneck(feat)), BatchNorm1d in training mode (batch statistics)."""

    @staticmethod
    def forward(ctx, feat, bn_w, bn_b, cls_w, run_mean, run_var, momentum, hip=None):
        # hip (training engine, direct_grads): the parameter gradients go straight into their flat-buffer views (zero at the top
        # of the step) and the per-step zero-filled scratch comes from the engine's arena: no fill / add launches here
        ctx.hip = hip if (hip is not None and hip.direct_grads) else None
        ctx.params = (bn_w, bn_b, cls_w)
        x = feat.contiguous().float()
        B, F = x.shape
        C = cls_w.shape[0]
        dev = x.device
        y = torch.empty(B, F, device=dev)
        mean, rstd = torch.empty(F, device=dev), torch.empty(F, device=dev)
        logits = torch.empty(B, C, device=dev)
        _lib.call("sig_bnneck_fwd", x.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(), _p(run_mean), _p(run_var), float(momentum),
                  cls_w.data_ptr(), B, F, C, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), logits.data_ptr(), _st())
        ctx.save_for_backward(x, y, bn_w, mean, rstd, cls_w)
        ctx.need_b = bn_b.requires_grad
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        x, y, bn_w, mean, rstd, cls_w = ctx.saved_tensors
        B, F = x.shape
        C = cls_w.shape[0]
        dev = x.device
        dl = dlogits.contiguous().float()
        hip = ctx.hip
        pw, pb, pc = ctx.params
        direct = hip is not None and pw.grad is not None and pc.grad is not None and (pb.grad is not None or not ctx.need_b)
        zeros = hip.zeros if hip is not None else (lambda *sh: torch.zeros(*sh, device=dev))
        dx, dy = zeros(B, F), torch.empty(B, F, device=dev)
        if direct:          # the kernels accumulate (+=): the flat gradient views are zero at the top of the step
            dw, dcls, db = pw.grad, pc.grad, (pb.grad if ctx.need_b else None)
        else:
            dw, dcls = zeros(F), zeros(C, F)
            db = zeros(F) if ctx.need_b else None
        _lib.call("sig_bnneck_bwd", x.data_ptr(), y.data_ptr(), bn_w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), cls_w.data_ptr(),
                  dl.data_ptr(), B, F, C, dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), _p(db), dcls.data_ptr(), _st())
        if direct:
            return dx, None, None, None, None, None, None, None
        return dx, dw, db, dcls, None, None, None, None


def bnneck_classifier(bn: torch.nn.BatchNorm1d, cls: torch.nn.Linear, feat: torch.Tensor, hip=None) -> torch.Tensor:
    if not feat.is_cuda:
        raise _lib.SignalHipError("signal_amd's ReID head runs on the GPU only")
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return BnneckClassifierFn.apply(feat, bn.weight, bn.bias, cls.weight, bn.running_mean, bn.running_var,
                                    0.1 if bn.momentum is None else bn.momentum, hip)


class ReidLossFn(torch.autograd.Function):
    """id_weight * CE_labelsmooth(score) + triplet_weight * batch-hard triplet(feat)."""

    @staticmethod
    def forward(ctx, score, feat, target, eps, w_id, w_tri, margin, hip=None):
        ctx.hip = hip
        s, f = score.contiguous().float(), feat.contiguous().float()
        t = target.contiguous().to(torch.int64)
        B, C = s.shape
        F = f.shape[1]
        dev = s.device
        loss = hip.zeros(1) if hip is not None else torch.zeros(1, device=dev)
        gram, coef = torch.empty(B, B, device=dev), torch.empty(2 * B, device=dev)
        pidx, nidx = torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
        _lib.call("sig_reid_loss", s.data_ptr(), f.data_ptr(), t.data_ptr(), B, F, C, float(eps), float(w_id), float(w_tri), float(margin),
                  None, loss.data_ptr(), None, gram.data_ptr(), pidx.data_ptr(), nidx.data_ptr(), coef.data_ptr(), None, _st())
        ctx.save_for_backward(s, f, t)
        ctx.hp = (float(eps), float(w_id), float(w_tri), float(margin))
        return loss[0] if hip is not None and hip._arena_armed else loss[0].clone()   # (a fresh per-step buffer either way)

    @staticmethod
    def backward(ctx, gout):
        s, f, t = ctx.saved_tensors
        eps, w_id, w_tri, margin = ctx.hp
        B, C = s.shape
        F = f.shape[1]
        dev = s.device
        up = gout.contiguous().float().reshape(1)
        zeros = ctx.hip.zeros if ctx.hip is not None else (lambda *sh: torch.zeros(*sh, device=dev))
        dlogits, dfeat = torch.empty(B, C, device=dev), zeros(B, F)
        scratch = zeros(1)
        gram, coef = torch.empty(B, B, device=dev), torch.empty(2 * B, device=dev)
        pidx, nidx = torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
        _lib.call("sig_reid_loss", s.data_ptr(), f.data_ptr(), t.data_ptr(), B, F, C, eps, w_id, w_tri, margin, up.data_ptr(),
                  scratch.data_ptr(), dlogits.data_ptr(), gram.data_ptr(), pidx.data_ptr(), nidx.data_ptr(), coef.data_ptr(),
                  dfeat.data_ptr(), _st())
        return dlogits, dfeat, None, None, None, None, None, None


def reid_loss(score, feat, target, eps, w_id, w_tri, margin, hip=None):
    if w_tri != 0.0:   # the reference fails on a one-identity batch (triplet_loss.py:79-84); device-side check, no host sync
        # (inside an engine step the check runs once per step: every (score, feature) pair of the step sees the same labels)
        if hip is None or not hip._arena_armed or hip._ids_checked is not target:
            torch._assert_async((target != target.reshape(-1)[0]).any(), "batch-hard mining needs at least two identities in the batch")
            if hip is not None and hip._arena_armed:
                hip._ids_checked = target
    return ReidLossFn.apply(score, feat, target, eps, w_id, w_tri, -1.0 if margin is None else float(margin), hip)
