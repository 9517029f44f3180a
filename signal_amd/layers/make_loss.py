"""ReID loss of the train step (layers/make_loss.py:29-193, softmax_loss.py:4-34, triplet_loss.py:16-135 of the
reference): ID_LOSS_WEIGHT * label-smoothed CE + TRIPLET_LOSS_WEIGHT * batch-hard triplet (soft-margin when
MODEL.NO_MARGIN, the default).  Same factory signature: make_loss(cfg, num_classes) -> (loss_func, center_criterion).

The loss runs in HIP only (signal_amd/csrc/reid.hip via modeling/reid_head.py: CE + Gram + batch-hard mining + both
gradients); host tensors are refused like everywhere else in signal_amd -- the CPU restatement lives in oracle/ and is
test infrastructure.  Not carried over: the list-of-heads form of `loss_func` (part-based models; Signal returns plain
tensors) and the unused CenterLoss."""
from __future__ import annotations

import torch

from .. import _lib


def _device_pair(score, feat, target):
    if isinstance(score, (list, tuple)) or isinstance(feat, (list, tuple)):
        raise NotImplementedError("list-of-heads scores / features (make_loss.py:120-146) are not used by Signal and are not "
                                  "implemented on the HIP path")
    for name, t in (("score", score), ("feat", feat), ("target", target)):
        if not torch.is_tensor(t) or not t.is_cuda:
            raise _lib.SignalHipError(f"loss_func: {name} must be a device tensor -- signal_amd has no CPU path (the CPU "
                                      f"oracle lives in oracle/ and is test-only)")


def make_loss(cfg, num_classes):
    from ..modeling.reid_head import reid_loss
    sampler = cfg.DATALOADER.SAMPLER
    if "triplet" not in cfg.MODEL.METRIC_LOSS_TYPE:
        raise ValueError(f"expected METRIC_LOSS_TYPE should be triplet but got {cfg.MODEL.METRIC_LOSS_TYPE}")
    eps = 0.1 if cfg.MODEL.IF_LABELSMOOTH == "on" else 0.0
    margin = None if cfg.MODEL.NO_MARGIN else cfg.SOLVER.MARGIN

    # loss_func.hip (set by the training engine): the model's HipPath, whose per-step arena then provides the loss kernels'
    # zero-filled scratch instead of one torch.zeros launch each
    if sampler == "softmax":
        def loss_func(score, feat, target, target_cam):          # plain cross entropy (make_loss.py:104-106)
            _device_pair(score, feat, target)
            return reid_loss(score, feat, target, 0.0, 1.0, 0.0, None, getattr(loss_func, "hip", None))
    elif sampler == "softmax_triplet":
        def loss_func(score, feat, target, target_cam):
            _device_pair(score, feat, target)
            return reid_loss(score, feat, target, eps, cfg.MODEL.ID_LOSS_WEIGHT, cfg.MODEL.TRIPLET_LOSS_WEIGHT, margin,
                             getattr(loss_func, "hip", None))
    else:
        raise ValueError(f"expected sampler should be softmax or softmax_triplet but got {sampler}")
    return loss_func, None   # the reference's CenterLoss is built but never used with METRIC_LOSS_TYPE='triplet'


class _WeightedSum(torch.autograd.Function):
    """sum_i w_i * term_i over 0-dim device tensors as two launches forward (stack, dot) and one backward (g * w; the terms'
    gradients are views of it), instead of one add / multiply kernel per term in each direction."""
    _w = {}

    @staticmethod
    def forward(ctx, weights, *terms):
        dev = terms[0].device
        key = (weights, dev)
        w = _WeightedSum._w.get(key)
        if w is None:
            w = _WeightedSum._w[key] = torch.tensor(weights, dtype=torch.float32, device=dev)
        ctx.w = w
        return torch.dot(torch.stack([t.reshape(()).float() for t in terms]), w)

    @staticmethod
    def backward(ctx, g):
        return (None, *(g * ctx.w).unbind(0))


def total_loss(cfg, output, loss_fn, target, target_cam, stage):
    """Loss assembly of engine/processor.py:176-256: sum of loss_fn over the (score, feat) pairs
    + Gram_Loss_weight * loss_area (+ PAT_Loss_weight * patch_loss)."""
    sign = output[0]
    alpha, beta = cfg.MODEL.Gram_Loss_weight, cfg.MODEL.PAT_Loss_weight
    tail = 0 if sign in (1, 2) else (1 if stage == "CLS" else 2)
    terms, weights = [], []
    for i in range(1, len(output) - tail, 2):
        terms.append(loss_fn(score=output[i], feat=output[i + 1], target=target, target_cam=target_cam))
        weights.append(1.0)
    if tail == 1:
        terms.append(output[-1]); weights.append(float(alpha))
    elif tail == 2:
        terms += [output[-2], output[-1]]; weights += [float(alpha), float(beta)]
    return _WeightedSum.apply(tuple(weights), *terms)
