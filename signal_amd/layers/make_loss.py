"""ReID loss of the train step (layers/make_loss.py:29-193, softmax_loss.py:4-34, triplet_loss.py:16-135 of the
reference): ID_LOSS_WEIGHT * label-smoothed CE + TRIPLET_LOSS_WEIGHT * batch-hard triplet (soft-margin when
MODEL.NO_MARGIN, the default).  Same factory signature: make_loss(cfg, num_classes) -> (loss_func, center_criterion).
On device tensors the loss runs in HIP (signal_amd/csrc/reid.hip via modeling/reid_head.py); the PyTorch code
below is the host-tensor / list-of-heads form and what the CPU tests compare against the reference fixture."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class CrossEntropyLabelSmooth(nn.Module):
    """y = (1 - eps) * onehot + eps / K ;  loss = (-y * log_softmax(x)).mean(0).sum()"""

    def __init__(self, num_classes, epsilon=0.1, use_gpu=True):
        super().__init__()
        self.num_classes, self.epsilon = num_classes, epsilon

    def forward(self, inputs, targets):
        logp = F.log_softmax(inputs.float(), dim=1)
        y = torch.full_like(logp, self.epsilon / self.num_classes)
        y.scatter_(1, targets.view(-1, 1), 1.0 - self.epsilon + self.epsilon / self.num_classes)
        return (-y * logp).mean(0).sum()


def euclidean_dist(x, y):
    xx = x.pow(2).sum(1, keepdim=True)
    yy = y.pow(2).sum(1, keepdim=True).t()
    return (xx + yy - 2.0 * x @ y.t()).clamp(min=1e-12).sqrt()


def hard_example_mining(dist_mat, labels):
    """hardest positive (max over same id, self included) and hardest negative (min over other ids) per anchor"""
    same = labels.view(-1, 1) == labels.view(1, -1)
    if same.is_cuda:      # no host sync on the hot path: device-side assertion
        torch._assert_async(~same.all(), "batch-hard mining needs at least two identities in the batch")
    elif bool(same.all()):
        raise ValueError("batch-hard mining needs at least two identities in the batch (triplet_loss.py:79-84)")
    d_ap = torch.where(same, dist_mat, dist_mat.new_full((), float("-inf"))).max(1).values
    d_an = torch.where(same, dist_mat.new_full((), float("inf")), dist_mat).min(1).values
    return d_ap, d_an


class TripletLoss:
    def __init__(self, margin=None, hard_factor=0.0):
        self.margin, self.hard_factor = margin, hard_factor

    def __call__(self, global_feat, labels, normalize_feature=False):
        f = global_feat.float()
        if normalize_feature:
            f = f / (f.norm(dim=-1, keepdim=True) + 1e-12)
        d_ap, d_an = hard_example_mining(euclidean_dist(f, f), labels)
        d_ap, d_an = d_ap * (1.0 + self.hard_factor), d_an * (1.0 - self.hard_factor)
        if self.margin is not None:
            loss = F.relu(d_ap - d_an + self.margin).mean()
        else:
            loss = F.softplus(-(d_an - d_ap)).mean()          # nn.SoftMarginLoss with y = 1
        return loss, d_ap, d_an


def make_loss(cfg, num_classes):
    sampler = cfg.DATALOADER.SAMPLER
    if "triplet" not in cfg.MODEL.METRIC_LOSS_TYPE:
        raise ValueError(f"expected METRIC_LOSS_TYPE should be triplet but got {cfg.MODEL.METRIC_LOSS_TYPE}")
    triplet = TripletLoss() if cfg.MODEL.NO_MARGIN else TripletLoss(cfg.SOLVER.MARGIN)
    smooth = cfg.MODEL.IF_LABELSMOOTH == "on"
    xent = CrossEntropyLabelSmooth(num_classes=num_classes) if smooth else (lambda s, t: F.cross_entropy(s.float(), t))

    def _id(score, target):
        if isinstance(score, list):
            rest = sum(xent(s, target) for s in score[1:]) / len(score[1:])
            return 0.5 * rest + 0.5 * xent(score[0], target)
        return xent(score, target)

    def _tri(feat, target):
        if isinstance(feat, list):
            rest = sum(triplet(f, target)[0] for f in feat[1:]) / len(feat[1:])
            return 0.5 * rest + 0.5 * triplet(feat[0], target)[0]
        return triplet(feat, target)[0]

    if sampler == "softmax":
        def loss_func(score, feat, target, target_cam):
            return F.cross_entropy(score.float(), target)
    elif sampler == "softmax_triplet":
        eps = 0.1 if smooth else 0.0
        margin = None if cfg.MODEL.NO_MARGIN else cfg.SOLVER.MARGIN

        def loss_func(score, feat, target, target_cam):
            if torch.is_tensor(score) and torch.is_tensor(feat) and score.is_cuda:
                # device tensors: one fused HIP path (sig_reid_loss) -- CE + gram + batch-hard mining + both gradients
                from ..modeling.reid_head import reid_loss
                return reid_loss(score, feat, target, eps, cfg.MODEL.ID_LOSS_WEIGHT, cfg.MODEL.TRIPLET_LOSS_WEIGHT, margin)
            # host tensors / the list-of-heads form of the reference: plain PyTorch
            return cfg.MODEL.ID_LOSS_WEIGHT * _id(score, target) + cfg.MODEL.TRIPLET_LOSS_WEIGHT * _tri(feat, target)
    else:
        raise ValueError(f"expected sampler should be softmax or softmax_triplet but got {sampler}")
    return loss_func, None   # the reference's CenterLoss is built but never used with METRIC_LOSS_TYPE='triplet'


def total_loss(cfg, output, loss_fn, target, target_cam, stage):
    """Loss assembly of engine/processor.py:176-256: sum of loss_fn over the (score, feat) pairs
    + Gram_Loss_weight * loss_area (+ PAT_Loss_weight * patch_loss)."""
    sign = output[0]
    alpha, beta = cfg.MODEL.Gram_Loss_weight, cfg.MODEL.PAT_Loss_weight
    tail = 0 if sign in (1, 2) else (1 if stage == "CLS" else 2)
    loss = 0
    for i in range(1, len(output) - tail, 2):
        loss = loss + loss_fn(score=output[i], feat=output[i + 1], target=target, target_cam=target_cam)
    if tail == 1:
        loss = loss + alpha * output[-1]
    elif tail == 2:
        loss = loss + alpha * output[-2] + beta * output[-1]
    return loss
