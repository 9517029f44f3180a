"""Optimizer construction with the reference's per-parameter rules (solver/make_optimizer.py:4-45): lr and
weight decay chosen by substring of the parameter NAME ("bias", "base", "classifier"), one param group per
parameter.  `make_optimizer` returns a torch.optim-compatible object; for Adam it is FusedAdam, which keeps the
same param_groups (so the reference's schedulers work unchanged) but applies the update with ONE HIP launch over
the model's flat parameter buffer."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib


def param_hyper(cfg, key: str):
    lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
    if "bias" in key:
        lr, wd = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR, cfg.SOLVER.WEIGHT_DECAY_BIAS
    if not cfg.MODEL.FROZEN and "base" in key and "adapter" not in key:
        lr = 0.000005 if cfg.MODEL.TRANSFORMER_TYPE == "ViT-B-16" else cfg.SOLVER.BASE_LR * 0.8
    if cfg.DATASETS.NAMES == "MSVR310" and "classifier" in key:
        lr, wd = cfg.SOLVER.BASE_LR * 100, cfg.SOLVER.WEIGHT_DECAY_BIAS
    if cfg.SOLVER.LARGE_FC_LR and ("classifier" in key or "arcface" in key):
        lr = cfg.SOLVER.BASE_LR * 2
    return lr, wd


# parameters that can never receive a gradient: torch.optim skips them because .grad is None
def gradless(name: str) -> bool:
    return name.startswith("SIM.token_selection.")


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params) semantics, one sig_adam_step launch per step.  Requires the model's parameters
    to live in signal_amd's flat buffer with gradients accumulated in flat.grad (HipPath.enable_direct_grads)."""

    def __init__(self, named_groups, hip, betas=(0.9, 0.999), eps=1e-8):
        self.hip = hip
        self._names = [n for n, _ in named_groups]
        super().__init__([g for _, g in named_groups], dict(lr=1e-3, weight_decay=0.0, betas=betas, eps=eps))
        fl = hip.flat
        self.m = torch.zeros_like(fl.data)
        self.v = torch.zeros_like(fl.data)
        self.t = 0
        ends = []
        self._by_name = {n: i for i, n in enumerate(self._names)}
        for n in fl.names:
            ends.append(fl.offsets[n] + (fl.byname[n].numel() + 63) // 64 * 64)
        self.seg_end = torch.tensor(ends, dtype=torch.int32, device=fl.device)
        self.seg_lr = torch.zeros(len(ends), dtype=torch.float32, device=fl.device)
        self.seg_wd = torch.zeros(len(ends), dtype=torch.float32, device=fl.device)
        self._host = None
        self.grad_scale = 1.0
        self.set_inactive(gradless)

    def set_inactive(self, pred):
        """Parameters with no gradient path (pred(name) true) or frozen ones are left untouched -- lr 0 and weight decay 0
        for their segment -- which is what torch.optim does with .grad = None."""
        fl = self.hip.flat
        self._seg_group = [None if pred(n) else self._by_name.get(n) for n in fl.names]
        self._host = None

    # ---- adopting a torch.optim.Adam built by the reference's make_optimizer on a CPU model (train.py:85) -------------
    @staticmethod
    def can_adopt(opt) -> bool:
        if type(opt) is not torch.optim.Adam or len(opt.state) != 0:
            return False
        d = opt.defaults
        plain = not (d.get("amsgrad") or d.get("maximize") or d.get("capturable") or d.get("differentiable"))
        return plain and all(len(g["params"]) == 1 and g["params"][0].is_cuda for g in opt.param_groups)

    @classmethod
    def adopt(cls, opt, model):
        """Same param_groups LIST and dicts as `opt` (a scheduler built on `opt` keeps steering the learning rates),
        update applied by the fused kernel."""
        hip = model.hip
        name_of = {id(p): n for n, p in model.named_parameters()}
        named = []
        for g in opt.param_groups:
            n = name_of.get(id(g["params"][0]))
            if n is None:
                raise ValueError("FusedAdam.adopt: the optimizer holds a parameter that is not the model's")
            named.append((n, g))
        g0 = opt.param_groups[0]
        self = cls(named, hip, betas=tuple(g0.get("betas", opt.defaults["betas"])), eps=float(g0.get("eps", opt.defaults["eps"])))
        self.param_groups = opt.param_groups
        return self

    def _sync_table(self):
        host = tuple((0.0, 0.0) if gi is None else (float(self.param_groups[gi]["lr"]), float(self.param_groups[gi]["weight_decay"]))
                     for gi in self._seg_group)
        if host != self._host:
            self._host = host
            self.seg_lr.copy_(torch.tensor([h[0] for h in host], dtype=torch.float32), non_blocking=True)
            self.seg_wd.copy_(torch.tensor([h[1] for h in host], dtype=torch.float32), non_blocking=True)

    @torch.no_grad()
    def step(self, closure=None, scaler=None):
        """scaler (engine.trainer.DeviceLossScaler) = fp16 mode: the gradient is divided by the device-resident loss scale,
        the whole update is skipped when scaler.check() found a non-finite gradient, and the bias-correction step count is
        the device's count of APPLIED steps (GradScaler.step semantics)."""
        fl = self.hip.flat
        self._sync_table()
        self.t += 1
        b1, b2 = self.defaults["betas"]
        _lib.call("sig_adam_step", fl.data.data_ptr(), fl.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                  fl.op16.data_ptr(), self.hip.dt, self.seg_end.data_ptr(), self.seg_lr.data_ptr(), self.seg_wd.data_ptr(),
                  len(self._seg_group), float(b1), float(b2), float(self.defaults["eps"]), self.t, float(self.grad_scale),
                  None if scaler is None else scaler.state.data_ptr(), fl.total, torch.cuda.current_stream().cuda_stream)
        self.hip.after_fused_step()

    def zero_grad(self, set_to_none: bool = False):
        self.hip.zero_grads()


def make_optimizer(cfg, model, center_criterion=None):
    """Reference signature.  Returns (optimizer, optimizer_center); optimizer_center is None (CenterLoss unused)."""
    named = []
    for key, value in model.named_parameters():
        if not value.requires_grad:
            continue
        lr, wd = param_hyper(cfg, key)
        named.append((key, {"params": [value], "lr": lr, "weight_decay": wd}))
    name = cfg.SOLVER.OPTIMIZER_NAME
    if name == "Adam" and next(model.parameters()).is_cuda:
        hip = model.hip
        hip.prepare(next(model.parameters()).device)
        hip.enable_direct_grads(skip=gradless)
        return FusedAdam(named, hip), None
    groups = [g for _, g in named]
    if name == "SGD":
        return torch.optim.SGD(groups, momentum=cfg.SOLVER.MOMENTUM), None
    if name == "AdamW":
        return torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY), None
    return getattr(torch.optim, name)(groups), None
