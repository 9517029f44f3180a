"""Learning-rate schedules of the reference's training script (train.py:87-91).

`create_scheduler(cfg, optimizer)` -> the epoch-stepped noisy cosine schedule used for RGBNT201 / RGBNT100
(solver/scheduler_factory.py:7-32 -> cosine_lr.py:61-100 -> scheduler.py:69-105); `WarmupMultiStepLR` -> the MSVR310
schedule (solver/lr_scheduler310.py:14-55).  Both only rewrite `param_group["lr"]` on the host, once per epoch;
`FusedAdam` picks the new per-parameter table up at its next step.

Behaviour kept on purpose:
 * warm-up quirk: during the first WARMUP_ITERS epochs EVERY group ramps linearly from 0.1 * BASE_LR to its own base
   value -- also the 5e-6 backbone groups, which therefore start ~7x ABOVE their base lr;
 * the cosine runs over the raw epoch index (warm-up is not a prefix), floor 0.001 * BASE_LR shared by all groups, one
   cycle; from epoch MAX_EPOCHS on every group sits at the floor;
 * multiplicative noise on every epoch in [0, MAX_EPOCHS): lr *= 1 + n, n ~ N(0,1) from torch.Generator(seed 42 + epoch)
   re-drawn until |n| < 0.67 (the same generator calls as the reference, so the factors are identical);
 * `processor.py:289` logs `scheduler._get_lr(epoch)[0]`, the noise-free value of group 0.
"""
from __future__ import annotations

import math
from bisect import bisect_right
from typing import List

import torch


class NoisyCosineLR:
    """step(epoch) sets lr of every param group; _get_lr(epoch) is the noise-free schedule."""

    NOISE_LIMIT = 0.67
    NOISE_SEED = 42

    def __init__(self, optimizer, epochs: int, floor: float, warmup_start: float, warmup_epochs: int):
        if epochs <= 0 or floor < 0:
            raise ValueError("NoisyCosineLR: epochs must be positive and the floor non-negative")
        self.optimizer = optimizer
        self.epochs, self.floor, self.warmup_start, self.warmup_epochs = int(epochs), float(floor), float(warmup_start), int(warmup_epochs)
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        # the reference constructor leaves every group at the warm-up start value until the first step(epoch)
        self._write([self.warmup_start] * len(self.base_values) if self.warmup_epochs else self.base_values)

    # ---- schedule ----
    def _get_lr(self, epoch: int) -> List[float]:
        if epoch < self.warmup_epochs:
            return [self.warmup_start + epoch * ((b - self.warmup_start) / self.warmup_epochs) for b in self.base_values]
        if epoch >= self.epochs:                      # the single cycle is over
            return [self.floor for _ in self.base_values]
        c = 0.5 * (1.0 + math.cos(math.pi * epoch / self.epochs))
        return [self.floor + (b - self.floor) * c for b in self.base_values]

    def noise_factor(self, epoch: int) -> float:
        if not 0 <= epoch < self.epochs:
            return 1.0
        g = torch.Generator()
        g.manual_seed(self.NOISE_SEED + epoch)
        while True:
            n = torch.randn(1, generator=g).item()
            if abs(n) < self.NOISE_LIMIT:
                return 1.0 + n

    def step(self, epoch: int, metric=None) -> None:
        f = self.noise_factor(epoch)
        self._write([v + v * (f - 1.0) for v in self._get_lr(epoch)])

    def _write(self, values) -> None:
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    # ---- checkpointing (the schedule is stateless apart from its constants) ----
    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd) -> None:
        self.__dict__.update(sd)


def create_scheduler(cfg, optimizer) -> NoisyCosineLR:
    """Reference factory signature (solver/scheduler_factory.py:7)."""
    return NoisyCosineLR(optimizer, epochs=cfg.SOLVER.MAX_EPOCHS, floor=0.001 * cfg.SOLVER.BASE_LR,
                         warmup_start=0.1 * cfg.SOLVER.BASE_LR, warmup_epochs=cfg.SOLVER.WARMUP_ITERS)


class WarmupMultiStepLR(torch.optim.lr_scheduler.LRScheduler):
    """lr = base * gamma^(#milestones <= epoch) * warm-up factor (lr_scheduler310.py:42-55); stepped with .step()."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500, warmup_method="linear",
                 last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError(f"milestones must be increasing, got {milestones}")
        if warmup_method not in ("constant", "linear"):
            raise ValueError(f"warmup_method must be 'constant' or 'linear', got {warmup_method}")
        self.milestones, self.gamma = list(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        w = 1.0
        if self.last_epoch < self.warmup_iters:
            if self.warmup_method == "constant":
                w = self.warmup_factor
            else:
                a = self.last_epoch / self.warmup_iters
                w = self.warmup_factor * (1 - a) + a
        decay = self.gamma ** bisect_right(self.milestones, self.last_epoch)
        return [b * w * decay for b in self.base_lrs]
