"""One optimisation step of the reference's training loop (engine/processor.py:141-261) as an engine object:
zero grads -> model(training=True) -> loss assembly -> backward (gradients accumulate in the flat buffer, block
buckets all-reduced while the lower blocks still run) -> fused Adam.  bf16 needs no GradScaler."""
from __future__ import annotations

import torch

from ..layers.make_loss import make_loss, total_loss
from ..parallel.reducer import GradReducer, plan_buckets
from ..solver.make_optimizer import gradless, make_optimizer


class TrainStep:
    def __init__(self, cfg, model, num_classes, world_size=1, loss_fn=None, optimizer=None):
        self.cfg, self.model = cfg, model
        self.stage = cfg.MODEL.stageName.strip()
        self.loss_fn = loss_fn or make_loss(cfg, num_classes)[0]
        hip = model.hip
        hip.prepare(next(model.parameters()).device)
        hip.enable_direct_grads()
        self.optimizer = optimizer or make_optimizer(cfg, model, None)[0]
        self.fused = hasattr(self.optimizer, "grad_scale")
        self.world = world_size
        self.reducer = None
        if world_size > 1:
            fl = hip.flat
            sizes = {n: fl.byname[n].numel() for n in fl.names}
            blocks, rest = plan_buckets(fl.names, fl.offsets, sizes, fl.total, skip=gradless)
            self.reducer = GradReducer(fl.grad, blocks, rest)
            self.reducer.broadcast_params(fl.data)      # DDP's construction-time broadcast from rank 0
            hip._pack()
            hip.on_block_grads_ready = self.reducer.on_block_ready
            if self.fused:
                self.optimizer.grad_scale = 1.0 / world_size
        model.train()

    def step(self, img, target, target_cam, target_view=None):
        hip = self.model.hip
        hip.flat.grad.zero_()
        out = self.model(img, label=target, cam_label=target_cam, view_label=target_view, training=True, sge=self.stage)
        loss = total_loss(self.cfg, out, self.loss_fn, target, target_cam, self.stage)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
            if not self.fused:
                hip.flat.grad.mul_(1.0 / self.world)
        self.optimizer.step()
        self.last_output = out
        return loss
