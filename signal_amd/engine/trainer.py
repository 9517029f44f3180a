"""One optimisation step of the reference's training loop (engine/processor.py:141-261) as an engine object:
zero grads -> model(training=True) -> loss assembly -> backward (gradients accumulate in the flat buffer, block
buckets all-reduced while the lower blocks still run) -> fused Adam.

Mixed precision follows the reference's AMP recipe (processor.py:119,165,259-261: autocast + GradScaler) with the
scaler kept on the device: with fp16 MFMA operands the loss gradient is multiplied by a dynamic scale, the optimizer
unscales, skips the update when a gradient overflowed and the scale backs off / grows -- no host synchronisation
anywhere.  bf16 operands need no scaling (scaler is None)."""
from __future__ import annotations

import os

import torch

from .. import _lib
from ..layers.make_loss import make_loss, total_loss
from ..parallel.reducer import GradReducer, plan_buckets, split_rest
from ..solver.make_optimizer import FusedAdam, gradless, make_optimizer


class DeviceLossScaler:
    """torch.cuda.amp.GradScaler semantics (init 65536, x2 after 2000 clean steps, x0.5 on overflow, the step is
    skipped when any gradient is non-finite) with the whole state in one device tensor:
    state = [scale, 1/scale, found_inf, growth_tracker, optimizer_step_count]."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 1.0 / init_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.init_scale = float(init_scale)

    @property
    def scale_tensor(self):          # 0-dim device view: the upstream gradient handed to loss.backward()
        return self.state[0]

    def check(self, flat_grad: torch.Tensor):
        """found_inf |= any non-finite gradient (after the all-reduce, so every rank takes the same decision)."""
        _lib.call("sig_grad_check", flat_grad.data_ptr(), flat_grad.numel(), self.state.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)

    def update(self):
        _lib.call("sig_loss_scale_update", self.state.data_ptr(), self.growth_factor, self.backoff_factor, self.growth_interval,
                  torch.cuda.current_stream().cuda_stream)

    def get_scale(self) -> float:    # host sync: logging / tests only
        return float(self.state[0].item())

    def describe(self):
        return {"init": self.init_scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "state": "device-resident, no host sync"}


class TrainStep:
    def __init__(self, cfg, model, num_classes, world_size=1, loss_fn=None, optimizer=None, stage=None, force_reducer=False):
        """force_reducer: build and drive the data-parallel reducer even with world_size 1 (needs an initialised process group):
        a single GPU then executes the complete exchange path -- bucket plan, hooks, asynchronous RCCL all-reduces on their side
        stream under the backward, the reserved-CU sizing of an nccl group -- with the identity as the sum (tests, bench --ddp-single)."""
        self.cfg, self.model = cfg, model
        # the raw string, as the reference compares it (its default 'CLS ' -- with the space -- takes the GAM+LAM branch)
        self.stage = cfg.MODEL.stageName if stage is None else stage
        self.loss_fn = loss_fn or make_loss(cfg, num_classes)[0]
        hip = model.hip
        try:
            self.loss_fn.hip = hip          # the loss kernels' per-step scratch comes from the engine's arena (layers/make_loss.py)
        except AttributeError:
            pass                            # a caller's own callable that takes no attributes: it keeps torch.zeros
        hip.prepare(next(model.parameters()).device)
        # parameters that cannot receive a gradient in this stage keep .grad = None, exactly what autograd leaves the
        # reference with: torch optimizers skip them (no weight decay on W_q/W_k, no moment updates)
        hip.enable_direct_grads(skip=self.inactive)
        hip.enable_wgrad_overwrite()
        optimizer = optimizer or make_optimizer(cfg, model, None)[0]
        if not isinstance(optimizer, FusedAdam) and FusedAdam.can_adopt(optimizer):
            # the reference's train.py builds the optimizer while the model is still on the CPU; keep its param_groups
            # (the scheduler holds them) and run the update as one HIP launch over the flat buffer
            optimizer = FusedAdam.adopt(optimizer, model)
        self.optimizer = optimizer
        self.fused = isinstance(optimizer, FusedAdam)
        if self.fused:
            self.optimizer.set_inactive(self.inactive)
        self.scaler = None
        if hip.operand_dtype == torch.float16:
            if not self.fused:
                raise NotImplementedError("fp16 operands need the fused Adam (device-side loss scaling); got "
                                          f"{type(optimizer).__name__}")
            self.scaler = DeviceLossScaler(hip.flat.device)
        self.world = world_size
        self.reducer = None
        self._one = None
        # CUs left to RCCL's channel workgroups while the gradient buckets are all-reduced under the backward pass: the
        # one-block-per-CU GEMMs are sized in rounds of the free CUs (include/signal_hip.h, sig_tune_reserved_cus)
        # Default 16 with an nccl (= RCCL) group, 0 otherwise; SIGNAL_RESERVED_CUS forces a value with any backend (the
        # gloo tests use it to run the backward with the re-sized tiles and grids).  Sizing for 240 CUs costs a single GPU
        # +0.3 % (measured); not reserving costs a second round on every 234-tile GEMM that meets a bucket in flight.
        # UNMEASURED against real RCCL traffic (no multi-GPU box this round or the last).
        self.reserved_cus = 0
        ddp = (world_size > 1 or force_reducer) and torch.distributed.is_initialized()
        if ddp:
            env = os.environ.get("SIGNAL_RESERVED_CUS")
            if env is not None:
                self.reserved_cus = int(env)
            elif torch.distributed.get_backend() == "nccl":
                self.reserved_cus = 16
        if self.reserved_cus and not hasattr(_lib.load(), "sig_tune_reserved_cus"):
            self.reserved_cus = 0          # an older library without the tuning exports (_lib._TUNING_ONLY): nothing to size
        if ddp:
            fl = hip.flat
            sizes = {n: fl.byname[n].numel() for n in fl.names}
            blocks, _ = plan_buckets(fl.names, fl.offsets, sizes, fl.total, skip=self.inactive)
            early, late = split_rest(fl.names, fl.offsets, sizes, skip=self.inactive, late_names=hip.embed_param_names)
            self.reducer = GradReducer(fl.grad, blocks, late, rest_early=early, force=force_reducer)
            self.reducer.broadcast_params(fl.data)      # DDP's construction-time broadcast from rank 0 (parameters ...
            self._buffers = [b for _, b in model.named_buffers()]
            self.reducer.broadcast_buffers(self._buffers)   # ... and buffers)
            hip._pack()
            hip.on_block_grads_ready = self.reducer.on_block_ready
            hip.on_head_grads_ready = self.reducer.on_head_ready
            if self.fused:
                self.optimizer.grad_scale = 1.0 / world_size
        model.train()

    def inactive(self, name: str) -> bool:
        """No gradient path in this stage: SIM.token_selection.* always (useA.py:46-48,155-157), AlignM.DAS_* while only
        GAM runs (useB.py:181-184)."""
        return gradless(name) or (self.stage == "CLS" and name.startswith("AlignM.DAS_"))

    def step(self, img, target, target_cam, target_view=None):
        hip = self.model.hip
        hip.zero_grads()                     # (the blocks' weight gradients are overwritten by the backward, not zeroed: hip.wgrad_overwrite)
        hip.arena_begin(hip.flat.device)     # ONE fill for every small zero-initialised buffer of the step's head stages
        try:
            return self._step(hip, img, target, target_cam, target_view)
        finally:
            hip.arena_end()

    def _step(self, hip, img, target, target_cam, target_view):
        if self.reducer is not None:
            self.reducer.broadcast_buffers(self._buffers)    # DDP broadcast_buffers: rank 0's BN running statistics, every forward
        out = self.model(img, label=target, cam_label=target_cam, view_label=target_view, training=True, sge=self.stage)
        loss = total_loss(self.cfg, out, self.loss_fn, target, target_cam, self.stage)
        if self.reserved_cus:
            _lib.load().sig_tune_reserved_cus(self.reserved_cus)
        try:
            if self.scaler is None:
                if self._one is None or self._one.device != loss.device:
                    self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
                loss.backward(gradient=self._one)          # (a cached 1: autograd would build one with a fill per step)
            else:
                loss.backward(gradient=self.scaler.scale_tensor)
        finally:
            if self.reserved_cus:
                _lib.load().sig_tune_reserved_cus(0)      # forward and the optimizer have the chip to themselves
        if self.reducer is not None:
            self.reducer.finish()
            if not self.fused:
                hip.flat.grad.mul_(1.0 / self.world)
        if self.scaler is None:
            self.optimizer.step()
        else:
            self.scaler.check(hip.flat.grad)
            self.optimizer.step(scaler=self.scaler)
            self.scaler.update()
        self.last_output = out
        return loss.detach().clone()         # (the step's scalars live in per-step scratch: the caller gets its own copy)
