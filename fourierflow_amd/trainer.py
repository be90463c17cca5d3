"""Data-parallel training step of the F-FNO block: one process per GPU, flat buffers.

Counterpart of the reference's manual-optimisation step
(fourierflow/routines/base.py:27-52 ``optimize_manually`` + grid_2d_markov.py:172-193
``_training_step``: conv -> LpLoss.rel -> zero_grad/backward/AdamW.step/scheduler.step) with the
optimiser and schedule of experiments/torus_li/markov/24_layers/config.yaml:36-47, and of Lightning's
DDPPlugin (commands/train.py:83-84), re-designed for MI355X:

  * every parameter is a view into ONE flat fp32 buffer, every gradient lands in ONE flat buffer
    (written directly by the HIP backward kernels),
  * data parallelism = ONE ``all_reduce`` of that flat gradient buffer per step over RCCL/xGMI
    (4.3 MB for markov/24 -- latency-bound, so a single collective beats DDP's bucketed reducer),
    the 1/world scaling is folded into the optimiser kernel,
  * AdamW + cosine-with-warmup is ONE fused kernel launch over the flat buffers.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _capi, _lib
from .engine import _p


def cosine_warmup_factor(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    """LambdaLR multiplier of the reference's CosineWithWarmupScheduler (schedulers/cosine_with_warmup.py:6-17)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class FFNOTrainer:
    def __init__(self, block, *, lr: float = 2.5e-3, weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 num_warmup_steps: int = 500, num_training_steps: int = 100000, num_cycles: float = 0.5,
                 process_group=None, broadcast_from_rank0: bool = True, loss_scale: float = 1.0, decoupled: bool = True):
        self.block = block
        self.engine = block.engine()
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.sched = (num_warmup_steps, num_training_steps, num_cycles)
        self.decoupled = decoupled     # True: torch.optim.AdamW; False: torch.optim.Adam (L2 weight decay)
        self.lr_factor = None          # optional callable() -> multiplier replacing the cosine schedule (StepLR per epoch)
        self.step_count = 0            # schedule position (LambdaLR's last_epoch = completed optimiser steps)
        self.opt_step = 0              # Adam bias-correction step; differs from step_count only after a resume that had to
                                       # restart the moments (routines/checkpoint.py)
        self.loss_scale = loss_scale   # StructuredMeshExperiment: gradients of loss * loss_scale (structured_mesh.py:29)
        self.pg = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        named = block.engine_parameters()
        dev = named[0][1].device
        self.device = dev
        n = self.engine.n_params
        self.pflat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        views = {}
        with torch.no_grad():
            for name, p in named:
                cnt = p.numel()
                v = self.pflat[off:off + cnt].view(p.shape)
                v.copy_(p)
                p.data = v  # the module's parameters now alias the flat buffer
                # What the engine binds is ``p.detach()``, not ``v``: ``p.data = v`` keeps p's OWN version counter, so an
                # in-place write through the Parameter (load_state_dict, p.copy_(), an external optimizer.step()) bumps
                # p._version only -- a bound ``v`` would never see it and the engine would keep stale derived operands
                # (weight-norm products, packed fragments, folded head).  detach() shares p's counter.
                views[name] = p.detach()
                off += cnt
        assert off == n
        if self.world > 1 and broadcast_from_rank0:
            torch.distributed.broadcast(self.pflat, src=0, group=process_group)
            self.engine.weights_changed()      # (written through the flat buffer, whose counter the engine does not watch)
        self.engine.bind(views)
        self.m = torch.zeros_like(self.pflat)
        self.v = torch.zeros_like(self.pflat)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self._gy = None
        self._tmp = None

    def current_lr(self) -> float:
        """lr used by the NEXT optimiser step (LambdaLR semantics: factor(number of completed steps))."""
        if self.lr_factor is not None:
            return self.lr * float(self.lr_factor())
        return self.lr * cosine_warmup_factor(self.step_count, *self.sched)

    def loss_and_grad(self, pred: torch.Tensor, target: torch.Tensor, affine: Optional[torch.Tensor] = None, fresh_loss: bool = False):
        lib = _lib.get_lib()
        B = pred.shape[0]
        n = pred.numel() // B
        if self._gy is None or self._gy.shape != pred.shape:
            self._gy = torch.empty_like(pred)
            self._tmp = torch.empty(int(lib.ffno_lploss_tmp_floats(B, n)), dtype=torch.float32, device=pred.device)
        if fresh_loss:
            # the kernel writes the loss into a NEW 4-byte tensor (a host-side allocation from the caching allocator, no launch):
            # the caller owns it -- no clone of a persistent buffer afterwards
            self.loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        _capi.check(lib.ffno_lploss_fwd_bwd(_p(pred), _p(target), _p(self.loss), _p(self._gy), _p(self._tmp), B, n, float(self.loss_scale),
                                            _p(affine), _lib.current_stream(self.device)), "lploss")
        return self.loss, self._gy     # gy: a persistent buffer, overwritten by the next call; loss: likewise unless fresh_loss

    def train_step(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """forward + relative-L2 loss + backward + (all-reduce) + AdamW + schedule; returns the loss (device)."""
        lib = _lib.get_lib()
        target = target.contiguous()
        # (the prediction is consumed by the loss kernel at once: a view of the engine's output buffer, not a copy; the loss lands in
        #  a tensor of its own: two copy launches less per step than forward() + loss.clone())
        if getattr(self.engine, "can_return_view", False):
            pred = self.engine.forward(self.block.prepare_input(x), True, own_output=False)
        else:
            pred = self.engine.forward(self.block.prepare_input(x), True)
        loss, gy = self.loss_and_grad(pred, target, fresh_loss=True)
        return self.apply_gradients(self.engine.backward(gy), loss, loss_is_fresh=True)

    def apply_gradients(self, gflat: torch.Tensor, loss: Optional[torch.Tensor] = None, loss_is_fresh: bool = False):
        """(all-reduce) + fused AdamW/cosine step on the flat buffers."""
        lib = _lib.get_lib()
        if self.world > 1:
            # ONE collective per step, enqueued behind the backward kernels (stream-ordered by ProcessGroupNCCL).  Not split
            # or overlapped on purpose: the buffer is complete only after the last backward kernel (weight-norm backward
            # reads the reduced feed-forward gradients), 4.3 MB over xGMI is ~0.1-0.2 ms of an ~7 ms step, and a second
            # collective would add its own launch latency.
            ev = getattr(self, "comm_events", None)      # optional: a list that receives (start, stop) device events (bench.py)
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            torch.distributed.all_reduce(gflat, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            if ev is not None:
                e1.record()       # (the launch stream waits for the collective's stream: the pair brackets it)
                ev.append((e0, e1))
        lr_t = self.current_lr()
        self.step_count += 1
        self.opt_step += 1
        # a fresh 4-byte tensor per step: the reference returns a new loss tensor each step and callers stack them
        loss_out = (loss if loss_is_fresh else loss.clone()) if loss is not None else None
        fn = lib.ffno_adamw_flat if self.decoupled else lib.ffno_adam_flat
        _capi.check(fn(_p(self.pflat), _p(gflat), _p(self.m), _p(self.v), self.pflat.numel(), lr_t, self.betas[0],
                       self.betas[1], self.eps, self.wd, self.opt_step, 1.0 / self.world,
                       _lib.current_stream(self.device)), "adamw" if self.decoupled else "adam")
        if hasattr(self.engine, "weights_changed"):
            self.engine.weights_changed()      # (the kernel wrote the parameters through a raw pointer)
        return loss_out

    @torch.no_grad()
    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return self.engine.forward(self.block.prepare_input(x), False)
