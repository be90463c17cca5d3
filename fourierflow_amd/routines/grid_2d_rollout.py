"""Grid2DRolloutExperiment -- counterpart of ``fourierflow.routines.Grid2DRolloutExperiment`` (reference
routines/grid_2d_rollout.py:14-172), the caller of BASELINE config 0 (experiments/torus_li/zongyi/4_layers): the model
sees the last 10 vorticity fields + 2 position channels, predicts the next field, is fed its own prediction and
repeats for ``n_steps``; the training loss is the mean per-step relative-L2 loss, back-propagated through the whole
rollout (:104-134).

Every ``conv`` call and every loss term is a HIP pass (the module's autograd node returns the input gradient, so the
chain rule across steps is plain torch.autograd); parameter gradients of all steps accumulate inside the engine's flat
buffer (no per-parameter torch kernels), then one (all-reduced) fused AdamW launch applies them.  The reference runs
this routine under Lightning's automatic optimisation with AdamW + ``StepLR(step_size, gamma)`` per epoch
(zongyi/4_layers/config.yaml:27-43): call :meth:`on_train_epoch_end` once per epoch.
"""
from typing import Optional

import torch
import torch.nn as nn

from ..ops import lp_rel_loss
from ..trainer import FFNOTrainer
from .checkpoint import CheckpointMixin, reject_unsupported_routine_kwargs


class Grid2DRolloutExperiment(CheckpointMixin, nn.Module):
    def __init__(self, conv: nn.Module, n_steps: int, k_max: int = 32, num_freq_bands: int = 8, freq_base: int = 2,
                 use_fourier_position: bool = False, append_pos: bool = True, teacher_forcing: bool = False,
                 step_size: float = 1.0, optimizer: Optional[dict] = None, scheduler: Optional[dict] = None, **unused):
        super().__init__()
        reject_unsupported_routine_kwargs(unused)
        if use_fourier_position:
            raise NotImplementedError("use_fourier_position=True (fourier_encode features) is not used by any shipped "
                                      "config of this routine and is not implemented")
        self.conv, self.n_steps = conv, n_steps
        self.append_pos, self.teacher_forcing, self.step_size = append_pos, teacher_forcing, step_size
        self._opt_kw = dict(lr=2.5e-3, weight_decay=1e-4)
        self._opt_kw.update(optimizer or {})
        self._sch_kw = dict(step_size=100, gamma=0.5)            # StepLR of zongyi/4_layers/config.yaml:33-37
        self._sch_kw.update(scheduler or {})
        self.current_epoch = 0
        self._trainer: Optional[FFNOTrainer] = None

    # -- optimiser plumbing -------------------------------------------------------------------------
    def trainer(self) -> FFNOTrainer:
        if self._trainer is None:
            tr = FFNOTrainer(self.conv, **self._opt_kw)
            step, gamma = int(self._sch_kw["step_size"]), float(self._sch_kw["gamma"])
            tr.lr_factor = lambda: gamma ** (self.current_epoch // step)      # torch.optim.lr_scheduler.StepLR, per epoch
            self._trainer = tr
        return self._trainer

    def on_train_epoch_end(self):
        self.current_epoch += 1

    # -- the rollout ----------------------------------------------------------------------------------
    @staticmethod
    def _positions(B, X, Y, device):
        tx = torch.linspace(0, 1, X, device=device)
        ty = torch.linspace(0, 1, Y, device=device)       # the reference reuses the X ticks (:45,97): square grids only
        return torch.stack([tx[:, None].expand(X, Y), ty[None, :].expand(X, Y)], dim=-1)[None].expand(B, X, Y, 2)

    def forward(self, data):
        """data['data'] [B, X, Y, 10 + n_steps] -> the tuple of ``_learning_step`` (:38-50)."""
        xx = data['data'][..., :10]
        B, X, Y, _ = xx.shape
        xx = torch.cat([xx, self._positions(B, X, Y, xx.device)], dim=-1)
        return self._learning_step({'x': xx, 'y': data['data'][..., 10:]})

    def _learning_step(self, batch):
        xx, yy = batch['x'], batch['y']
        B, X, Y, _ = xx.shape
        if X != Y:
            raise ValueError("Grid2DRolloutExperiment: square grids only (the reference builds both position channels "
                             "from the X ticks, grid_2d_rollout.py:96-99)")
        pos_feats = self._positions(B, X, Y, xx.device)
        P = 2
        embeds = xx.contiguous()
        loss = 0
        step_losses, preds = [], []
        try:
            for t in range(self.n_steps):
                y = yy[..., t:t + 1].contiguous()
                self.conv.weights_frozen = t > 0      # the parameters do not change inside a rollout: pack them once
                im = self.conv(embeds)['forecast']
                l = lp_rel_loss(im, y)
                step_losses.append(l)
                loss = loss + l
                preds.append(im)
                if self.teacher_forcing and self.training:
                    im = y
                if self.append_pos:
                    embeds = torch.cat((embeds[..., 1:-P], im, pos_feats), dim=-1)
                else:
                    embeds = torch.cat((embeds[..., 1:], im), dim=-1)
        finally:
            self.conv.weights_frozen = False
        pred = torch.cat(preds, dim=-1)
        loss = loss / self.n_steps
        with torch.no_grad():
            yyc = yy[..., :self.n_steps].contiguous()
            loss_full = lp_rel_loss(pred.detach().contiguous(), yyc)
            pred_norm = torch.norm(pred, dim=[1, 2], keepdim=True)
            yy_norm = torch.norm(yyc, dim=[1, 2], keepdim=True)
            p = ((pred / pred_norm) * (yyc / yy_norm)).sum(dim=[1, 2]).mean(dim=0)
            has_diverged = p < 0.95
            diverged_idx = has_diverged.nonzero()
            diverged_t = diverged_idx[0, 0] if len(diverged_idx) > 0 else len(has_diverged)
            time_until = diverged_t * self.step_size
        return loss, loss_full, pred, step_losses, p, time_until

    # -- steps ------------------------------------------------------------------------------------------
    def training_step(self, batch, batch_idx: int = 0):
        """loss of the rollout, backward through every step, (all-reduce,) fused AdamW: one optimisation step."""
        tr = self.trainer()
        eng = tr.engine
        self.train()
        self.conv.fused_grad_accumulation = True      # parameter gradients of all n_steps passes add up inside the engine
        self.conv.max_live_passes = max(self.conv.max_live_passes, self.n_steps)
        eng.zero_grad()
        loss, loss_full, *_ = self._learning_step(batch)
        loss.backward()
        tr.apply_gradients(eng.gflat)
        return loss.detach(), loss_full

    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0):
        self.eval()
        loss, loss_full, preds, _, _, time_until = self._learning_step(batch)
        return {'valid_loss_avg': loss, 'valid_loss': loss_full, 'valid_time_until': time_until}

    @torch.no_grad()
    def test_step(self, batch, batch_idx: int = 0):
        self.eval()
        loss, loss_full, _, step_losses, p, time_until = self._learning_step(batch)
        return {'test_loss_avg': loss, 'test_loss': loss_full, 'test_time_until': time_until,
                'test_correlations': p, 'test_losses': torch.stack(step_losses)}
