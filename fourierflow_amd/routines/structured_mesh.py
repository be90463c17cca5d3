"""StructuredMeshExperiment -- counterpart of ``fourierflow.routines.StructuredMeshExperiment``
(reference routines/structured_mesh.py:8-31): plain x -> y regression of a mesh operator
(``FNOFactorizedMesh3D``) under the relative-L2 loss with the manual optimisation step of routines/base.py:27-52.
"""
from typing import Optional

import torch.nn as nn

from ..trainer import FFNOTrainer
from .checkpoint import CheckpointMixin, reject_unsupported_routine_kwargs


class StructuredMeshExperiment(CheckpointMixin, nn.Module):
    def __init__(self, model: nn.Module, loss_scale: float = 1.0, optimizer: Optional[dict] = None,
                 scheduler: Optional[dict] = None, optimizer_type: str = "adamw", **unused):
        super().__init__()
        reject_unsupported_routine_kwargs(unused)
        if optimizer_type not in ("adamw", "adam"):
            raise NotImplementedError(f"optimizer_type={optimizer_type!r}: the fused step implements AdamW and Adam")
        self.model, self.loss_scale, self.optimizer_type = model, loss_scale, optimizer_type
        self._opt_kw = dict(lr=1e-3, weight_decay=1e-4)
        self._opt_kw.update(optimizer or {})
        # the F-FNO configs run cosine-with-warm-up per step, the geo-FNO baselines (FNOMesh2D) torch.optim.Adam +
        # StepLR(step_size, gamma) per EPOCH -- recognised by its keyword arguments
        self._step_lr = scheduler is not None and "step_size" in scheduler
        self._sch_kw = dict(step_size=100, gamma=0.5) if self._step_lr else \
            dict(num_warmup_steps=500, num_training_steps=100000, num_cycles=0.5)
        self._sch_kw.update(scheduler or {})
        self.current_epoch = 0
        self._trainer: Optional[FFNOTrainer] = None

    def trainer(self) -> FFNOTrainer:
        if self._trainer is None:
            kw = dict(loss_scale=self.loss_scale, decoupled=self.optimizer_type == "adamw", **self._opt_kw)
            if self._step_lr:
                tr = FFNOTrainer(self.model, **kw)
                step, gamma = int(self._sch_kw["step_size"]), float(self._sch_kw["gamma"])
                tr.lr_factor = lambda: gamma ** (self.current_epoch // step)
                self._trainer = tr
            else:
                self._trainer = FFNOTrainer(self.model, **kw, **self._sch_kw)
        return self._trainer

    def on_train_epoch_end(self):
        self.current_epoch += 1

    def training_step(self, batch, batch_idx: int = 0):
        return self.trainer().train_step(batch['x'], batch['y'])

    def validation_step(self, batch, batch_idx: int = 0):
        tr = self.trainer()
        pred = tr.predict(batch['x'])
        loss, _ = tr.loss_and_grad(pred, batch['y'].contiguous())
        return loss
