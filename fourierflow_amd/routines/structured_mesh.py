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
                 scheduler: Optional[dict] = None, **unused):
        super().__init__()
        reject_unsupported_routine_kwargs(unused)
        self.model, self.loss_scale = model, loss_scale
        self._opt_kw = dict(lr=1e-3, weight_decay=1e-4)
        self._opt_kw.update(optimizer or {})
        self._sch_kw = dict(num_warmup_steps=500, num_training_steps=100000, num_cycles=0.5)
        self._sch_kw.update(scheduler or {})
        self._trainer: Optional[FFNOTrainer] = None

    def trainer(self) -> FFNOTrainer:
        if self._trainer is None:
            self._trainer = FFNOTrainer(self.model, loss_scale=self.loss_scale, **self._opt_kw, **self._sch_kw)
        return self._trainer

    def training_step(self, batch, batch_idx: int = 0):
        return self.trainer().train_step(batch['x'], batch['y'])

    def validation_step(self, batch, batch_idx: int = 0):
        tr = self.trainer()
        pred = tr.predict(batch['x'])
        loss, _ = tr.loss_and_grad(pred, batch['y'].contiguous())
        return loss
