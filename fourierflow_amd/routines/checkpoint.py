"""Checkpoint / resume for the routines -- counterpart of the reference's Lightning checkpoints
(``Routine.load_lightning_model_state``, reference routines/base.py:79-102; files written by
callbacks/model_checkpoint.py ``CustomModelCheckpoint`` and resumed by commands/train.py:74-80,116).

A checkpoint is a ``torch.save``d dict in the Lightning layout the reference reads:
``{'state_dict': routine.state_dict(), 'epoch', 'global_step', 'optimizer_states', 'lr_schedulers'}``.
The routine mirrors keep the reference's attribute names (``conv.*`` / ``model.*``, ``normalizer.*``, ``_float``), so the
``state_dict`` of a checkpoint written by the reference loads here with ``strict=True`` and vice versa.  The optimiser
part holds this framework's flat AdamW state (first/second moments of the flat parameter buffer + step count).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

# buffers of the reference's velocity / super-resolution code paths that its loader drops (routines/base.py:89-99)
REMOVE_KEYS = ['kx', 'ky', 'lap', 'kx_32', 'ky_32', 'lap_32', 'kx_64', 'ky_64', 'lap_64', 'kx_128', 'ky_128', 'lap_128',
               'kx_256', 'ky_256', 'lap_256']


def reject_unsupported_routine_kwargs(kwargs: Dict[str, Any]) -> None:
    """The routines accept (and ignore) the reference Routine's logging / Lightning keyword arguments, but options that
    would change the optimisation (routines/base.py:27-52) and are not built must fail loudly, not silently."""
    if kwargs.get("clip_val") is not None:
        raise NotImplementedError("clip_val (torch.nn.utils.clip_grad_value_) is not folded into the fused optimiser step; "
                                  "no shipped F-FNO config sets it")
    if kwargs.get("accumulate_grad_batches", 1) not in (None, 1):
        raise NotImplementedError("accumulate_grad_batches > 1 is not implemented (no shipped config uses it)")


class CheckpointMixin:
    """Needs ``self.trainer() -> FFNOTrainer`` and to be an ``nn.Module``."""

    def checkpoint_dict(self, epoch: int = 0, global_step: Optional[int] = None) -> Dict[str, Any]:
        tr = self.trainer()
        return {
            'epoch': int(epoch),
            'global_step': int(tr.step_count if global_step is None else global_step),
            'pytorch-lightning_version': 'ffno-mi355x',
            'state_dict': {k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
            'optimizer_states': [{'flat_adamw': True, 'exp_avg': tr.m.cpu().clone(), 'exp_avg_sq': tr.v.cpu().clone(),
                                  'step': int(tr.opt_step), 'param_names': list(tr.engine.param_names)}],
            'lr_schedulers': [{'last_epoch': int(tr.step_count)}],
        }

    def save_checkpoint(self, path: str, epoch: int = 0, global_step: Optional[int] = None) -> None:
        torch.save(self.checkpoint_dict(epoch, global_step), path)

    def load_lightning_model_state(self, checkpoint_path, map_location=None) -> None:
        """Model (and normaliser) weights only, like the reference before testing (commands/train.py:125-130)."""
        ckpt = torch.load(checkpoint_path, map_location=map_location or 'cpu', weights_only=False)
        state_dict = dict(ckpt['state_dict'])
        strict = True
        for key in REMOVE_KEYS:
            if key in state_dict:
                del state_dict[key]
                strict = False
        self.load_state_dict(state_dict, strict=strict)

    def resume_from_checkpoint(self, checkpoint_path) -> Dict[str, Any]:
        """Weights + optimiser moments + schedule position (commands/train.py:74-80: ``last.ckpt``)."""
        self.load_lightning_model_state(checkpoint_path)
        ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        tr = self.trainer()
        opt = (ckpt.get('optimizer_states') or [None])[0]
        if opt is not None and opt.get('flat_adamw'):
            if list(opt['param_names']) != list(tr.engine.param_names):
                raise ValueError("checkpoint optimiser state belongs to a different parameter layout")
            tr.m.copy_(opt['exp_avg'])
            tr.v.copy_(opt['exp_avg_sq'])
            tr.opt_step = int(opt['step'])
            tr.step_count = int((ckpt.get('lr_schedulers') or [{}])[0].get('last_epoch', opt['step']))
        elif opt is not None and self._load_reference_optimizer_state(tr, opt):
            # a checkpoint written by the reference (torch.optim.AdamW.state_dict()): per-parameter moments, converted
            tr.step_count = int((ckpt.get('lr_schedulers') or [{}])[0].get('last_epoch', ckpt.get('global_step', 0)))
        else:
            # no usable optimiser state: restart the moments AND the bias correction (a bias-correction step far from 0
            # with zero moments would scale the first updates by ~1/(1-beta1) / sqrt(1/(1-beta2)) ~ 3x lr); the schedule
            # keeps its position
            tr.m.zero_()
            tr.v.zero_()
            tr.opt_step = 0
            tr.step_count = int(ckpt.get('global_step', 0))
        return {'epoch': int(ckpt.get('epoch', 0)), 'global_step': int(ckpt.get('global_step', 0))}

    def _load_reference_optimizer_state(self, tr, opt) -> bool:
        """torch.optim.Adam(W).state_dict() -> the flat moment buffers.  Parameter i of the optimiser is the i-th entry of
        the routine's ``parameters()`` (routines/base.py:60-66 passes ``self.parameters()``), whose order the module
        mirrors reproduce (same registration order, shared tensors reported once).  Returns False if the state does not fit."""
        state = opt.get('state') if isinstance(opt, dict) else None
        groups = opt.get('param_groups') if isinstance(opt, dict) else None
        if not state or not groups:
            return False
        names = [n for n, _ in self.named_parameters()]
        idx = [i for g in groups for i in g['params']]
        if len(idx) != len(names):
            return False
        eng = tr.engine
        prefix = next((p for p in ("conv.", "model.") if any(n.startswith(p) for n in names)), "")
        step = None
        tr.m.zero_()
        tr.v.zero_()
        for i, name in zip(idx, names):
            short = name[len(prefix):] if name.startswith(prefix) else name
            if short not in eng._offsets:       # a parameter the engine does not train (unused Fourier weights of mode != 'full')
                continue
            st = state.get(i)
            if st is None:
                continue
            o, shape = eng._offsets[short], eng.param_shapes[short]
            cnt = 1
            for d in shape:
                cnt *= d
            if tuple(st['exp_avg'].shape) != tuple(shape):
                import warnings
                warnings.warn(f"optimiser state of parameter {i} ({tuple(st['exp_avg'].shape)}) does not fit {short} "
                              f"{tuple(shape)}: the Adam moments are restarted", RuntimeWarning)
                return False
            tr.m[o:o + cnt].copy_(st['exp_avg'].reshape(-1))
            tr.v[o:o + cnt].copy_(st['exp_avg_sq'].reshape(-1))
            step = int(st['step']) if step is None else step
        if step is None:
            return False
        tr.opt_step = step
        return True
