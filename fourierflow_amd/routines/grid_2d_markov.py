"""Markov one-step training routine around the F-FNO block -- counterpart of
``fourierflow.routines.Grid2DMarkovExperiment`` (reference routines/grid_2d_markov.py:23-193, 374-390)
without Lightning / wandb / jax: feature build (positional channels, running normaliser, Gaussian noise),
the operator, inverse-normalise + relative-L2 loss, the manual optimisation step
(routines/base.py:27-52), epoch-0 statistics accumulation, and the autoregressive rollout of
``_valid_step`` (:195-326) used by predict/infer.

Everything on the device is a HIP kernel of libffno_hip.so; torch supplies memory, the stream and the
Gaussian noise samples.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _capi, _lib
from ..engine import _p
from ..modules.normalizer import Normalizer
from ..trainer import FFNOTrainer
from .checkpoint import CheckpointMixin, reject_unsupported_routine_kwargs


class Grid2DMarkovExperiment(CheckpointMixin, nn.Module):
    def __init__(self, conv: nn.Module, n_steps: Optional[int] = None, low: float = 0, high: float = 1,
                 use_position: bool = True, append_force: bool = False, append_mu: bool = False,
                 max_accumulations: float = 1e6, should_normalize: bool = True, use_fourier_position: bool = False,
                 noise_std: float = 0.0, shuffle_grid: bool = False, use_velocity: bool = False,
                 learn_difference: bool = False, optimizer: Optional[dict] = None, scheduler: Optional[dict] = None,
                 domain=((0.0, 2 * math.pi), (0.0, 2 * math.pi)), grid_size=(64,), **unused):
        super().__init__()
        reject_unsupported_routine_kwargs(unused)
        if use_fourier_position:
            raise NotImplementedError("use_fourier_position=True is not built: the reference's own Markov routine cannot run it "
                                      "either (encode_positions reads self.k_max, which its constructor never sets: "
                                      "grid_2d_markov.py:23-45,116)")
        self.conv = conv
        self.shuffle_grid = bool(shuffle_grid)
        if self.shuffle_grid:       # torus_li/ablation/shuffle_xy_grid: fixed random row / column permutations around the model
            assert len(grid_size) == 1, 'shuffle_grid only supports one size'         # grid_2d_markov.py:75-80
            for name, idx in (("x", torch.randperm(grid_size[0])), ("y", torch.randperm(grid_size[0]))):
                self.register_buffer(f"_{name}_idx", idx, persistent=False)            # plain attributes in the reference:
                self.register_buffer(f"_{name}_inv", torch.argsort(idx), persistent=False)   # not part of the state_dict
        self.use_position, self.append_force, self.append_mu = bool(use_position), bool(append_force), bool(append_mu)
        self.n_steps, self.low, self.high = n_steps, low, high
        self.should_normalize, self.noise_std, self.learn_difference = should_normalize, noise_std, learn_difference
        self.normalizer = Normalizer([conv.input_dim], max_accumulations)
        self.register_buffer('_float', torch.FloatTensor([0.1]))
        self.use_velocity, self.domain = use_velocity, tuple(tuple(float(v) for v in d) for d in domain)
        if use_velocity:
            # same buffers as the reference (grid_2d_markov.py:82-94) so its checkpoints load strictly; the HIP kernel
            # derives the wavenumbers from the domain lengths itself
            for size in grid_size:
                lx, ly = self.domain[0][1] - self.domain[0][0], self.domain[1][1] - self.domain[1][0]
                kx, ky = np.meshgrid(np.fft.fftfreq(size, d=lx / size), np.fft.rfftfreq(size, d=ly / size), indexing="ij")
                lap = (2 * np.pi * 1j) ** 2 * (np.abs(kx) ** 2 + np.abs(ky) ** 2)
                lap[0, 0] = 1
                self.register_buffer(f'kx_{size}', torch.from_numpy(kx.astype(np.float32)))
                self.register_buffer(f'ky_{size}', torch.from_numpy(ky.astype(np.float32)))
                self.register_buffer(f'lap_{size}', torch.from_numpy(lap.astype(np.complex64)))
        self._vel_ws = None
        self._opt_kw = dict(lr=2.5e-3, weight_decay=1e-4)
        self._opt_kw.update(optimizer or {})
        # the F-FNO configs run cosine-with-warm-up per step; the FNOZongyi2DBlock ablations (torus_li/ablation/zongyi_markov*)
        # run torch.optim.lr_scheduler.StepLR(step_size, gamma) per EPOCH -- recognised by its keyword arguments
        self._step_lr = scheduler is not None and "step_size" in scheduler
        self._sch_kw = dict(step_size=100, gamma=0.5) if self._step_lr else \
            dict(num_warmup_steps=500, num_training_steps=100000, num_cycles=0.5)
        self._sch_kw.update(scheduler or {})
        self.current_epoch = 0
        self._trainer: Optional[FFNOTrainer] = None
        self._derived = None
        self._partial = None
        self._affine = None

    # ----------------------------------------------------------------------------------------------
    def trainer(self) -> FFNOTrainer:
        if self._trainer is None:
            if self._step_lr:
                tr = FFNOTrainer(self.conv, **self._opt_kw)
                step, gamma = int(self._sch_kw["step_size"]), float(self._sch_kw["gamma"])
                tr.lr_factor = lambda: gamma ** (self.current_epoch // step)
                self._trainer = tr
            else:
                self._trainer = FFNOTrainer(self.conv, **self._opt_kw, **self._sch_kw)
        return self._trainer

    def _build_features(self, batch: Dict[str, torch.Tensor], noise: Optional[torch.Tensor] = None,
                        add_noise: bool = True) -> torch.Tensor:
        """x [B, M, N, Cx] -> normalised features [B, M, N, Cx + 2] (+ noise), accumulating the running
        statistics while training (grid_2d_markov.py:124-170, normalizer.py:45-55)."""
        x = batch['x'].contiguous()
        _lib.require_device_tensor(x, "batch['x']")
        if self.use_velocity:       # [B, M, N, 1] vorticity -> [B, M, N, 3] (vorticity, u, v)  (grid_2d_markov.py:130-144)
            x = self._velocity(x)
        B, M, N, Cx = x.shape
        extra, keep = None, []
        if not self.use_position or self.append_force or self.append_mu:      # grid_2d_markov.py:146-162
            force = batch['f'].contiguous().float() if self.append_force else None
            mu = batch['mu'].contiguous().float() if self.append_mu else None
            for t, shape in ((force, (B, M, N)), (mu, (B,))):
                if t is not None:
                    _lib.require_device_tensor(t, "batch['f'] / batch['mu']")
                    if tuple(t.shape) != shape:
                        raise ValueError(f"expected a tensor of shape {shape}, got {tuple(t.shape)}")
            keep = [force, mu]
            extra = ctypes.byref(_capi.MarkovExtra(_p(force), _p(mu), int(self.use_position), 0))
        D = Cx + (2 if self.use_position else 0) + int(self.append_force) + int(self.append_mu)
        if D != self.conv.input_dim:
            raise ValueError(f"conv.input_dim={self.conv.input_dim} but the features have {D} channels")
        dev = x.device
        if self._derived is None or self._derived.device != dev:
            self._derived = torch.zeros(2 * D, dtype=torch.float32, device=dev)
            self._partial = torch.empty(256 * 32, dtype=torch.float32, device=dev)
        nz = self.normalizer
        acc = self.should_normalize and nz.should_accumulate()
        state = nz.pack_state()
        if not add_noise:
            noise = None
        elif noise is None and self.noise_std:
            noise = torch.randn(B, M, N, D, device=dev)      # `x += randn * noise_std` (grid_2d_markov.py:168)
        out = torch.empty(B, M, N, D, dtype=torch.float32, device=dev)

        def features(state, accumulate):
            rc = _lib.get_lib().ffno_markov_features(_p(x), _p(state), _p(self._derived), _p(noise), _p(out), _p(self._partial),
                                                     B, M, N, Cx, float(self.low), float(self.high), float(self.noise_std),
                                                     self._eps(), int(accumulate),
                                                     int(self.should_normalize), extra, _lib.current_stream(dev))
            _capi.check(rc, "markov_features")

        features(state, acc)
        if acc:
            nz.unpack_state(state)
            nz._n_acc_host += 1.0
            # data parallel: every rank accumulated its own shard.  Sum the increments over ranks FIRST, then normalise this very
            # batch once more with the statistics of the GLOBAL batch stream (a second pass over 1.5 MB), so that all ranks
            # normalise every step -- this one included -- with identical mean / std, the loss's inverse affine (read from
            # `_derived`) is the global one as well, and any rank's checkpoint holds the full statistics.
            if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                    torch.distributed.get_world_size() > 1:
                nz.sync_across_ranks()
                features(nz.pack_state(), False)
        del keep
        return out

    def _velocity(self, x: torch.Tensor) -> torch.Tensor:
        B, M, N, Cx = x.shape
        if Cx != 1:
            raise ValueError("use_velocity expects the single-channel vorticity field [B, M, N, 1]")
        lib = _lib.get_lib()
        need = int(lib.ffno_velocity_ws_floats(B, M, N))
        if self._vel_ws is None or self._vel_ws.numel() < need or self._vel_ws.device != x.device:
            self._vel_ws = torch.empty(need, dtype=torch.float32, device=x.device)
        out = torch.empty(B, M, N, 3, dtype=torch.float32, device=x.device)
        lx, ly = self.domain[0][1] - self.domain[0][0], self.domain[1][1] - self.domain[1][0]
        _capi.check(lib.ffno_velocity_features(_p(x), _p(out), _p(self._vel_ws), B, M, N, lx, ly,
                                               _lib.current_stream(x.device)), "velocity_features")
        return out

    def _eps(self) -> float:
        if not hasattr(self, "_eps_host"):
            self._eps_host = float(self.normalizer.std_epsilon.flatten()[0].item())   # one sync, at first use
        return self._eps_host

    def _affine_tensor(self):
        """{std[0], mean[0]} for Normalizer.inverse(channel=0) fused into the loss kernel."""
        if not self.should_normalize:
            return None
        D = self.conv.input_dim
        self._affine = torch.stack([self._derived[D], self._derived[0]]).contiguous()
        return self._affine

    def _training_step(self, batch, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """features -> conv -> inverse-normalise -> LpLoss.rel, then the manual optimisation step."""
        tr = self.trainer()
        feats = self._build_features(batch, noise)
        targets = (batch['dy'] if self.learn_difference else batch['y']).contiguous()
        pred = self._unshuffle(tr.engine.forward(self._shuffle(feats), True))
        loss, gy = tr.loss_and_grad(pred, targets, self._affine_tensor())
        tr.apply_gradients(tr.engine.backward(self._shuffle(gy)))      # adjoint of the inverse gather = the forward gather
        return loss

    def _shuffle(self, t: torch.Tensor) -> torch.Tensor:
        """x[:, x_idx][:, :, y_idx] (grid_2d_markov.py:177-178)."""
        if not self.shuffle_grid:
            return t
        return t.index_select(1, self._x_idx).index_select(2, self._y_idx).contiguous()

    def _unshuffle(self, t: torch.Tensor) -> torch.Tensor:
        """im[:, :, y_inv][:, x_inv] (grid_2d_markov.py:182-183)."""
        if not self.shuffle_grid:
            return t
        return t.index_select(2, self._y_inv).index_select(1, self._x_inv).contiguous()

    def training_step(self, batch, epoch: int, noise: Optional[torch.Tensor] = None):
        """Epoch 0 only accumulates the normaliser statistics (grid_2d_markov.py:376-378); later epochs train."""
        self.current_epoch = int(epoch)
        if self.should_normalize and epoch < 1:
            with torch.no_grad():
                self._build_features(batch, noise)
            return None
        return self._training_step(batch, noise)

    @torch.no_grad()
    def rollout(self, x0: torch.Tensor, n_steps: Optional[int] = None, f: Optional[torch.Tensor] = None,
                mu: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Autoregressive inference (grid_2d_markov.py:263-321): feed each de-normalised prediction back as the
        next input (the force map ``f`` [B, M, N] and viscosity ``mu`` [B] stay fixed).  x0 [B, M, N, 1] -> [B, M, N, n_steps]."""
        was_training = self.normalizer.training
        self.normalizer.eval()
        tr = self.trainer()
        preds, x, prev = [], x0, x0
        for _ in range(n_steps or self.n_steps or 1):
            feats = self._build_features({'x': x, 'f': f, 'mu': mu}, add_noise=False)   # no noise at validation (:292-293)
            im = self._unshuffle(tr.engine.forward(self._shuffle(feats), False))      # :297-304
            if self.should_normalize:
                D = self.conv.input_dim
                im = im * self._derived[D] + self._derived[0]
            if self.learn_difference:
                im = prev + im
                prev = im
            preds.append(im)
            x = im
        self.normalizer.train(was_training)
        return torch.cat(preds, dim=-1)
