"""Normalizer -- mirror of ``fourierflow.modules.normalizer.Normalizer`` (reference normalizer.py:6-77):
running per-channel mean/std with the same buffers (``count``, ``n_accumulations``, ``sum``,
``sum_squared``, ``one``, ``std_epsilon``), so reference checkpoints load unchanged.

In the Markov routine the accumulate + normalise arithmetic runs inside the fused HIP feature kernel
(``ffno_markov_features``); this module owns the state and exposes ``mean`` / ``std`` / ``inverse`` with
the reference's semantics.
"""
import torch
import torch.nn as nn


class Normalizer(nn.Module):
    def __init__(self, size, max_accumulations=10**6, std_epsilon=1e-8):
        super().__init__()
        self.max_accumulations = max_accumulations
        self.register_buffer('count', torch.tensor(0.0))
        self.register_buffer('n_accumulations', torch.tensor(0.0))
        self.register_buffer('sum', torch.full(size, 0.0))
        self.register_buffer('sum_squared', torch.full(size, 0.0))
        self.register_buffer('one', torch.tensor(1.0))
        self.register_buffer('std_epsilon', torch.full(size, std_epsilon))
        self._n_acc_host = 0.0   # host mirror of n_accumulations (avoids a device sync per step)

    # -- packed state for the fused kernel: {sum[D], sum_squared[D], count, n_accumulations} --------------
    def pack_state(self) -> torch.Tensor:
        return torch.cat([self.sum, self.sum_squared, self.count.reshape(1), self.n_accumulations.reshape(1)]).contiguous()

    def unpack_state(self, state: torch.Tensor):
        D = self.sum.numel()
        self.sum.copy_(state[:D])
        self.sum_squared.copy_(state[D:2 * D])
        self.count.copy_(state[2 * D])
        self.n_accumulations.copy_(state[2 * D + 1])

    def should_accumulate(self) -> bool:
        return self.training and self._n_acc_host < self.max_accumulations

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._n_acc_host = float(self.n_accumulations.item())

    @property
    def mean(self):
        safe_count = torch.maximum(self.count, self.one)
        return self.sum / safe_count

    @property
    def std(self):
        safe_count = torch.maximum(self.count, self.one)
        std = torch.sqrt(self.sum_squared / safe_count - self.mean**2)
        return torch.maximum(std, self.std_epsilon)

    def inverse(self, x, channel=None):
        if channel is None:
            return x * self.std + self.mean
        return x * self.std[channel] + self.mean[channel]

    def forward(self, x):  # pragma: no cover - deliberate
        raise NotImplementedError("Normalizer.forward is fused into the HIP feature kernel of the Markov routine "
                                  "(fourierflow_amd.routines.Grid2DMarkovExperiment._build_features)")
