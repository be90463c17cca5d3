"""Normalizer -- mirror of ``fourierflow.modules.normalizer.Normalizer`` (reference normalizer.py:6-77):
running per-channel mean/std with the same buffers (``count``, ``n_accumulations``, ``sum``,
``sum_squared``, ``one``, ``std_epsilon``), so reference checkpoints load unchanged.

In the Markov routine the accumulate + normalise arithmetic runs inside the fused HIP feature kernel
(``ffno_markov_features``); this module owns the state and exposes ``forward`` / ``mean`` / ``std`` / ``inverse`` with
the reference's semantics, plus ``sync_across_ranks`` for data-parallel runs.
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
        # a checkpoint holds the statistics of the GLOBAL stream (every rank saved and loads the same synced sums): they are
        # the baseline of the next sync_across_ranks, not a local increment to be summed over ranks once more
        self._synced = self._sync_vector().clone()

    def _sync_vector(self) -> torch.Tensor:
        return torch.cat([self.sum, self.sum_squared, self.count.reshape(1)])

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)      # .to(device) / .cuda(): the sync baseline moves with the buffers
        if getattr(self, "_synced", None) is not None:
            self._synced = fn(self._synced)
        return out

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

    def sync_across_ranks(self, group=None):
        """Data parallel: make every rank hold the statistics of the GLOBAL batch stream.  Each rank accumulated its own
        shard since the last call; the increments are summed over ranks (one small all-reduce of {sum, sum_squared, count})
        so that all ranks normalise with identical mean / std and any rank's checkpoint holds the full statistics.
        n_accumulations counts accumulation CALLS and stays per-rank (identical on every rank by construction).
        The reference never ran this path under DDP (commands/train.py:83 reads a key its configs do not set); its
        DistributedDataParallel(broadcast_buffers=True) would have kept rank 0's shard statistics only."""
        dist = torch.distributed
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        D = self.sum.numel()
        base = getattr(self, "_synced", None)
        if base is None:          # nothing synced or loaded yet: everything this rank holds is its own contribution
            base = torch.zeros(2 * D + 1, dtype=self.sum.dtype, device=self.sum.device)
        elif base.device != self.sum.device:
            base = base.to(self.sum.device)
        cur = self._sync_vector()
        delta = cur - base                      # this rank's contribution since the last sync
        dist.all_reduce(delta, op=dist.ReduceOp.SUM, group=group)
        cur = base + delta
        self.sum.copy_(cur[:D])
        self.sum_squared.copy_(cur[D:2 * D])
        self.count.copy_(cur[2 * D])
        self._synced = cur.clone()

    def forward(self, x):
        """Stand-alone use (reference normalizer.py:28-43): accumulate while training, then (x - mean) / std over the last
        axis.  Runs the fused HIP statistics / normalise kernel of the Markov routine on the flattened [pixels, D] view."""
        import ctypes
        from .. import _capi, _lib
        from ..engine import _p
        _lib.require_device_tensor(x, "Normalizer input")
        D = self.sum.numel()
        if x.shape[-1] != D:
            raise ValueError(f"last axis must have {D} channels, got {tuple(x.shape)}")
        if D > 16:
            raise NotImplementedError("the fused statistics kernel holds up to 16 channels (the Markov routines use 3-6)")
        xc = x.contiguous()
        rows = xc.numel() // D
        acc = self.should_accumulate()
        state = self.pack_state()
        derived = torch.zeros(2 * D, dtype=torch.float32, device=x.device)
        partial = torch.empty(256 * 32, dtype=torch.float32, device=x.device)
        out = torch.empty_like(xc)
        # the feature kernel with no extra channels: use_position = 0 through the `extra` descriptor, [rows, 1, 1, D]
        extra = ctypes.byref(_capi.MarkovExtra(None, None, 0, 0))
        eps = float(self.std_epsilon.flatten()[0].item())
        rc = _lib.get_lib().ffno_markov_features(_p(xc), _p(state), _p(derived), None, _p(out), _p(partial), rows, 1, 1, D,
                                                 0.0, 1.0, 0.0, eps, int(acc), 1, extra, _lib.current_stream(x.device))
        _capi.check(rc, "normalizer")
        if acc:
            self.unpack_state(state)
            self._n_acc_host += 1.0
        return out.view(x.shape)
