"""``fourier_encode`` -- mirror of ``fourierflow.modules.position.fourier_encode`` (reference position.py:6-31): sinusoidal
encodings of coordinates, ``[..., n_dims] -> [..., n_dims, 2 * num_bands + 1]`` (sin | cos of ``x * scale * pi`` for
log-spaced scales between 1 and ``max_freq / 2``, then the raw coordinate).  A constant of the grid, computed once per
geometry with a handful of element-wise torch ops on whatever device ``x`` lives on; not part of the HIP hot path."""
from math import log, pi

import torch


def fourier_encode(x, max_freq, num_bands=4, base=2):
    x = x.unsqueeze(-1)
    scales = torch.logspace(0., log(max_freq / 2) / log(base), num_bands, base=base, device=x.device, dtype=x.dtype)
    a = x * scales * pi
    return torch.cat([a.sin(), a.cos(), x], dim=-1)
