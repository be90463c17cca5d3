"""Helpers that run the oracle (oracle/ffno_oracle.py) over golden_util inputs with autograd."""
import re

import numpy as np
import torch

import golden_util as gu
from oracle import ffno_oracle as orc

_DUP = re.compile(r"^spectral_layers\.\d+\.(fourier_weight\.\d+|(?:backcast|forecast)_ff\..*)$")


def canonical_key(key: str, sd_keys) -> str:
    """Name under which the reference's named_parameters() reports a (possibly shared) tensor."""
    m = _DUP.match(key)
    if m and m.group(1) in sd_keys:
        return m.group(1)
    return key


def torch_state_dict(sd_np, dtype=torch.float32, requires_grad=True):
    """numpy reference-layout state dict -> (aliased torch sd, {unique name: leaf tensor})."""
    uniq = {}
    sd = {}
    for k, v in sd_np.items():
        ck = canonical_key(k, sd_np.keys())
        if ck not in uniq:
            uniq[ck] = torch.tensor(sd_np[ck], dtype=dtype, requires_grad=requires_grad)
        sd[k] = uniq[ck]
    return sd, uniq


def oracle_block_run(kw, seed, B, M, N, dtype=torch.float32):
    kwf = gu.full_kwargs(kw)
    sd_np = gu.make_block_state_dict(kw, seed)
    x_np, t_np = gu.make_block_io(kw, seed, B, M, N)
    sd, uniq = torch_state_dict(sd_np, dtype)
    out = orc.ffno2d_block(sd, torch.tensor(x_np, dtype=dtype), modes=kwf["modes"], n_layers=kwf["n_layers"],
                           use_fork=kwf["use_fork"], mode=kwf["mode"], n_ff_layers=kwf["n_ff_layers"],
                           layer_norm=kwf["layer_norm"])
    loss = orc.lp_rel_loss(out["forecast"], torch.tensor(t_np, dtype=dtype))
    loss.backward()
    grads = {k: (p.grad.detach().numpy() if p.grad is not None else None) for k, p in uniq.items()}
    return out, loss, grads
