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


def engine_relu_masks(engine):
    """The HIP path's ReLU active sets in the form ``orc.feedforward(relu_mask=...)`` takes: {(kind, layer): [bool tensor]}
    on the CPU.  Feeding them to the oracle removes the ReLU bit-flip discontinuity from a gradient comparison."""
    return {k: ([t.cpu().bool() for t in v] if isinstance(v, (list, tuple)) else [v.cpu().bool()])
            for k, v in engine.relu_active_sets().items()}


def oracle_block_run(kw, seed, B, M, N, dtype=torch.float32, relu_masks=None, io=None, sd_np=None, dropout_keeps=None):
    kwf = gu.full_kwargs(kw)
    sd_np = gu.make_block_state_dict(kw, seed) if sd_np is None else sd_np
    x_np, t_np = io if io is not None else gu.make_block_io(kw, seed, B, M, N)
    sd, uniq = torch_state_dict(sd_np, dtype)
    out = orc.ffno2d_block(sd, torch.tensor(x_np, dtype=dtype), modes=kwf["modes"], n_layers=kwf["n_layers"],
                           use_fork=kwf["use_fork"], mode=kwf["mode"], n_ff_layers=kwf["n_ff_layers"],
                           layer_norm=kwf["layer_norm"], relu_masks=relu_masks, dropout=kwf.get("dropout", 0.0),
                           in_dropout=kwf.get("in_dropout", 0.0), dropout_keeps=dropout_keeps)
    loss = orc.lp_rel_loss(out["forecast"], torch.tensor(t_np, dtype=dtype))
    loss.backward()
    grads = {k: (p.grad.detach().numpy() if p.grad is not None else None) for k, p in uniq.items()}
    return out, loss, grads


def relu_flips(kw, seed, B, M, N, masks, io=None, sd_np=None, return_out=False):
    """Number of hidden units on which the oracle's OWN fp32 ReLU decisions differ from the HIP path's active sets
    (``masks`` = engine_relu_masks(engine)), over every backcast feed-forward of the block, evaluated along the oracle's
    unmodified forward.  Zero flips = the two implementations sit on the same linear piece of the network, so their
    gradients may be compared at rounding level directly (no mask injection needed)."""
    orc.RELU_TRACE = trace = []
    try:
        with torch.no_grad():
            kwf = gu.full_kwargs(kw)
            sd_np = gu.make_block_state_dict(kw, seed) if sd_np is None else sd_np
            x_np, _ = io if io is not None else gu.make_block_io(kw, seed, B, M, N)
            sd, _ = torch_state_dict(sd_np, torch.float32, requires_grad=False)
            plain = orc.ffno2d_block(sd, torch.tensor(x_np), modes=kwf["modes"], n_layers=kwf["n_layers"], use_fork=kwf["use_fork"],
                                     mode=kwf["mode"], n_ff_layers=kwf["n_ff_layers"], layer_norm=kwf["layer_norm"])
    finally:
        orc.RELU_TRACE = None
    flips, layer = 0, {"backcast": 0, "forecast": 0}
    for prefix, i, active in trace:
        kind = "forecast" if "forecast_ff" in prefix else "backcast" if "backcast_ff" in prefix else None
        if kind is None or i != 0:
            continue
        key = (kind, layer[kind])
        layer[kind] += 1
        if kwf["use_fork"] and key == ("backcast", kwf["n_layers"] - 1):
            continue        # with fork heads the last backcast only feeds the dead x_L: the HIP path does not evaluate it
        if key in masks:
            flips += int((active.reshape(-1) != masks[key][0].reshape(-1)).sum())
    if return_out:      # (the oracle's own, unmasked forward: callers that also want it do not run the oracle a third time)
        return flips, plain
    return flips


GRAD_TOL = 5e-5


def check_grads_at_rounding_level(label, grads, run_oracle, tol=GRAD_TOL):
    """Every parameter gradient of the HIP path against the oracle evaluated on the SAME ReLU active sets.

    ``run_oracle(dtype)`` -> {name: gradient} (the caller binds the masks).  The bar is ``tol`` (5e-5) relative L2 on every
    parameter.  A few gradients are sums with heavy cancellation (the head / fork biases: sum over all pixels of a signed,
    near-zero-mean field) where two correct fp32 evaluations differ by more than that; for those the bar is relative to
    the reference op sequence's OWN fp32 rounding noise, measured as |oracle(fp32) - oracle(fp64)|: the HIP result must be
    within max(tol, 4 x that noise) of the fp64 value.  Returns (worst name, worst error); prints the observed errors."""
    g32 = run_oracle(torch.float32)
    errs = {n: _rel(grads[n], g32[n]) for n in grads}
    bad = [n for n, e in errs.items() if e >= tol]
    if bad:
        g64 = run_oracle(torch.float64)
        for n in bad:
            noise = _rel(g32[n], g64[n])
            e64 = _rel(grads[n], g64[n])
            print(f"[{label}] {n}: vs fp32 oracle {errs[n]:.2e}, vs fp64 oracle {e64:.2e}, fp32 oracle's own noise {noise:.2e}")
            assert e64 < max(tol, 4 * noise), (n, e64, noise)
            errs[n] = min(errs[n], e64)
    worst = max(errs, key=errs.get)
    print(f"[{label}] worst gradient vs oracle on the same active sets: {errs[worst]:.2e} ({worst}); "
          f"median {float(np.median(list(errs.values()))):.2e}; {len(bad)} cancellation-limited")
    return worst, errs[worst]


def _rel(a, b):
    if b is None:        # parameter the loss does not depend on (e.g. the last backcast_ff with fork heads): must be zero
        return float(np.linalg.norm(np.asarray(a, np.float64)))
    if a is None:
        return float(np.linalg.norm(np.asarray(b, np.float64)))
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
