"""CPU oracle for the F-FNO hot path (TEST INFRASTRUCTURE -- not a product path).

This file is a *restatement*, in plain CPU torch ops, of the algorithm the
reference implements in

  fourierflow/modules/factorized_fno/grid_2d.py:51-99   (SpectralConv2d.forward_fourier)
  fourierflow/modules/factorized_fno/grid_2d.py:42-49   (SpectralConv2d.forward)
  fourierflow/modules/factorized_fno/grid_2d.py:154-177 (FNOFactorized2DBlock.forward)
  fourierflow/modules/feedforward.py:6-24               (FeedForward)
  fourierflow/modules/linear.py:41-52                   (WNLinear = nn.Linear + weight_norm dim=0)
  fourierflow/modules/loss.py:33-46                     (LpLoss.rel)

It is written functionally over a reference-compatible ``state_dict`` (same key
names / shapes as the reference module) so the golden
vectors under tests/golden/ -- produced by importing the real reference in the
build container with tools/make_golden.py -- pin it (tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The shipped package (fourierflow_amd/) never does.

Also restated here, with their own golden vectors: FNOFactorizedMesh2D / Mesh3D (mesh_2d.py, mesh_3d.py),
FNOPlus2DBlock (zongyi_fno/grid_plus_2d.py), CNOFactorized2DBlock / Mesh2D / Mesh3D (factorized_cno/*.py, modules/dct.py), FNOZongyi2DBlock (zongyi_fno/grid_2d.py), FNOMesh2D / FNOMesh3D
(zongyi_fno/mesh_2d.py, mesh_3d.py), Normalizer, the Markov
feature build (routines/grid_2d_markov.py:124-170).

Parity status: PINNED against golden vectors generated from the imported reference (tests/golden/*.npz, generator
tools/make_golden.py).  Two pieces restate reference routines whose modules cannot be imported here (they need jax /
wandb / pytorch_lightning):
  * ``velocity_features`` / ``velocity_wavenumbers`` (the `use_velocity` branch of routines/grid_2d_markov.py:82-94,130-144):
    PINNED since round 2 by a run of the reference's own `_build_features` / `encode_positions` bodies, lifted from the
    file at generation time (tools/make_golden_velocity.py -> tests/golden/markov_velocity.npz); only the wavenumber meshes
    (jax_cfd ``Grid.rfft_mesh()``, an un-vendored dependency pinned at git rev eb4d723e) are restated from that
    function's published definition.  Also held to analytic known answers (tests/test_velocity.py).
  * ``rollout_learning_step`` (routines/grid_2d_rollout.py:75-150): PINNED since round 4 by a run of the reference's own
    `forward` / `_learning_step` bodies, lifted from the file at generation time and executed over the reference's real
    FNOZongyi2DBlock and LpLoss (tools/make_golden_rollout.py -> tests/golden/rollout_step.npz: losses, predictions,
    correlations, time_until and every parameter gradient for append_pos on / off and teacher forcing in train / eval mode).
``relu_mask`` / ``relu_masks`` arguments are a test device (gradient comparisons without the ReLU bit-flip
discontinuity, see ``feedforward``); they change nothing when absent.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# WNLinear  (linear.py:41-52; torch.nn.utils.weight_norm, dim=0)
# --------------------------------------------------------------------------
def wn_weight(g: Tensor, v: Tensor) -> Tensor:
    """W = g * v / ||v||_2 taken over every dim but 0 (one norm per output row)."""
    return v * (g / v.norm(2, dim=1, keepdim=True))


def linear_from_sd(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """Apply the (possibly weight-normed) linear stored under ``prefix``."""
    if prefix + "weight_g" in sd:
        w = wn_weight(sd[prefix + "weight_g"], sd[prefix + "weight_v"])
    else:
        w = sd[prefix + "weight"]
    return F.linear(x, w, sd.get(prefix + "bias"))


# --------------------------------------------------------------------------
# FeedForward  (feedforward.py:6-24)
# --------------------------------------------------------------------------
# tests only: set to a list to collect (prefix, hidden index, active set) of every ReLU this module evaluates (ReLU-flip
# detection in gradient comparisons: tests/oracle_util.py::relu_flips)
RELU_TRACE = None


def feedforward(sd: Dict[str, Tensor], prefix: str, x: Tensor, n_layers: int = 2,
                layer_norm: bool = False, relu_mask=None, dropout: float = 0.0, dropout_keep=None) -> Tensor:
    """n_layers x [linear -> dropout(p) -> ReLU except last -> LayerNorm iff last & layer_norm]  (feedforward.py:13-23).

    ``dropout`` > 0: nn.Dropout in training mode.  With ``dropout_keep`` (tests only: one keep mask per linear layer,
    [pixels, features] -- the masks another implementation drew) the dropout is ``x * keep / (1 - p)``; without it the masks come
    from torch's generator like the reference's.

    ``relu_mask`` (tests only): a list with one boolean tensor per hidden activation, [pixels, hidden] -- the ACTIVE SET
    another implementation chose.  The ReLU is then evaluated as ``pre * mask``: identical to relu(pre) wherever the two
    implementations agree on the sign, and it removes the one discontinuity of the block (a pre-activation within an ulp
    of zero flips between any two correct fp32 implementations and moves the gradients by ~1e-3), so gradients can be
    compared at rounding level."""
    for i in range(n_layers):
        x = linear_from_sd(sd, f"{prefix}layers.{i}.0.", x)
        if dropout > 0.0:
            if dropout_keep is not None:
                x = x * dropout_keep[i].reshape(x.shape).to(x.dtype) / (1.0 - dropout)
            else:
                x = F.dropout(x, dropout, training=True)
        if i < n_layers - 1:
            if RELU_TRACE is not None:      # tests only: this evaluation's own active set, before any injected one is applied
                RELU_TRACE.append((prefix, i, (x.detach() > 0)))
            if relu_mask is not None:
                x = x * relu_mask[i].reshape(x.shape).to(x.dtype)
            else:
                x = torch.relu(x)
        elif layer_norm:
            x = F.layer_norm(x, x.shape[-1:], sd[f"{prefix}layers.{i}.3.weight"],
                             sd[f"{prefix}layers.{i}.3.bias"])
    return x


# --------------------------------------------------------------------------
# SpectralConv2d.forward_fourier  (grid_2d.py:51-99)
# --------------------------------------------------------------------------
def dct_matrix(L: int, dtype) -> Tensor:
    """Orthonormal DCT-II matrix D[k, n] = s_k cos(pi (2n + 1) k / 2L), s_0 = sqrt(1/L), s_k = sqrt(2/L)
    (what modules/dct.py:dct(x, norm='ortho') applies along the last axis; idct(norm='ortho') applies D^T)."""
    n = torch.arange(L, dtype=torch.float64)
    D = torch.cos(math.pi * (2 * n[None, :] + 1) * n[:, None] / (2 * L)) * math.sqrt(2.0 / L)
    D[0] = D[0] / math.sqrt(2.0)
    return D.to(dtype)


def dct_branch(x_cf: Tensor, w: Tensor, modes: int, dim: int) -> Tensor:
    """One axis of the CNOFactorized* spectral conv (factorized_cno/grid_2d.py:57-70 / :72-87, mesh_2d.py, mesh_3d.py):
    DCT-II(norm='ortho') along ``dim`` -> keep ``modes`` lowest coefficients -> per-mode REAL channel mix with
    w[I, O, modes] -> zero-padded inverse DCT.  x_cf channels-first [B, I, *spatial]."""
    D = dct_matrix(x_cf.shape[dim], x_cf.dtype)[:modes]                    # [modes, L]
    xt = x_cf.movedim(dim, -1)                                             # [B, I, ..., L]
    kept = torch.einsum("...n,kn->...k", xt, D)                            # truncated DCT
    mixed = torch.einsum("bi...k,iok->bo...k", kept, w)                    # per-mode channel mix
    return torch.einsum("...k,kn->...n", mixed, D).movedim(-1, dim)        # zero-padded inverse = D^T


def spectral_branch(x_cf: Tensor, w: Optional[Tensor], modes: int, dim: int, mode: str) -> Tensor:
    """One axis of the factorized spectral conv on a channels-first tensor [B, I, M, N].

    rfft(norm='ortho') along ``dim`` -> keep ``modes`` lowest bins -> per-mode
    complex channel mix with w[I, O, modes, 2] -> zero-padded irfft(norm='ortho').
    """
    if w is not None and w.dim() == 3:          # real [I, O, modes] weights: the DCT operators (CNOFactorized*)
        return dct_branch(x_cf, w, modes, dim)
    L = x_cf.shape[dim]
    spec = torch.fft.rfft(x_cf, dim=dim, norm="ortho")
    kept = spec.narrow(dim, 0, modes)
    if mode == "full":
        wc = torch.view_as_complex(w.contiguous())
        eq = "bixy,ioy->boxy" if dim in (-1, 3) else "bixy,iox->boxy"
        kept = torch.einsum(eq, kept, wc)
    elif mode != "low-pass":
        raise ValueError(mode)
    shape = list(kept.shape)
    shape[dim] = L // 2 + 1
    padded = kept.new_zeros(shape)
    padded.narrow(dim, 0, modes).copy_(kept)
    return torch.fft.irfft(padded, n=L, dim=dim, norm="ortho")


def forward_fourier(x: Tensor, w_y: Optional[Tensor], w_x: Optional[Tensor], modes: int,
                    mode: str = "full") -> Tensor:
    """x [B, M, N, I] channels-last -> [B, M, N, O].

    ``w_y`` is fourier_weight[0] (mixes along the LAST spatial axis, grid_2d.py:65-68),
    ``w_x`` is fourier_weight[1] (first spatial axis, grid_2d.py:83-86).
    """
    x_cf = x.permute(0, 3, 1, 2)
    xy = spectral_branch(x_cf, w_y, modes, -1, mode)
    xx = spectral_branch(x_cf, w_x, modes, -2, mode)
    return (xx + xy).permute(0, 2, 3, 1)


def forward_fourier_plus(x: Tensor, w0: Tensor, w1: Tensor, modes: int) -> Tensor:
    """Non-factorized spectral conv of FNOPlus2DBlock (zongyi_fno/grid_plus_2d.py:52-83): rfft2(norm='ortho'), the two
    corner blocks [:K, :K] and [-K:, :K] mixed with w0 / w1 [I, O, K, K, 2], zero-padded irfft2."""
    x_cf = x.permute(0, 3, 1, 2)
    B, I, M, N = x_cf.shape
    x_ft = torch.fft.rfft2(x_cf, s=(M, N), norm="ortho")
    out_ft = x_ft.new_zeros(B, w0.shape[1], M, N // 2 + 1)
    out_ft[:, :, :modes, :modes] = torch.einsum("bixy,ioxy->boxy", x_ft[:, :, :modes, :modes],
                                                torch.view_as_complex(w0.contiguous()))
    out_ft[:, :, -modes:, :modes] = torch.einsum("bixy,ioxy->boxy", x_ft[:, :, -modes:, :modes],
                                                 torch.view_as_complex(w1.contiguous()))
    return torch.fft.irfft2(out_ft, s=(M, N), norm="ortho").permute(0, 2, 3, 1)


# --------------------------------------------------------------------------
# FNOFactorized2DBlock.forward  (grid_2d.py:154-177); with spectral="plus" the same block around the non-factorized
# spectral conv = FNOPlus2DBlock.forward (zongyi_fno/grid_plus_2d.py:138-161)
# --------------------------------------------------------------------------
def ffno2d_block(sd: Dict[str, Tensor], x: Tensor, *, modes: int, n_layers: int,
                 use_fork: bool = False, mode: str = "full", n_ff_layers: int = 2,
                 layer_norm: bool = False, return_intermediates: bool = False, spectral: str = "factorized",
                 relu_masks=None, dropout: float = 0.0, in_dropout: float = 0.0, dropout_keeps=None):
    """Forward of the whole block over a reference-layout state_dict.

    Returns {'forecast', 'forecast_list'} like the reference; with
    ``return_intermediates`` also the per-layer inputs for test diagnostics.
    ``relu_masks`` (tests only): {("backcast" | "forecast", layer): [mask per hidden activation]}, see ``feedforward``.
    """
    rm = relu_masks or {}
    dk = dropout_keeps or {}      # tests only: {("backcast" | "forecast", layer): [keep mask per linear], "in": keep mask}
    def head(t: Tensor) -> Tensor:
        return linear_from_sd(sd, "out.1.", linear_from_sd(sd, "out.0.", t))

    x = linear_from_sd(sd, "in_proj.", x)
    if in_dropout > 0.0:          # x = self.drop(x)  (grid_2d.py:113,158), training mode
        x = x * dk["in"].reshape(x.shape).to(x.dtype) / (1.0 - in_dropout) if "in" in dk else F.dropout(x, in_dropout, training=True)
    forecast = 0
    forecast_list: List[Tensor] = []
    inter = [x]
    b = None
    for i in range(n_layers):
        pre = f"spectral_layers.{i}."
        s = x
        if mode != "no-fourier" and spectral == "plus":
            s = forward_fourier_plus(x, sd[pre + "fourier_weight.0"], sd[pre + "fourier_weight.1"], modes)
        elif mode != "no-fourier":
            s = forward_fourier(x, sd.get(pre + "fourier_weight.0"), sd.get(pre + "fourier_weight.1"),
                                modes, mode)
        b = feedforward(sd, pre + "backcast_ff.", s, n_ff_layers, layer_norm, rm.get(("backcast", i)), dropout,
                        dk.get(("backcast", i)))
        if use_fork:
            f_out = head(feedforward(sd, pre + "forecast_ff.", s, n_ff_layers, layer_norm, rm.get(("forecast", i)), dropout,
                                     dk.get(("forecast", i))))
            forecast = forecast + f_out
            forecast_list.append(f_out)
        x = x + b
        inter.append(x)
    if not use_fork:
        forecast = head(b)
    out = {"forecast": forecast, "forecast_list": forecast_list}
    if return_intermediates:
        out["layer_inputs"] = inter
    return out


# --------------------------------------------------------------------------
# LpLoss.rel  (loss.py:33-46): mean_b ||x_b - y_b||_2 / ||y_b||_2
# --------------------------------------------------------------------------
def lp_rel_loss(pred: Tensor, target: Tensor) -> Tensor:
    n = pred.shape[0]
    d = (pred.reshape(n, -1) - target.reshape(n, -1)).norm(2, dim=1)
    return (d / target.reshape(n, -1).norm(2, dim=1)).mean()


# --------------------------------------------------------------------------
# Cosine-with-warmup multiplier  (schedulers/cosine_with_warmup.py:6-17)
# --------------------------------------------------------------------------
def cosine_warmup_factor(step: int, num_warmup_steps: int, num_training_steps: int,
                         num_cycles: float = 0.5) -> float:
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


# --------------------------------------------------------------------------
# Reference-compatible random init (for benches / tests that do not load golden weights).
# Shapes & init rules: grid_2d.py:103-152, linear.py:41-52, SURVEY.md a3/a5.
# --------------------------------------------------------------------------
def init_block_state_dict(*, modes: int, width: int, input_dim: int, n_layers: int,
                          share_weight: bool, factor: int, ff_weight_norm: bool, gain: float = 1.0,
                          seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def add_linear(prefix: str, fan_in: int, fan_out: int):
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)
        w = (torch.rand(fan_out, fan_in, generator=gen, dtype=torch.float64) * 2 - 1) * bound
        bias = (torch.rand(fan_out, generator=gen, dtype=torch.float64) * 2 - 1) * bound
        if ff_weight_norm:
            sd[prefix + "weight_g"] = w.norm(2, dim=1, keepdim=True).to(dtype)
            sd[prefix + "weight_v"] = w.to(dtype)
        else:
            sd[prefix + "weight"] = w.to(dtype)
        sd[prefix + "bias"] = bias.to(dtype)

    def fourier_pair():
        # xavier_normal_ on [I, O, K, 2]: fan_in = O*K*2, fan_out = I*K*2
        std = gain * math.sqrt(2.0 / (2 * width * modes * 2))
        return [(torch.randn(width, width, modes, 2, generator=gen, dtype=torch.float64) * std).to(dtype)
                for _ in range(2)]

    add_linear("in_proj.", input_dim, width)
    shared = fourier_pair() if share_weight else None
    if shared is not None:
        sd["fourier_weight.0"], sd["fourier_weight.1"] = shared
    for i in range(n_layers):
        pre = f"spectral_layers.{i}."
        fw = shared if shared is not None else fourier_pair()
        sd[pre + "fourier_weight.0"], sd[pre + "fourier_weight.1"] = fw
        add_linear(pre + "backcast_ff.layers.0.0.", width, width * factor)
        add_linear(pre + "backcast_ff.layers.1.0.", width * factor, width)
    add_linear("out.0.", width, 128)
    add_linear("out.1.", 128, 1)
    return sd


# --------------------------------------------------------------------------
# FNOFactorizedMesh3D  (factorized_fno/mesh_3d.py:55-112 forward_fourier, :160-189 forward)
# --------------------------------------------------------------------------
def spectral_branch_nd(x_cf: Tensor, w: Tensor, modes: int, dim: int) -> Tensor:
    """One axis of the 3-D factorized spectral conv on channels-first [B, I, S1, S2, S3]."""
    if w.dim() == 3:                            # CNOFactorizedMesh3D
        return dct_branch(x_cf, w, modes, dim)
    L = x_cf.shape[dim]
    kept = torch.fft.rfft(x_cf, dim=dim, norm="ortho").narrow(dim, 0, modes)
    letter = {-3: "x", -2: "y", -1: "z"}[dim]
    kept = torch.einsum(f"bixyz,io{letter}->boxyz", kept, torch.view_as_complex(w.contiguous()))
    shape = list(kept.shape)
    shape[dim] = L // 2 + 1
    padded = kept.new_zeros(shape)
    padded.narrow(dim, 0, modes).copy_(kept)
    return torch.fft.irfft(padded, n=L, dim=dim, norm="ortho")


def forward_fourier_3d(x: Tensor, weights, modes) -> Tensor:
    """x [B, S1, S2, S3, I]; weights[0|1|2] mix the x|y|z axis with modes[0|1|2] (mesh_3d.py:62-103)."""
    x_cf = x.permute(0, 4, 1, 2, 3)
    out = (spectral_branch_nd(x_cf, weights[2], modes[2], -1) + spectral_branch_nd(x_cf, weights[1], modes[1], -2) +
           spectral_branch_nd(x_cf, weights[0], modes[0], -3))
    return out.permute(0, 2, 3, 4, 1)


def mesh3d_grid(shape, dtype) -> Tensor:
    """linspace(0,1) coordinate channels (mesh_3d.py:178-189)."""
    B, X, Y, Z = shape[0], shape[1], shape[2], shape[3]
    gx = torch.linspace(0, 1, X, dtype=torch.float64).to(dtype).reshape(1, X, 1, 1, 1).expand(B, X, Y, Z, 1)
    gy = torch.linspace(0, 1, Y, dtype=torch.float64).to(dtype).reshape(1, 1, Y, 1, 1).expand(B, X, Y, Z, 1)
    gz = torch.linspace(0, 1, Z, dtype=torch.float64).to(dtype).reshape(1, 1, 1, Z, 1).expand(B, X, Y, Z, 1)
    return torch.cat((gx, gy, gz), dim=-1)


def ffno_mesh3d(sd: Dict[str, Tensor], x: Tensor, *, modes, n_layers: int, padding: int = 8, relu_masks=None) -> Tensor:
    """x [B, X, Y, Z, input_dim - 3] -> [B, X, Y, Z, output_dim] (mesh_3d.py:160-176)."""
    rm = relu_masks or {}
    x = torch.cat((x, mesh3d_grid(x.shape, x.dtype)), dim=-1)
    x = linear_from_sd(sd, "in_proj.", x)
    x = F.pad(x.permute(0, 4, 1, 2, 3), [0, padding, 0, padding, 0, padding]).permute(0, 2, 3, 4, 1)
    b = None
    for i in range(n_layers):
        pre = f"spectral_layers.{i}."
        s = forward_fourier_3d(x, [sd[pre + f"fourier_weight.{w}"] for w in range(3)], modes)
        b = feedforward(sd, pre + "backcast_ff.", s, relu_mask=rm.get(("backcast", i)))
        x = x + b
    b = b[:, :-padding, :-padding, :-padding, :]
    return linear_from_sd(sd, "out.1.", linear_from_sd(sd, "out.0.", b))


# --------------------------------------------------------------------------
# Velocity features of Grid2DMarkovExperiment (routines/grid_2d_markov.py:82-94 buffers, :130-144 use).
# The wavenumber meshes come from jax_cfd (``Grid(shape, domain).rfft_mesh()``; dependency pinned at git rev
# eb4d723e in the reference's pyproject.toml:29, NOT present under /root/reference and not installable here), so
# they are restated from that function's published definition: fftfreq over every axis but the last, rfftfreq over
# the last, spacing = domain length / n, meshgrid(indexing='ij').  PARITY OF THIS PIECE IS PINNED ANALYTICALLY
# (plane-wave known answers, curl/divergence identities in tests/test_velocity.py), not by a reference run:
# fourierflow.routines cannot be imported without jax.
# --------------------------------------------------------------------------
def velocity_wavenumbers(X: int, Y: int, domain=((0.0, 2 * math.pi), (0.0, 2 * math.pi))):
    """-> kx [X, Y//2+1], ky [X, Y//2+1] (float32), lap [X, Y//2+1] (complex64), as registered at :88-94."""
    import numpy as np
    lx, ly = domain[0][1] - domain[0][0], domain[1][1] - domain[1][0]
    kx, ky = np.meshgrid(np.fft.fftfreq(X, d=lx / X), np.fft.rfftfreq(Y, d=ly / Y), indexing="ij")
    lap = (2 * np.pi * 1j) ** 2 * (np.abs(kx) ** 2 + np.abs(ky) ** 2)
    lap[0, 0] = 1
    return (torch.from_numpy(kx.astype(np.float32)), torch.from_numpy(ky.astype(np.float32)),
            torch.from_numpy(lap.astype(np.complex64)))


def velocity_features(x: Tensor, domain=((0.0, 2 * math.pi), (0.0, 2 * math.pi))) -> Tensor:
    """x = vorticity [B, X, Y, 1] -> cat([x, u, v], -1) exactly as grid_2d_markov.py:130-144."""
    B, X, Y, _ = x.shape
    kx, ky, lap = velocity_wavenumbers(X, Y, domain)
    cdt = torch.complex128 if x.dtype == torch.float64 else torch.complex64
    omega_hat = torch.fft.rfftn(x, dim=[1, 2], norm="backward")
    psi_hat = -omega_hat / lap.to(cdt)[None, :, :, None]
    q = 2 * math.pi * 1j * ky.to(x.dtype)[None, :, :, None] * psi_hat
    q = torch.fft.irfftn(q, dim=[1, 2], norm="backward")
    v = -2 * math.pi * 1j * kx.to(x.dtype)[None, :, :, None] * psi_hat
    v = torch.fft.irfftn(v, dim=[1, 2], norm="backward")
    return torch.cat([x, q, v], dim=-1)


# --------------------------------------------------------------------------
# FNOFactorizedMesh2D  (factorized_fno/mesh_2d.py:56-106 forward_fourier, :146-175 forward)
# NOTE the weight order differs from grid_2d.py: here fourier_weight[0] mixes the FIRST spatial axis (x, modes_x)
# and fourier_weight[1] the LAST (y, modes_y) -- mesh_2d.py:71-75,92-96.
# --------------------------------------------------------------------------
def mesh2d_grid(shape, dtype) -> Tensor:
    """linspace(0,1) coordinate channels (mesh_2d.py:167-175)."""
    B, X, Y = shape[0], shape[1], shape[2]
    gx = torch.linspace(0, 1, X, dtype=torch.float64).to(dtype).reshape(1, X, 1, 1).expand(B, X, Y, 1)
    gy = torch.linspace(0, 1, Y, dtype=torch.float64).to(dtype).reshape(1, 1, Y, 1).expand(B, X, Y, 1)
    return torch.cat((gx, gy), dim=-1)


def ffno_mesh2d(sd: Dict[str, Tensor], x: Tensor, *, modes, n_layers: int, padding: int = 8, relu_masks=None) -> Tensor:
    """x [B, X, Y, input_dim - 2] -> [B, X, Y, 1] (mesh_2d.py:146-165); modes = (modes_x, modes_y)."""
    rm = relu_masks or {}
    x = torch.cat((x, mesh2d_grid(x.shape, x.dtype)), dim=-1)
    x = linear_from_sd(sd, "in_proj.", x)
    x = F.pad(x.permute(0, 3, 1, 2), [0, padding, 0, padding]).permute(0, 2, 3, 1)
    b = None
    for i in range(n_layers):
        pre = f"spectral_layers.{i}."
        x_cf = x.permute(0, 3, 1, 2)
        s = (spectral_branch(x_cf, sd[pre + "fourier_weight.1"], modes[1], dim=-1, mode="full") +
             spectral_branch(x_cf, sd[pre + "fourier_weight.0"], modes[0], dim=-2, mode="full")).permute(0, 2, 3, 1)
        b = feedforward(sd, pre + "backcast_ff.", s, relu_mask=rm.get(("backcast", i)))
        x = x + b
    b = b[:, :-padding, :-padding, :]
    return linear_from_sd(sd, "out.1.", linear_from_sd(sd, "out.0.", b))


# --------------------------------------------------------------------------
# FNOZongyi2DBlock  (zongyi_fno/grid_2d.py:16-129) -- BASELINE config 0, CPU plumbing case.
# The non-factorized baseline: full rfft2, two corner blocks of modes x modes weights, irfft2,
# + Linear residual + ReLU.  (Reference quirk kept: irfft2 is called with s=(N, M), grid_2d.py:68.)
# --------------------------------------------------------------------------
def zongyi_spectral_conv(sd: Dict[str, Tensor], prefix: str, x: Tensor, n_modes: int, conv_residual: bool = True) -> Tensor:
    B, M, N, I = x.shape
    lin = F.linear(x, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])
    xf = torch.fft.rfft2(x.permute(0, 3, 1, 2), s=(M, N), norm="ortho")
    out = xf.new_zeros(B, sd[prefix + "fourier_weight.0"].shape[1], M, N // 2 + 1)
    w0 = torch.view_as_complex(sd[prefix + "fourier_weight.0"].contiguous())
    w1 = torch.view_as_complex(sd[prefix + "fourier_weight.1"].contiguous())
    out[:, :, :n_modes, :n_modes] = torch.einsum("bixy,ioxy->boxy", xf[:, :, :n_modes, :n_modes], w0)
    out[:, :, -n_modes:, :n_modes] = torch.einsum("bixy,ioxy->boxy", xf[:, :, -n_modes:, :n_modes], w1)
    y = torch.fft.irfft2(out, s=(N, M), norm="ortho").permute(0, 2, 3, 1)
    if conv_residual:
        return torch.relu(y + lin)
    return torch.relu(F.linear(y, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"]))


def fno_zongyi_2d(sd: Dict[str, Tensor], x: Tensor, *, modes: int, n_layers: int, residual: bool = False,
                  conv_residual: bool = True) -> Dict[str, Tensor]:
    x = F.linear(x, sd["in_proj.weight"], sd["in_proj.bias"])
    for i in range(n_layers):
        y = zongyi_spectral_conv(sd, f"spectral_layers.{i}.", x, modes, conv_residual)
        x = y + x if residual else y
    x = torch.relu(F.linear(x, sd["feedforward.0.weight"], sd["feedforward.0.bias"]))
    return {"forecast": F.linear(x, sd["feedforward.2.weight"], sd["feedforward.2.bias"])}


# --------------------------------------------------------------------------
# FNOMesh2D (zongyi_fno/mesh_2d.py:14-106), the geo-FNO baseline: fc0 on (x, grid) -> pad 8 at the far ends ->
# n_layers x [rfft2 -> corner blocks [:m1, :m2] / [-m1:, :m2] x complex weights -> irfft2, + 1x1 conv, GELU except
# last] -> crop -> fc1 -> GELU -> fc2.  State-dict layout of the reference (weights1/2 complex [I, O, m1, m2]).
# --------------------------------------------------------------------------
def fno_mesh2d_grid(B: int, X: int, Y: int, dtype=torch.float32) -> Tensor:
    gx = torch.linspace(0, 1, X, dtype=dtype).reshape(1, X, 1, 1).expand(B, X, Y, 1)          # :99-105
    gy = torch.linspace(0, 1, Y, dtype=dtype).reshape(1, 1, Y, 1).expand(B, X, Y, 1)
    return torch.cat((gx, gy), dim=-1)


def fno_mesh2d(sd: Dict[str, Tensor], x: Tensor, *, modes1: int, modes2: int, n_layers: int, padding: int = 8) -> Tensor:
    B, X, Y, _ = x.shape
    h = torch.cat((x, fno_mesh2d_grid(B, X, Y, x.dtype)), dim=-1)                              # :79-80
    h = F.linear(h, sd["fc0.weight"], sd["fc0.bias"]).permute(0, 3, 1, 2)                      # :81-82
    h = F.pad(h, [0, padding, 0, padding])                                                     # :84
    for i in range(n_layers):
        hf = torch.fft.rfft2(h)                                                                # :41
        out = hf.new_zeros(B, h.shape[1], h.shape[-2], h.shape[-1] // 2 + 1)
        out[:, :, :modes1, :modes2] = torch.einsum("bixy,ioxy->boxy", hf[:, :, :modes1, :modes2], sd[f"convs.{i}.weights1"])
        out[:, :, -modes1:, :modes2] = torch.einsum("bixy,ioxy->boxy", hf[:, :, -modes1:, :modes2], sd[f"convs.{i}.weights2"])
        x1 = torch.fft.irfft2(out, s=(h.shape[-2], h.shape[-1]))                               # :52
        x2 = F.conv2d(h, sd[f"ws.{i}.weight"], sd[f"ws.{i}.bias"])                             # :89
        h = x1 + x2
        if i < n_layers - 1:
            h = F.gelu(h)                                                                      # :91-92
    h = h[..., :-padding, :-padding].permute(0, 2, 3, 1)                                       # :94-95
    h = F.gelu(F.linear(h, sd["fc1.weight"], sd["fc1.bias"]))
    return F.linear(h, sd["fc2.weight"], sd["fc2.bias"])


def fno_mesh3d(sd: Dict[str, Tensor], x: Tensor, *, modes1: int, modes2: int, modes3: int, n_layers: int,
               padding: int = 5) -> Tensor:
    """FNOMesh3D (zongyi_fno/mesh_3d.py:63-113): x [B, X, Y, Z, 1] -> [B, X, Y, Z, 4]; four corner blocks (:38-57)."""
    B, X, Y, Z, _ = x.shape
    lin = lambda n, d: torch.linspace(0, 1, n, dtype=x.dtype).reshape([n if i == d else 1 for i in range(3)])   # noqa: E731
    grid = torch.stack([lin(X, 0).expand(X, Y, Z), lin(Y, 1).expand(X, Y, Z), lin(Z, 2).expand(X, Y, Z)], dim=-1)
    h = torch.cat((x, grid[None].expand(B, X, Y, Z, 3)), dim=-1)                               # :87-88
    h = F.linear(h, sd["fc0.weight"], sd["fc0.bias"]).permute(0, 4, 1, 2, 3)                   # :89-90
    h = F.pad(h, [0, padding, 0, padding, 0, padding])                                         # :91
    m1, m2, m3 = modes1, modes2, modes3
    for i in range(n_layers):
        hf = torch.fft.rfftn(h, dim=[-3, -2, -1])                                              # :40
        out = hf.new_zeros(B, h.shape[1], h.shape[-3], h.shape[-2], h.shape[-1] // 2 + 1)
        mul = lambda a, w: torch.einsum("bixyz,ioxyz->boxyz", a, w)                            # noqa: E731
        out[:, :, :m1, :m2, :m3] = mul(hf[:, :, :m1, :m2, :m3], sd[f"convs.{i}.weights1"])     # :45-52
        out[:, :, -m1:, :m2, :m3] = mul(hf[:, :, -m1:, :m2, :m3], sd[f"convs.{i}.weights2"])
        out[:, :, :m1, -m2:, :m3] = mul(hf[:, :, :m1, -m2:, :m3], sd[f"convs.{i}.weights3"])
        out[:, :, -m1:, -m2:, :m3] = mul(hf[:, :, -m1:, -m2:, :m3], sd[f"convs.{i}.weights4"])
        x1 = torch.fft.irfftn(out, s=(h.shape[-3], h.shape[-2], h.shape[-1]))                  # :56
        x2 = F.conv3d(h, sd[f"ws.{i}.weight"], sd[f"ws.{i}.bias"])                             # :95
        h = x1 + x2
        if i < n_layers - 1:
            h = F.gelu(h)
    h = h[..., :-padding, :-padding, :-padding].permute(0, 2, 3, 4, 1)                         # :100-101
    h = F.gelu(F.linear(h, sd["fc1.weight"], sd["fc1.bias"]))
    return F.linear(h, sd["fc2.weight"], sd["fc2.bias"])


# --------------------------------------------------------------------------
# Grid2DRolloutExperiment._learning_step (routines/grid_2d_rollout.py:75-150, use_fourier_position=False): feed the
# model its own forecasts for n_steps; loss = mean of the per-step LpLoss.rel; also the validation metrics.
# ``conv`` maps [B, X, Y, C] -> forecast [B, X, Y, 1].
# --------------------------------------------------------------------------
def rollout_positions(B: int, X: int, Y: int, dtype=torch.float32) -> Tensor:
    ticks = torch.linspace(0, 1, X, dtype=dtype)                       # :96 (the X ticks serve both axes, :97-98)
    gx = ticks[None, :, None, None].expand(B, X, Y, 1)
    gy = ticks[None, None, :, None].expand(B, X, Y, 1)
    return torch.cat([gx, gy], dim=-1)


def rollout_learning_step(conv, xx: Tensor, yy: Tensor, n_steps: int, append_pos: bool = True,
                          teacher_forcing: bool = False, training: bool = True, step_size: float = 1.0):
    B, X, Y, _ = xx.shape
    pos = rollout_positions(B, X, Y, xx.dtype)
    embeds, P = xx, 2
    loss, step_losses, preds = 0, [], []
    for t in range(n_steps):
        y = yy[..., t:t + 1]
        im = conv(embeds)
        l = lp_rel_loss(im.reshape(B, -1), y.reshape(B, -1))           # :112
        step_losses.append(l)
        loss = loss + l
        preds.append(im)
        if teacher_forcing and training:                               # :117-118
            im = y
        if append_pos:                                                 # :123-126
            embeds = torch.cat((embeds[..., 1:-P], im, pos), dim=-1)
        else:
            embeds = torch.cat((embeds[..., 1:], im), dim=-1)
    pred = torch.cat(preds, dim=-1)
    loss = loss / n_steps
    yy = yy[..., :n_steps]
    loss_full = lp_rel_loss(pred.reshape(B, -1), yy.reshape(B, -1))    # :129
    pn = torch.norm(pred, dim=[1, 2], keepdim=True)
    yn = torch.norm(yy, dim=[1, 2], keepdim=True)
    p = ((pred / pn) * (yy / yn)).sum(dim=[1, 2]).mean(dim=0)          # :131-135
    div = (p < 0.95).nonzero()
    time_until = (int(div[0, 0]) if len(div) > 0 else len(p)) * step_size
    return loss, loss_full, pred, step_losses, p, time_until


# --------------------------------------------------------------------------
# Markov routine glue: Normalizer (modules/normalizer.py:18-77) and _build_features / _training_step
# (routines/grid_2d_markov.py:124-193, use_position=True, no velocity/force/mu).
# --------------------------------------------------------------------------
class NormalizerState:
    """Running sums exactly as the reference's buffers (count, sum, sum_squared, n_accumulations)."""

    def __init__(self, size: int, max_accumulations: float = 1e6, std_epsilon: float = 1e-8, dtype=torch.float32):
        self.sum = torch.zeros(size, dtype=dtype)
        self.sum_squared = torch.zeros(size, dtype=dtype)
        self.count = torch.zeros((), dtype=dtype)
        self.n_accumulations = torch.zeros((), dtype=dtype)
        self.max_accumulations, self.eps = max_accumulations, torch.full((size,), std_epsilon, dtype=dtype)

    @property
    def mean(self):
        return self.sum / torch.clamp(self.count, min=1.0)

    @property
    def std(self):
        return torch.maximum(torch.sqrt(self.sum_squared / torch.clamp(self.count, min=1.0) - self.mean ** 2), self.eps)

    def forward(self, x: Tensor, training: bool = True) -> Tensor:
        flat = x.reshape(-1, x.shape[-1])
        if training and self.n_accumulations < self.max_accumulations:
            self.sum = self.sum + flat.sum(dim=0)
            self.sum_squared = self.sum_squared + (flat ** 2).sum(dim=0)
            self.count = self.count + flat.shape[0]
            self.n_accumulations = self.n_accumulations + 1
        return ((flat - self.mean) / self.std).reshape(x.shape)

    def inverse(self, x: Tensor, channel: int) -> Tensor:
        return x * self.std[channel] + self.mean[channel]


def markov_features(x: Tensor, norm: Optional[NormalizerState], noise: Optional[Tensor], noise_std: float,
                    low: float = 0.0, high: float = 1.0, training: bool = True, use_position: bool = True,
                    force: Optional[Tensor] = None, mu: Optional[Tensor] = None) -> Tensor:
    """grid_2d_markov.py:146-170: x | position | force map f [B, M, N] (append_force) | viscosity mu [B] (append_mu)."""
    B, M, N, _ = x.shape
    feats = x
    if use_position:
        gm = torch.linspace(low, high, M, dtype=x.dtype)
        gn = torch.linspace(low, high, N, dtype=x.dtype)
        pos = torch.stack(torch.meshgrid(gm, gn, indexing="ij"), dim=-1).expand(B, M, N, 2)
        feats = torch.cat([feats, pos], dim=-1)
    if force is not None:
        feats = torch.cat([feats, force.reshape(B, M, N, 1).to(x.dtype)], dim=-1)
    if mu is not None:
        feats = torch.cat([feats, mu.reshape(B, 1, 1, 1).to(x.dtype).expand(B, M, N, 1)], dim=-1)
    if norm is not None:
        feats = norm.forward(feats, training)
    if noise is not None:
        feats = feats + noise * noise_std
    return feats
