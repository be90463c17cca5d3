"""Deterministic inputs shared by tools/make_golden.py (which feeds them to the imported
reference) and the tests (which feed them to the oracle / the HIP path).

Weights and inputs are a pure function of (kwargs, seed) through numpy's
RandomState (MT19937, stable across numpy versions), so the committed fixtures
only need to hold the reference's OUTPUTS (forecast, loss, gradients) -- small files.
State-dict key layout = the reference's: shared tensors
appear under every duplicate key the reference registers.
"""
from __future__ import annotations

import ast
import math
import os
from typing import Dict

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

BLOCK_DEFAULTS = dict(input_dim=12, dropout=0.0, in_dropout=0.0, n_layers=4, share_weight=False,
                      share_fork=False, factor=2, ff_weight_norm=False, n_ff_layers=2, gain=1,
                      layer_norm=False, use_fork=False, mode="full")


def full_kwargs(kw: dict) -> dict:
    out = dict(BLOCK_DEFAULTS)
    out.update(kw)
    return out


def _linear(rs, sd, prefix, fan_in, fan_out, wnorm):
    bound = 1.0 / math.sqrt(fan_in)
    w = rs.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32)
    b = rs.uniform(-bound, bound, size=(fan_out,)).astype(np.float32)
    if wnorm:
        # g deliberately != ||v|| so the weight-norm arithmetic is actually exercised
        g = (np.linalg.norm(w, axis=1, keepdims=True) * rs.uniform(0.7, 1.3, size=(fan_out, 1))).astype(np.float32)
        sd[prefix + "weight_g"] = g
        sd[prefix + "weight_v"] = w
    else:
        sd[prefix + "weight"] = w
    sd[prefix + "bias"] = b


def _ff(rs, width, factor, wnorm, n_ff_layers, layer_norm):
    sd: Dict[str, np.ndarray] = {}
    for i in range(n_ff_layers):
        fin = width if i == 0 else width * factor
        fout = width if i == n_ff_layers - 1 else width * factor
        _linear(rs, sd, f"layers.{i}.0.", fin, fout, wnorm)
        if layer_norm and i == n_ff_layers - 1:
            sd[f"layers.{i}.3.weight"] = rs.uniform(0.5, 1.5, size=(fout,)).astype(np.float32)
            sd[f"layers.{i}.3.bias"] = rs.uniform(-0.2, 0.2, size=(fout,)).astype(np.float32)
    return sd


def make_block_state_dict(kw: dict, seed: int, plus: bool = False) -> Dict[str, np.ndarray]:
    """Reference-layout state_dict (incl. duplicated shared keys) for FNOFactorized2DBlock(**kw); ``plus``: for
    FNOPlus2DBlock(**kw) -- identical structure, Fourier weights [C, C, K, K, 2] (grid_plus_2d.py:22-29)."""
    kw = full_kwargs(kw)
    rs = np.random.RandomState(seed)
    C, K = kw["width"], kw["modes"]
    wn, f, nff, ln = kw["ff_weight_norm"], kw["factor"], kw["n_ff_layers"], kw["layer_norm"]
    sd: Dict[str, np.ndarray] = {}
    _linear(rs, sd, "in_proj.", kw["input_dim"], C, wn)

    def fourier_pair():
        if plus:
            std = 2.0 * kw["gain"] * math.sqrt(2.0 / (2 * C * K * K * 2))   # (x2: keeps the spectral path visible)
            return [(rs.standard_normal((C, C, K, K, 2)) * std).astype(np.float32) for _ in range(2)]
        std = kw["gain"] * math.sqrt(2.0 / (2 * C * K * 2))
        return [(rs.standard_normal((C, C, K, 2)) * std).astype(np.float32) for _ in range(2)]

    shared_fc = shared_bc = None
    if kw["share_fork"]:
        if kw["use_fork"]:
            shared_fc = _ff(rs, C, f, wn, nff, ln)
            for k, v in shared_fc.items():
                sd["forecast_ff." + k] = v
        shared_bc = _ff(rs, C, f, wn, nff, ln)
        for k, v in shared_bc.items():
            sd["backcast_ff." + k] = v
    shared_fw = None
    if kw["share_weight"]:
        shared_fw = fourier_pair()
        sd["fourier_weight.0"], sd["fourier_weight.1"] = shared_fw
    for i in range(kw["n_layers"]):
        pre = f"spectral_layers.{i}."
        fw = shared_fw if shared_fw is not None else fourier_pair()
        sd[pre + "fourier_weight.0"], sd[pre + "fourier_weight.1"] = fw
        if kw["use_fork"]:
            fc = shared_fc if shared_fc is not None else _ff(rs, C, f, wn, nff, ln)
            for k, v in fc.items():
                sd[pre + "forecast_ff." + k] = v
        bc = shared_bc if shared_bc is not None else _ff(rs, C, f, wn, nff, ln)
        for k, v in bc.items():
            sd[pre + "backcast_ff." + k] = v
    _linear(rs, sd, "out.0.", C, 128, wn)
    _linear(rs, sd, "out.1.", 128, 1, wn)
    return sd


def make_mesh3d_state_dict(kw: dict, seed: int) -> Dict[str, np.ndarray]:
    """Reference-layout state_dict for FNOFactorizedMesh3D(**kw) (mesh_3d.py:116-157)."""
    rs = np.random.RandomState(seed)
    C, wn, f = kw["width"], kw["ff_weight_norm"], kw["factor"]
    Ks = (kw["modes_x"], kw["modes_y"], kw["modes_z"])
    sd: Dict[str, np.ndarray] = {}
    _linear(rs, sd, "in_proj.", kw["input_dim"], C, wn)

    def fourier_triple():
        return [(rs.standard_normal((C, C, K, 2)) * math.sqrt(2.0 / (2 * C * K * 2))).astype(np.float32) for K in Ks]

    shared = fourier_triple() if kw["share_weight"] else None
    if shared is not None:
        for w in range(3):
            sd[f"fourier_weight.{w}"] = shared[w]
    for i in range(kw["n_layers"]):
        pre = f"spectral_layers.{i}."
        fw = shared if shared is not None else fourier_triple()
        for w in range(3):
            sd[pre + f"fourier_weight.{w}"] = fw[w]
        for k, v in _ff(rs, C, f, wn, 2, False).items():
            sd[pre + "backcast_ff." + k] = v
    _linear(rs, sd, "out.0.", C, 128, wn)
    _linear(rs, sd, "out.1.", 128, kw["output_dim"], wn)
    return sd


def make_mesh2d_state_dict(kw: dict, seed: int) -> Dict[str, np.ndarray]:
    """Reference-layout state_dict for FNOFactorizedMesh2D(**kw) (mesh_2d.py:109-144): fourier_weight [x, y]."""
    rs = np.random.RandomState(seed)
    C, wn, f = kw["width"], kw["ff_weight_norm"], kw["factor"]
    Ks = (kw["modes_x"], kw["modes_y"])
    sd: Dict[str, np.ndarray] = {}
    _linear(rs, sd, "in_proj.", kw["input_dim"], C, wn)

    def fourier_pair():
        return [(rs.standard_normal((C, C, K, 2)) * math.sqrt(2.0 / (2 * C * K * 2))).astype(np.float32) for K in Ks]

    shared = fourier_pair() if kw["share_weight"] else None
    if shared is not None:
        for w in range(2):
            sd[f"fourier_weight.{w}"] = shared[w]
    for i in range(kw["n_layers"]):
        pre = f"spectral_layers.{i}."
        fw = shared if shared is not None else fourier_pair()
        for w in range(2):
            sd[pre + f"fourier_weight.{w}"] = fw[w]
        for k, v in _ff(rs, C, f, wn, 2, False).items():
            sd[pre + "backcast_ff." + k] = v
    _linear(rs, sd, "out.0.", C, 128, wn)
    _linear(rs, sd, "out.1.", 128, 1, wn)
    return sd


def make_mesh2d_io(kw: dict, seed: int, B: int, S):
    rs = np.random.RandomState(seed + 300007)
    x = rs.standard_normal((B, *S, kw["input_dim"] - 2)).astype(np.float32)
    target = rs.standard_normal((B, *S, 1)).astype(np.float32)
    return x, target


def make_mesh3d_io(kw: dict, seed: int, B: int, S):
    rs = np.random.RandomState(seed + 200003)
    x = rs.standard_normal((B, *S, kw["input_dim"] - 3)).astype(np.float32)
    target = rs.standard_normal((B, *S, kw["output_dim"])).astype(np.float32)
    return x, target


def make_block_io(kw: dict, seed: int, B: int, M: int, N: int):
    rs = np.random.RandomState(seed + 100003)
    x = rs.standard_normal((B, M, N, full_kwargs(kw)["input_dim"])).astype(np.float32)
    target = rs.standard_normal((B, M, N, 1)).astype(np.float32)
    return x, target


def make_spectral_io(seed: int, B: int, M: int, N: int, C: int, K: int):
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    std = math.sqrt(2.0 / (2 * C * K * 2))
    w0 = (rs.standard_normal((C, C, K, 2)) * std).astype(np.float32)
    w1 = (rs.standard_normal((C, C, K, 2)) * std).astype(np.float32)
    gy = rs.standard_normal((B, M, N, C)).astype(np.float32)
    return x, w0, w1, gy


# ---- compact storage of big gradient tensors -------------------------------------------
FULL_LIMIT = 8192
N_SAMPLES = 512


def sample_index(n: int) -> np.ndarray:
    return np.random.RandomState(12345).randint(0, n, size=N_SAMPLES)


def pack_array(name: str, a: np.ndarray, out: dict):
    a = np.asarray(a)
    if a.size <= FULL_LIMIT:
        out[name] = a
    else:
        flat = a.reshape(-1).astype(np.float64)
        out[name + "#shape"] = np.array(a.shape)
        out[name + "#l2"] = np.array(np.sqrt((flat ** 2).sum()))
        out[name + "#sum"] = np.array(flat.sum())
        out[name + "#samples"] = a.reshape(-1)[sample_index(a.size)]


def packed_names(npz) -> list:
    names = set()
    for k in npz.files:
        names.add(k.split("#")[0])
    return sorted(names)


def compare_packed(npz, name: str, got: np.ndarray, rtol: float):
    """Relative-L2 style comparison of ``got`` against a (possibly compacted) fixture entry.
    Returns the worst relative error found."""
    got = np.asarray(got, dtype=np.float64)
    if name in npz.files:
        ref = npz[name].astype(np.float64)
        assert ref.shape == got.shape, (name, ref.shape, got.shape)
        den = max(np.linalg.norm(ref), 1e-30)
        return float(np.linalg.norm(got - ref) / den)
    shape = tuple(npz[name + "#shape"])
    assert shape == got.shape, (name, shape, got.shape)
    l2 = float(npz[name + "#l2"])
    flat = got.reshape(-1)
    idx = sample_index(flat.size)
    ref_s = npz[name + "#samples"].astype(np.float64)
    scale = l2 / math.sqrt(flat.size)  # rms of the reference tensor
    e_samples = float(np.abs(flat[idx] - ref_s).max() / max(scale, 1e-30)) / 8.0  # max-vs-rms slack
    e_l2 = abs(float(np.linalg.norm(flat)) - l2) / max(l2, 1e-30)
    e_sum = abs(float(flat.sum()) - float(npz[name + "#sum"])) / max(l2 * math.sqrt(flat.size), 1e-30)
    return max(e_samples, e_l2, e_sum)


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def golden_kwargs(npz) -> dict:
    return ast.literal_eval(str(npz["kwargs"]))


def make_zongyi_state_dict(kw: dict, seed: int = 51, grid: int = 64):
    """Deterministic FNOZongyi2DBlock weights (zongyi_fno/grid_2d.py:81-117 layout) + the config-0 input batch
    [2, 64, 64, input_dim].  Fourier weights use std 0.02 (the reference's init, gain 1/(I*O), is ~1e-5 and
    would leave the spectral path numerically invisible)."""
    rs = np.random.RandomState(seed)
    W, I, K = kw["width"], kw["input_dim"], kw["modes1"]
    sd = {}

    def lin(prefix, fin, fout):
        sd[prefix + "weight"] = (rs.standard_normal((fout, fin)) / math.sqrt(fin)).astype(np.float32)
        sd[prefix + "bias"] = (rs.standard_normal(fout) * 0.1).astype(np.float32)

    lin("in_proj.", I, W)
    for l in range(kw["n_layers"]):
        pre = f"spectral_layers.{l}."
        lin(pre + "linear.", W, W)
        for j in range(2):
            sd[pre + f"fourier_weight.{j}"] = (rs.standard_normal((W, W, K, K, 2)) * 0.02).astype(np.float32)
    lin("feedforward.0.", W, 128)
    lin("feedforward.2.", 128, 1)
    x = rs.standard_normal((2, grid, grid, I)).astype(np.float32)
    return sd, x


def make_geofno_state_dict(kw: dict, seed: int):
    """Deterministic FNOMesh2D weights (zongyi_fno/mesh_2d.py:56-76 layout; convs.{i}.weights1/2 complex64).  The spectral
    weights use std 0.05 (the reference's init, scale / (in*out) * U[0,1), is ~5e-4 and would leave the spectral path
    numerically invisible next to the 1x1 convolutions)."""
    rs = np.random.RandomState(seed)
    W = kw["width"]
    modes = (kw["modes1"], kw["modes2"]) + ((kw["modes3"],) if "modes3" in kw else ())
    nd = len(modes)
    sd = {}

    def lin(prefix, fin, fout, shape=None):
        sd[prefix + "weight"] = (rs.standard_normal(shape or (fout, fin)) / math.sqrt(fin)).astype(np.float32)
        sd[prefix + "bias"] = (rs.standard_normal(fout) * 0.1).astype(np.float32)

    lin("fc0.", 4, W)
    for l in range(kw["n_layers"]):
        for j in range(1, (2 if nd == 2 else 4) + 1):
            w = rs.standard_normal((W, W, *modes, 2)) * 0.05
            sd[f"convs.{l}.weights{j}"] = (w[..., 0] + 1j * w[..., 1]).astype(np.complex64)
    for l in range(kw["n_layers"]):
        lin(f"ws.{l}.", W, W, shape=(W, W) + (1,) * nd)
    lin("fc1.", W, 128)
    lin("fc2.", 128, 1 if nd == 2 else 4)
    return sd


def make_geofno_io(seed: int, B: int, X: int, Y: int, Z: int = 0):
    rs = np.random.RandomState(seed + 100)
    if Z:       # FNOMesh3D: one input channel, four output channels
        return rs.standard_normal((B, X, Y, Z, 1)).astype(np.float32), rs.standard_normal((B, X, Y, Z, 4)).astype(np.float32)
    return rs.standard_normal((B, X, Y, 2)).astype(np.float32), rs.standard_normal((B, X, Y, 1)).astype(np.float32)


def make_cno_case(kind: str, kw: dict, seed: int, B: int, S):
    """State dict + (x, target) of a CNOFactorized* golden: the F-FNO generators with the trailing (re, im) axis of every
    Fourier weight dropped (real [in, out, modes] weights, factorized_cno/grid_2d.py:26-29) and scaled to a visible size."""
    if kind == "2d":
        sd, (x, t) = make_block_state_dict(kw, seed), make_block_io(kw, seed, B, *S)
    elif kind == "mesh2d":
        sd, (x, t) = make_mesh2d_state_dict(kw, seed), make_mesh2d_io(kw, seed, B, S)
    else:
        sd, (x, t) = make_mesh3d_state_dict(kw, seed), make_mesh3d_io(kw, seed, B, S)
    sd = {k: (np.ascontiguousarray(v[..., 0]) * 3 if "fourier_weight" in k else v) for k, v in sd.items()}
    return sd, x, t
