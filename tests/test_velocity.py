"""Velocity features (reference routines/grid_2d_markov.py:82-94,130-144): the oracle against analytic
known answers and vector-calculus identities (the reference routine itself needs jax and cannot be run here), and the HIP
kernels against the oracle."""
import math

import numpy as np
import pytest
import torch

from backend_util import be, host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc


def grid(X, Y):
    x = torch.arange(X, dtype=torch.float64) * (2 * math.pi / X)
    y = torch.arange(Y, dtype=torch.float64) * (2 * math.pi / Y)
    return torch.meshgrid(x, y, indexing="ij")


@pytest.mark.parametrize("a,b", [(1, 0), (0, 2), (3, 5), (2, -3)])
def test_oracle_plane_wave_known_answer(a, b):
    """w = cos(a x + b y):  psi = w / (a^2 + b^2),  u = psi_y = -b sin(.)/(a^2+b^2),  v = -psi_x = a sin(.)/(a^2+b^2)."""
    X, Y = 16, 24
    gx, gy = grid(X, Y)
    ph = a * gx + b * gy
    w = torch.cos(ph)[None, :, :, None]
    out = orc.velocity_features(w)
    k2 = a * a + b * b
    np.testing.assert_allclose(out[0, :, :, 0].numpy(), torch.cos(ph).numpy(), atol=1e-12)
    np.testing.assert_allclose(out[0, :, :, 1].numpy(), (-b * torch.sin(ph) / k2).numpy(), atol=1e-12)
    np.testing.assert_allclose(out[0, :, :, 2].numpy(), (a * torch.sin(ph) / k2).numpy(), atol=1e-12)


def test_oracle_curl_recovers_vorticity_and_flow_is_divergence_free():
    X, Y = 32, 32
    rs = np.random.RandomState(0)
    w = torch.tensor(rs.standard_normal((2, X, Y, 1)))
    w = w - w.mean(dim=(1, 2), keepdim=True)
    # band-limit below Nyquist so spectral derivatives are exact
    wh = torch.fft.fftn(w, dim=[1, 2])
    m = torch.fft.fftfreq(X, 1 / X).abs()[:, None] < X // 2 - 1
    n = torch.fft.fftfreq(Y, 1 / Y).abs()[None, :] < Y // 2 - 1
    w = torch.fft.ifftn(wh * (m & n)[None, :, :, None], dim=[1, 2]).real
    out = orc.velocity_features(w)
    u, v = out[..., 1], out[..., 2]

    def ddx(f, dim, n):
        k = torch.fft.fftfreq(n, 1 / n)
        shape = [1, n, 1] if dim == 1 else [1, 1, n]
        return torch.fft.ifft(torch.fft.fft(f, dim=dim) * (1j * k).reshape(shape), dim=dim).real

    curl = ddx(v, 1, X) - ddx(u, 2, Y)
    div = ddx(u, 1, X) + ddx(v, 2, Y)
    assert rel_l2(curl.numpy(), w[..., 0].numpy()) < 1e-6        # the wavenumber buffers are float32 / complex64 (:92-94)
    assert float(div.abs().max()) < 1e-6


def test_wavenumber_buffers_have_the_reference_layout():
    kx, ky, lap = orc.velocity_wavenumbers(64, 64)
    assert kx.shape == ky.shape == lap.shape == (64, 33) and lap.dtype == torch.complex64
    assert lap[0, 0] == 1 and abs(kx[1, 0].item() - 1 / (2 * math.pi)) < 1e-7 and abs(kx[32, 0].item() + 32 / (2 * math.pi)) < 1e-5
    assert abs(ky[0, 32].item() - 32 / (2 * math.pi)) < 1e-5 and abs(lap[1, 2].real.item() + 5.0) < 1e-5


@pytest.mark.parametrize("B,X,Y,lx,ly", [(2, 16, 12, 2 * math.pi, 2 * math.pi), (1, 64, 64, 2 * math.pi, 2 * math.pi),
                                         (3, 10, 8, 1.0, 3.0), (2, 256, 256, 2 * math.pi, 2 * math.pi)])
def test_velocity_kernel_matches_oracle(be, B, X, Y, lx, ly):
    if be.kind == "emu" and X > 64:
        pytest.skip("large case runs on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(X + Y)
    w = rs.standard_normal((B, X, Y)).astype(np.float32)
    dw, out = be.put(w), be.empty((B, X, Y, 3))
    ws = be.zeros(lib.ffno_velocity_ws_floats(B, X, Y))
    assert lib.ffno_velocity_features(p(dw), p(out), p(ws), B, X, Y, lx, ly, None) == 0
    ref = orc.velocity_features(torch.tensor(w, dtype=torch.float64)[..., None], domain=((0.0, lx), (0.0, ly))).numpy()
    got = be.get(out)
    np.testing.assert_array_equal(got[..., 0], w)
    assert rel_l2(got[..., 1], ref[..., 1]) < 1e-5
    assert rel_l2(got[..., 2], ref[..., 2]) < 1e-5


def test_velocity_rejects_bad_arguments(be):
    z = be.zeros(64)
    p = be.ptr
    assert be.lib.ffno_velocity_features(p(z), p(z), p(z), 1, 4, 5, 1.0, 1.0, None) == -2     # odd last axis
    assert be.lib.ffno_velocity_features(None, p(z), p(z), 1, 4, 4, 1.0, 1.0, None) == -1
    assert be.lib.ffno_velocity_features(p(z), p(z), p(z), 1, 4, 4, 0.0, 1.0, None) == -1


# ---- pinned by a reference run: tests/golden/markov_velocity.npz holds the outputs of the reference's OWN
# Grid2DMarkovExperiment._build_features (routines/grid_2d_markov.py:124-170, executed by tools/make_golden_velocity.py with the
# wavenumber buffers rebuilt from jax_cfd's published rfft_mesh definition): vorticity -> (w, u, v) [-> + position] -> running
# normaliser, two consecutive accumulating calls.
import os

import golden_util as gu

_VEL = os.path.join(gu.GOLDEN_DIR, "markov_velocity.npz")


def _vel_cases():
    g = np.load(_VEL)
    for tag in ("a", "b", "c"):
        size, B, use_pos, norm = [int(v) for v in g[f"{tag}.meta"]]
        yield tag, g, size, B, bool(use_pos), bool(norm), tuple(map(tuple, g[f"{tag}.domain"])), tuple(g[f"{tag}.lowhigh"])


def test_oracle_velocity_and_features_match_the_reference_run():
    for tag, g, size, B, use_pos, norm, domain, (low, high) in _vel_cases():
        D = 3 + (2 if use_pos else 0)
        st = orc.NormalizerState(D, max_accumulations=1000) if norm else None
        for i in (0, 1):
            x = torch.from_numpy(g[f"{tag}.x{i}"])
            feats = orc.markov_features(orc.velocity_features(x, domain), st, None, 0.0, low=low, high=high, training=True,
                                        use_position=use_pos)
            assert rel_l2(feats.numpy(), g[f"{tag}.f{i}"]) < 2e-6, (tag, i)
        if norm:
            assert rel_l2(st.sum.numpy(), g[f"{tag}.norm_sum"]) < 1e-5 and float(st.count) == float(g[f"{tag}.norm_count"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_routine_features_match_the_reference_run(host_device, tag):
    """Grid2DMarkovExperiment._build_features on the HIP path (velocity kernels + fused feature / normaliser kernel)."""
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    case = {c[0]: c for c in _vel_cases()}[tag]
    _, g, size, B, use_pos, norm, domain, (low, high) = case
    if host_device == "cpu" and size > 32:
        pytest.skip("64 x 64 case runs on the GPU")
    D = 3 + (2 if use_pos else 0)
    blk = FNOFactorized2DBlock(modes=4, width=32, n_layers=1, input_dim=D, factor=2)
    exp = Grid2DMarkovExperiment(blk, use_velocity=True, use_position=use_pos, should_normalize=norm, low=low, high=high,
                                 domain=domain, grid_size=[size], max_accumulations=1000, noise_std=0.0).to(host_device)
    exp.train()
    for i in (0, 1):
        feats = exp._build_features({"x": torch.from_numpy(g[f"{tag}.x{i}"].copy()).to(host_device)})
        assert rel_l2(feats.cpu().numpy(), g[f"{tag}.f{i}"]) < 1e-5, (tag, i)
    if norm:
        assert rel_l2(exp.normalizer.sum.cpu().numpy(), g[f"{tag}.norm_sum"]) < 1e-5
        assert float(exp.normalizer.count.item()) == float(g[f"{tag}.norm_count"])
