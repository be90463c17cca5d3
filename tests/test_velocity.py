"""Velocity features (reference routines/grid_2d_markov.py:82-94,130-144): the oracle against analytic
known answers and vector-calculus identities (the reference routine itself needs jax and cannot be run here), and the HIP
kernels against the oracle."""
import math

import numpy as np
import pytest
import torch

from backend_util import be, rel_l2  # noqa: F401
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
