"""Run the real spectral kernel sources through the CPU wave emulator and compare with the oracle."""
import numpy as np
import pytest
import torch

import golden_util as gu
from emu_util import emu_lib, ptr, rel_l2, twiddle
from oracle import ffno_oracle as orc

pytestmark = pytest.mark.emu
TOL = 1e-5


def ref_spec(x, K, axis):
    """spec[k][r][ri][c] from torch.fft (float64)."""
    xt = torch.tensor(x, dtype=torch.float64)
    B, M, N, C = x.shape
    if axis == 0:
        f = torch.fft.rfft(xt, dim=2, norm="ortho")[:, :, :K]      # [B,M,K,C]
        f = f.permute(2, 0, 1, 3).reshape(K, B * M, C)
    else:
        f = torch.fft.rfft(xt, dim=1, norm="ortho")[:, :K]         # [B,K,N,C]
        f = f.permute(1, 0, 2, 3).reshape(K, B * N, C)
    return torch.stack([f.real, f.imag], dim=2).numpy()            # [K,R,2,C]


@pytest.mark.parametrize("B,M,N,C,K", [(1, 8, 12, 64, 3), (2, 6, 10, 32, 5), (1, 4, 64, 64, 16), (1, 9, 7, 32, 4),
                                       (1, 4, 40, 64, 20)])
@pytest.mark.parametrize("axis", [0, 1])
def test_dft_fwd(B, M, N, C, K, axis):
    L = N if axis == 0 else M
    if K > L // 2 + 1:
        pytest.skip("modes exceed axis")
    lib = emu_lib()
    rs = np.random.RandomState(B * 100 + M + N + C + K + axis)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    R = B * M if axis == 0 else B * N
    spec = np.full((K, R, 2, C), np.nan, np.float32)
    tw = twiddle(L)
    assert lib.ffno_dft_fwd(ptr(x), ptr(spec), ptr(tw), B, M, N, C, K, axis, 0, None) == 0
    ref = ref_spec(x, K, axis)
    assert not np.isnan(spec).any()
    assert rel_l2(spec, ref) < TOL


@pytest.mark.parametrize("B,M,N,C,K", [(1, 8, 12, 64, 3), (2, 6, 10, 32, 5), (1, 4, 64, 64, 16), (1, 40, 6, 64, 9)])
@pytest.mark.parametrize("axis", [0, 1])
def test_dft_inv_roundtrip_matches_irfft(B, M, N, C, K, axis):
    L = N if axis == 0 else M
    if K > L // 2 + 1:
        pytest.skip("modes exceed axis")
    lib = emu_lib()
    rs = np.random.RandomState(7 + B + M + N + C + K + axis)
    R = B * M if axis == 0 else B * N
    spec = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    # reference: zero-padded irfft (imag of DC ignored by C2R)
    sc = torch.tensor(spec[:, :, 0] + 1j * spec[:, :, 1])          # [K,R,C]
    if axis == 0:
        full = torch.zeros(B, M, L // 2 + 1, C, dtype=torch.complex128)
        full[:, :, :K] = sc.reshape(K, B, M, C).permute(1, 2, 0, 3)
        ref = torch.fft.irfft(full, n=L, dim=2, norm="ortho").numpy()
    else:
        full = torch.zeros(B, L // 2 + 1, N, C, dtype=torch.complex128)
        full[:, :K] = sc.reshape(K, B, N, C).permute(1, 0, 2, 3)
        ref = torch.fft.irfft(full, n=L, dim=1, norm="ortho").numpy()
    tw = twiddle(L)
    out = np.full((B, M, N, C), np.nan, np.float32)
    assert lib.ffno_dft_inv(ptr(spec), ptr(out), None, ptr(tw), B, M, N, C, K, axis, 1, 0, None) == 0
    assert rel_l2(out, ref) < TOL
    # accumulate + residual
    resid = rs.standard_normal((B, M, N, C)).astype(np.float32)
    out2 = out.copy()
    assert lib.ffno_dft_inv(ptr(spec), ptr(out2), ptr(resid), ptr(tw), B, M, N, C, K, axis, 1, 1, None) == 0
    assert rel_l2(out2, 2 * ref + resid) < TOL


@pytest.mark.parametrize("R,C,K", [(40, 64, 3), (70, 32, 2), (33, 64, 1)])
@pytest.mark.parametrize("conj_t", [0, 1])
def test_mode_mix(R, C, K, conj_t):
    lib = emu_lib()
    rs = np.random.RandomState(R + C + K + conj_t)
    w = rs.standard_normal((C, C, K, 2)).astype(np.float32)
    xs = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    wp = np.zeros((K, 2, C, C), np.float32)
    wpt = np.zeros((K, 2, C, C), np.float32)
    assert lib.ffno_fw_pack(ptr(w), ptr(wp), ptr(wpt), C, K, None) == 0
    np.testing.assert_array_equal(wp, w.transpose(2, 3, 0, 1))
    np.testing.assert_array_equal(wpt, w.transpose(2, 3, 1, 0))
    ys = np.full((K, R, 2, C), np.nan, np.float32)
    assert lib.ffno_mode_mix(ptr(xs), ptr(wpt if conj_t else wp), ptr(ys), R, C, K, conj_t, None) == 0
    xc = xs[:, :, 0].astype(np.float64) + 1j * xs[:, :, 1]
    wc = w[..., 0].astype(np.float64) + 1j * w[..., 1]
    if conj_t == 0:
        ref = np.einsum("kri,iok->kro", xc, wc)
    else:
        ref = np.einsum("kro,iok->kri", xc, np.conj(wc))
    ref = np.stack([ref.real, ref.imag], axis=2)
    assert rel_l2(ys, ref) < TOL


@pytest.mark.parametrize("R,C,K,nsplit", [(37, 64, 2, 3), (64, 32, 3, 4)])
def test_fw_grad(R, C, K, nsplit):
    lib = emu_lib()
    rs = np.random.RandomState(R + C + K)
    xs = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    dys = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    partial = np.zeros((nsplit, K, 2, C, C), np.float32)
    gw = np.zeros((C, C, K, 2), np.float32)
    assert lib.ffno_fw_grad_partial(ptr(xs), ptr(dys), ptr(partial), R, C, K, nsplit, 0, None) == 0
    assert lib.ffno_fw_grad_reduce(ptr(partial), ptr(gw), C, K, nsplit, 0, None) == 0
    xc = xs[:, :, 0].astype(np.float64) + 1j * xs[:, :, 1]
    dc = dys[:, :, 0].astype(np.float64) + 1j * dys[:, :, 1]
    ref = np.einsum("kri,kro->iok", np.conj(xc), dc)
    ref = np.stack([ref.real, ref.imag], axis=-1)
    assert rel_l2(gw, ref) < TOL
    # beta / accumulate paths
    assert lib.ffno_fw_grad_partial(ptr(xs), ptr(dys), ptr(partial), R, C, K, nsplit, 1, None) == 0
    assert lib.ffno_fw_grad_reduce(ptr(partial), ptr(gw), C, K, nsplit, 1, None) == 0
    assert rel_l2(gw, 3 * ref) < TOL


@pytest.mark.parametrize("tag", ["c64_rect", "c32_odd"])
def test_spectral2d_operator_vs_golden(tag):
    """Operator-level entry points against the reference's golden vectors (fwd + bwd)."""
    lib = emu_lib()
    g = gu.load_golden("spectral_" + tag)
    B, M, N, C, K, seed = [int(v) for v in g["meta"]]
    x, w0, w1, gy = gu.make_spectral_io(seed, B, M, N, C, K)
    ws = np.zeros(lib.ffno_spectral2d_ws_floats(B, M, N, C, K), np.float32)
    twn, twm = twiddle(N), twiddle(M)
    y = np.full_like(x, np.nan)
    assert lib.ffno_spectral2d_fwd(ptr(x), ptr(w0), ptr(w1), ptr(y), ptr(ws), ptr(twn), ptr(twm), B, M, N, C, K, 0, None) == 0
    assert gu.compare_packed(g, "y", y, TOL) < TOL
    gx = np.full_like(x, np.nan)
    gw0, gw1 = np.zeros_like(w0), np.zeros_like(w1)
    assert lib.ffno_spectral2d_bwd(ptr(x), ptr(w0), ptr(w1), ptr(gy), ptr(gx), ptr(gw0), ptr(gw1), ptr(ws), ptr(twn),
                                   ptr(twm), B, M, N, C, K, 0, 0, 0, None) == 0
    assert gu.compare_packed(g, "gx", gx, TOL) < TOL
    assert gu.compare_packed(g, "gw0", gw0, TOL) < 2e-5
    assert gu.compare_packed(g, "gw1", gw1, TOL) < 2e-5
