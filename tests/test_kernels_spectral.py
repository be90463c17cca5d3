"""Spectral kernels through the C ABI vs fp64 torch.fft / einsum references and the reference's golden
vectors -- on the CPU wave emulator (-m "not gpu") and on the MI355X (-m gpu)."""
import ctypes

import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import be, rel_l2  # noqa: F401

TOL = 1e-5  # fp32 tolerance (relative L2) stated by BASELINE.json's north_star


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


SHAPES_FWD = [(1, 8, 12, 64, 3), (2, 6, 10, 32, 5), (1, 4, 64, 64, 16), (1, 9, 7, 32, 4), (1, 4, 40, 64, 20),
              (1, 4, 64, 64, 33), (1, 130, 4, 64, 64)]


@pytest.mark.parametrize("B,M,N,C,K", SHAPES_FWD)
@pytest.mark.parametrize("axis", [0, 1])
def test_dft_fwd(be, B, M, N, C, K, axis):
    L = N if axis == 0 else M
    if K > L // 2 + 1:
        pytest.skip("modes exceed axis")
    rs = np.random.RandomState(B * 100 + M + N + C + K + axis)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    R = B * M if axis == 0 else B * N
    dx, spec, tw = be.put(x), be.empty((K, R, 2, C)), be.twiddle(L)
    assert be.lib.ffno_dft_fwd(be.ptr(dx), be.ptr(spec), be.ptr(tw), B, M, N, C, K, axis, 0, None) == 0
    got = be.get(spec)
    assert not np.isnan(got).any()
    assert rel_l2(got, ref_spec(x, K, axis)) < TOL


def test_dft_rejects_too_many_modes(be):
    x, spec, tw = be.zeros((1, 4, 8, 64)), be.zeros((6, 4, 2, 64)), be.twiddle(8)
    assert be.lib.ffno_dft_fwd(be.ptr(x), be.ptr(spec), be.ptr(tw), 1, 4, 8, 64, 6, 0, 0, None) == -3   # FFNO_EMODES
    assert be.lib.ffno_dft_fwd(be.ptr(x), be.ptr(spec), be.ptr(tw), 1, 4, 8, 48, 2, 0, 0, None) == -2   # width
    assert be.lib.ffno_dft_fwd(None, be.ptr(spec), be.ptr(tw), 1, 4, 8, 64, 2, 0, 0, None) == -1        # null


@pytest.mark.parametrize("B,M,N,C,K", [(1, 8, 12, 64, 3), (2, 6, 10, 32, 5), (1, 4, 64, 64, 16), (1, 40, 6, 64, 9),
                                       (1, 4, 64, 64, 33), (1, 100, 4, 32, 7)])
@pytest.mark.parametrize("axis", [0, 1])
def test_dft_inv_matches_zero_padded_irfft(be, B, M, N, C, K, axis):
    L = N if axis == 0 else M
    if K > L // 2 + 1:
        pytest.skip("modes exceed axis")
    rs = np.random.RandomState(7 + B + M + N + C + K + axis)
    R = B * M if axis == 0 else B * N
    spec = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    sc = torch.tensor(spec[:, :, 0] + 1j * spec[:, :, 1])          # [K,R,C]; imag of DC/Nyquist must be ignored
    if axis == 0:
        full = torch.zeros(B, M, L // 2 + 1, C, dtype=torch.complex128)
        full[:, :, :K] = sc.reshape(K, B, M, C).permute(1, 2, 0, 3)
        ref = torch.fft.irfft(full, n=L, dim=2, norm="ortho").numpy()
    else:
        full = torch.zeros(B, L // 2 + 1, N, C, dtype=torch.complex128)
        full[:, :K] = sc.reshape(K, B, N, C).permute(1, 0, 2, 3)
        ref = torch.fft.irfft(full, n=L, dim=1, norm="ortho").numpy()
    dspec, tw, out = be.put(spec), be.twiddle(L), be.empty((B, M, N, C))
    assert be.lib.ffno_dft_inv(be.ptr(dspec), be.ptr(out), None, be.ptr(tw), B, M, N, C, K, axis, 1, 0, None) == 0
    assert rel_l2(be.get(out), ref) < TOL
    resid = rs.standard_normal((B, M, N, C)).astype(np.float32)   # accumulate + residual epilogue
    dres = be.put(resid)
    assert be.lib.ffno_dft_inv(be.ptr(dspec), be.ptr(out), be.ptr(dres), be.ptr(tw), B, M, N, C, K, axis, 1, 1, None) == 0
    assert rel_l2(be.get(out), 2 * ref + resid) < TOL


@pytest.mark.parametrize("R,C,K", [(40, 64, 3), (70, 32, 2), (33, 64, 1), (300, 64, 2)])
@pytest.mark.parametrize("conj_t", [0, 1])
def test_mode_mix(be, R, C, K, conj_t):
    rs = np.random.RandomState(R + C + K + conj_t)
    w = rs.standard_normal((C, C, K, 2)).astype(np.float32)
    xs = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    dw, dxs = be.put(w), be.put(xs)
    wp, wpt, ys = be.zeros((K, 2, C, C)), be.zeros((K, 2, C, C)), be.empty((K, R, 2, C))
    assert be.lib.ffno_fw_pack(be.ptr(dw), be.ptr(wp), be.ptr(wpt), C, K, None) == 0
    np.testing.assert_array_equal(be.get(wp), w.transpose(2, 3, 0, 1))
    np.testing.assert_array_equal(be.get(wpt), w.transpose(2, 3, 1, 0))
    assert be.lib.ffno_mode_mix(be.ptr(dxs), be.ptr(wpt if conj_t else wp), be.ptr(ys), R, C, K, conj_t, None) == 0
    xc = xs[:, :, 0].astype(np.float64) + 1j * xs[:, :, 1]
    wc = w[..., 0].astype(np.float64) + 1j * w[..., 1]
    ref = np.einsum("kri,iok->kro", xc, wc) if conj_t == 0 else np.einsum("kro,iok->kri", xc, np.conj(wc))
    assert rel_l2(be.get(ys), np.stack([ref.real, ref.imag], axis=2)) < TOL


@pytest.mark.parametrize("R,C,K,nsplit", [(37, 64, 2, 3), (64, 32, 3, 4), (500, 64, 2, 5), (96, 64, 2, 4), (16, 64, 1, 3)])
def test_fw_grad(be, R, C, K, nsplit):
    rs = np.random.RandomState(R + C + K)
    xs = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    dys = rs.standard_normal((K, R, 2, C)).astype(np.float32)
    dxs, ddys = be.put(xs), be.put(dys)
    partial, gw = be.zeros((nsplit, K, 2, C, C)), be.zeros((C, C, K, 2))
    assert be.lib.ffno_fw_grad_partial(be.ptr(dxs), be.ptr(ddys), be.ptr(partial), R, C, K, nsplit, 0, 1, 0, 0, None) == 0
    assert be.lib.ffno_fw_grad_reduce(be.ptr(partial), be.ptr(gw), C, K, nsplit, 0, None) == 0
    xc = xs[:, :, 0].astype(np.float64) + 1j * xs[:, :, 1]
    dc = dys[:, :, 0].astype(np.float64) + 1j * dys[:, :, 1]
    ref = np.einsum("kri,kro->iok", np.conj(xc), dc)
    ref = np.stack([ref.real, ref.imag], axis=-1)
    assert rel_l2(be.get(gw), ref) < TOL
    assert be.lib.ffno_fw_grad_partial(be.ptr(dxs), be.ptr(ddys), be.ptr(partial), R, C, K, nsplit, 1, 1, 0, 0, None) == 0
    assert be.lib.ffno_fw_grad_reduce(be.ptr(partial), be.ptr(gw), C, K, nsplit, 1, None) == 0
    assert rel_l2(be.get(gw), 3 * ref) < TOL


@pytest.mark.parametrize("R,C,K,nl,nsplit", [(37, 64, 2, 3, 4), (16, 32, 2, 5, 3), (48, 64, 2, 3, 4), (32, 64, 1, 4, 7), (16, 64, 1, 8, 1)])
def test_fw_grad_over_layers(be, R, C, K, nl, nsplit):
    """One launch contracts over the lines of several layers (shared Fourier weights)."""
    rs = np.random.RandomState(R + nl)
    pad = 7 * 2 * C                       # layer stride larger than one layer's spectra
    stride = K * R * 2 * C + pad
    xs = rs.standard_normal((nl, stride)).astype(np.float32)
    dys = rs.standard_normal((nl, stride)).astype(np.float32)
    dxs, ddys = be.put(xs), be.put(dys)
    partial, gw = be.zeros((nsplit, K, 2, C, C)), be.zeros((C, C, K, 2))
    assert be.lib.ffno_fw_grad_partial(be.ptr(dxs), be.ptr(ddys), be.ptr(partial), R, C, K, nsplit, 0, nl, stride, stride, None) == 0
    assert be.lib.ffno_fw_grad_reduce(be.ptr(partial), be.ptr(gw), C, K, nsplit, 0, None) == 0
    x4 = xs[:, :K * R * 2 * C].reshape(nl, K, R, 2, C).astype(np.float64)
    d4 = dys[:, :K * R * 2 * C].reshape(nl, K, R, 2, C).astype(np.float64)
    xc, dc = x4[:, :, :, 0] + 1j * x4[:, :, :, 1], d4[:, :, :, 0] + 1j * d4[:, :, :, 1]
    ref = np.einsum("lkri,lkro->iok", np.conj(xc), dc)
    assert rel_l2(be.get(gw), np.stack([ref.real, ref.imag], axis=-1)) < TOL


def _amax_word(a):
    return np.array([np.abs(a).max()], np.float32).view(np.uint32)


@pytest.mark.parametrize("R,C,K,nl,nsplit,mags", [(37, 64, 2, 1, 3, (1.0,)), (48, 64, 2, 3, 4, (1.0, 1e-3, 30.0)), (64, 32, 3, 2, 4, (1e5, 1e4)),
                                                 (16, 64, 1, 8, 1, (1e-6,) * 8), (96, 64, 2, 4, 5, (1.0, 1e-9, 1.0, 1e3)),
                                                 (32, 64, 2, 4, 3, ((1e-6, 1e-3, 1.0, 1e3), (1e4, 10.0, 1e-2, 1e-5)))])
def test_fw_grad_on_split_fp16_operands(be, R, C, K, nl, nsplit, mags):
    """ffno_fw_grad_partial_h2 (round 6: three fp16 MFMAs per product, operands scaled from the range words of the tensors the spectra
    come from) against the fp64 contraction: layers of very different magnitude in one launch (one scale pair for the launch: the
    SUM is what must be right), any overall magnitude, ragged line counts; and NULL words = the bf16x3 kernel."""
    L = 64
    rs = np.random.RandomState(R + C + K + nl)
    stride = K * R * 2 * C
    # spectra of tensors with max |.| = mag: |spectrum| <= 2 sqrt(L) mag; here up to ~4 mag
    # (the last case: activations GROW and gradients SHRINK through the layers, every layer's products equally large -- one scale
    #  per operand for the whole launch would leave the small end of each operand with a dozen bits: the scales are per layer)
    xm, dm = mags if isinstance(mags[0], tuple) else (mags, mags)
    xs = np.stack([rs.standard_normal(stride).astype(np.float32) * m for m in xm])
    dys = np.stack([rs.standard_normal(stride).astype(np.float32) * m for m in dm])
    xw = np.concatenate([_amax_word(xs[l]) for l in range(nl)])       # (a valid word: max |tensor| >= max |spectrum| / (2 sqrt(L)))
    dw = np.concatenate([_amax_word(dys[l]) for l in range(nl)])
    dxs, ddys = be.put(xs), be.put(dys)
    partial, gw = be.zeros((nsplit, K, 2, C, C)), be.zeros((C, C, K, 2))
    p = be.ptr
    assert be.lib.ffno_fw_grad_partial_h2(p(dxs), p(ddys), p(partial), R, C, K, nsplit, 0, nl, stride, stride, p(be.put(xw)), p(be.put(dw)),
                                          L, None) == 0
    assert be.lib.ffno_fw_grad_reduce(p(partial), p(gw), C, K, nsplit, 0, None) == 0
    x4 = xs.reshape(nl, K, R, 2, C).astype(np.float64)
    d4 = dys.reshape(nl, K, R, 2, C).astype(np.float64)
    xc, dc = x4[:, :, :, 0] + 1j * x4[:, :, :, 1], d4[:, :, :, 0] + 1j * d4[:, :, :, 1]
    ref = np.einsum("lkri,lkro->iok", np.conj(xc), dc)
    ref = np.stack([ref.real, ref.imag], axis=-1)
    got = np.array(be.get(gw)).copy()
    assert rel_l2(got, ref) < TOL
    # NULL words: the any-range bf16x3 kernel through the same entry point
    assert be.lib.ffno_fw_grad_partial_h2(p(dxs), p(ddys), p(partial), R, C, K, nsplit, 0, nl, stride, stride, None, None, L, None) == 0
    assert be.lib.ffno_fw_grad_reduce(p(partial), p(gw), C, K, nsplit, 0, None) == 0
    assert rel_l2(be.get(gw), ref) < TOL and rel_l2(be.get(gw), got) < 1e-6


def test_fw_grad_multi_on_split_fp16_operands(be):
    """The per-layer launch (unshared Fourier weights): one scale pair per problem -- a layer 1e8 times smaller than its neighbour
    keeps its own relative accuracy."""
    R, C, K, n, nsplit, L = 32, 64, 2, 3, 2, 40
    rs = np.random.RandomState(5)
    stride = K * R * 2 * C
    mags = (1.0, 1e-8, 1e4)
    xs = np.stack([rs.standard_normal(stride).astype(np.float32) * m for m in mags])
    dys = np.stack([rs.standard_normal(stride).astype(np.float32) * m for m in mags])
    xw = np.concatenate([_amax_word(xs[l]) for l in range(n)])
    dw = np.concatenate([_amax_word(dys[l]) for l in range(n)])
    pstride = nsplit * 2 * K * C * C
    partial = be.zeros((n, pstride))
    p = be.ptr
    assert be.lib.ffno_fw_grad_partial_multi_h2(p(be.put(xs)), p(be.put(dys)), p(partial), R, C, K, nsplit, n, stride, stride, pstride,
                                                p(be.put(xw)), p(be.put(dw)), L, None) == 0
    part = np.array(be.get(partial)).reshape(n, nsplit, K, 2, C, C).sum(axis=1)       # [n][k][ri][i][o]
    for z in range(n):
        x4 = xs[z].reshape(K, R, 2, C).astype(np.float64)
        d4 = dys[z].reshape(K, R, 2, C).astype(np.float64)
        ref = np.einsum("kri,kro->kio", np.conj(x4[:, :, 0] + 1j * x4[:, :, 1]), d4[:, :, 0] + 1j * d4[:, :, 1])
        assert rel_l2(part[z][:, 0], ref.real) < TOL and rel_l2(part[z][:, 1], ref.imag) < TOL, z


@pytest.mark.parametrize("tag", ["c64_rect", "c32_odd", "tiny_lowpass_as_c32"])
def test_spectral2d_operator_vs_reference_golden(be, tag):
    """ffno_spectral2d_fwd/bwd against golden vectors of the reference's forward_fourier + autograd."""
    if tag == "tiny_lowpass_as_c32":
        pytest.skip("tiny goldens use C=4 (outside the kernel set); covered by the oracle tests")
    g = gu.load_golden("spectral_" + tag)
    B, M, N, C, K, seed = [int(v) for v in g["meta"]]
    x, w0, w1, gy = gu.make_spectral_io(seed, B, M, N, C, K)
    lib = be.lib
    ws = be.zeros(lib.ffno_spectral2d_ws_floats(B, M, N, C, K))
    twn, twm = be.twiddle(N), be.twiddle(M)
    dx, dw0, dw1, dgy, y = be.put(x), be.put(w0), be.put(w1), be.put(gy), be.empty(x.shape)
    p = be.ptr
    # round 6: the level-1 entry point runs the FUSED split kernels (one launch per axis) for every shape they take -- the
    # reference's own forward_fourier golden meets the kernels that ship
    assert lib.ffno_spectral2d_path(B, M, N, C, K) == 1
    assert lib.ffno_spectral2d_fwd(p(dx), p(dw0), p(dw1), p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 0, None) == 0
    assert gu.compare_packed(g, "y", be.get(y), TOL) < TOL
    gx, gw0, gw1 = be.empty(x.shape), be.zeros(w0.shape), be.zeros(w1.shape)
    assert lib.ffno_spectral2d_bwd(p(dx), p(dw0), p(dw1), p(dgy), p(gx), p(gw0), p(gw1), p(ws), p(twn), p(twm),
                                   B, M, N, C, K, 0, 0, 0, None) == 0
    assert gu.compare_packed(g, "gx", be.get(gx), TOL) < TOL
    assert gu.compare_packed(g, "gw0", be.get(gw0), TOL) < 2e-5
    assert gu.compare_packed(g, "gw1", be.get(gw1), TOL) < 2e-5


def _fp64_forward_fourier(x, w0, w1, K):
    """grid_2d.py:51-99 in fp64 (x [B, M, N, C]; w0 mixes the last axis, w1 the first)."""
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
    B, I, M, N = xt.shape
    wa = torch.view_as_complex(torch.tensor(w0, dtype=torch.float64).contiguous())
    wb = torch.view_as_complex(torch.tensor(w1, dtype=torch.float64).contiguous())
    fy = torch.fft.rfft(xt, dim=-1, norm="ortho")
    oy = torch.zeros(B, I, M, N // 2 + 1, dtype=torch.complex128)
    oy[..., :K] = torch.einsum("bixy,ioy->boxy", fy[..., :K], wa)
    fx = torch.fft.rfft(xt, dim=-2, norm="ortho")
    ox = torch.zeros(B, I, M // 2 + 1, N, dtype=torch.complex128)
    ox[:, :, :K] = torch.einsum("bixy,iox->boxy", fx[:, :, :K], wb)
    return (torch.fft.irfft(oy, n=N, dim=-1, norm="ortho") + torch.fft.irfft(ox, n=M, dim=-2, norm="ortho")).permute(0, 2, 3, 1).numpy()


def test_spectral2d_pack_cache_follows_the_declared_weight_version(be):
    """ffno_spectral2d_weights_version (include/ffno.h): by default every call re-packs the weights; with a declared non-zero version
    a repeated call on the same (ws, weights, shape) skips the re-pack -- shown here by changing the weights IN PLACE behind the
    library's back: same declared version -> the result still follows the OLD weights (the packs in ws were reused), a new version
    (or 0) -> the new weights."""
    lib, p = be.lib, be.ptr
    B, M, N, C, K = 1, 8, 16, 64, 4
    rs = np.random.RandomState(7)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    w0, w1 = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32), (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
    assert lib.ffno_spectral2d_path(B, M, N, C, K) == 1
    ws = be.zeros(lib.ffno_spectral2d_ws_floats(B, M, N, C, K))
    twn, twm = be.twiddle(N), be.twiddle(M)
    dx, dw0, dw1, y = be.put(x), be.put(w0), be.put(w1), be.empty(x.shape)

    def run():
        assert lib.ffno_spectral2d_fwd(p(dx), p(dw0), p(dw1), p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 0, None) == 0
        return np.array(be.get(y)).copy()

    ref_old = _fp64_forward_fourier(x, w0, w1, K)
    assert rel_l2(run(), ref_old) < TOL                       # undeclared: packs on every call
    assert lib.ffno_spectral2d_weights_version(p(ws), 41) == 0
    assert rel_l2(run(), ref_old) < TOL                       # version 41: packed for it
    # the weights change in place (same pointers); the caller has NOT declared a new version
    w0b = (w0 * 1.5).astype(np.float32)
    if be.kind == "emu":
        dw0[...] = w0b
    else:
        dw0.copy_(torch.from_numpy(w0b))
    y_stale = run()
    assert rel_l2(y_stale, ref_old) < TOL                     # the packs of version 41 were reused: no re-pack happened
    ref_new = _fp64_forward_fourier(x, w0b, w1, K)
    assert lib.ffno_spectral2d_weights_version(p(ws), 42) == 0
    assert rel_l2(run(), ref_new) < TOL                       # declared: re-packed
    assert rel_l2(run(), ref_new) < TOL
    assert lib.ffno_spectral2d_weights_version(p(ws), 0) == 0     # back to "unknown": every call packs
    assert rel_l2(run(), ref_new) < TOL
    assert lib.ffno_spectral2d_weights_version(None, 1) == -1


@pytest.mark.parametrize("B,M,N,K", [(1, 8, 32, 4), (2, 16, 64, 8)])
def test_spectral2d_forward_through_the_two_launch_pair(be, B, M, N, K):
    """Where the inference kernels take the shape (width 64, <= 16 modes, N a multiple of 32) the forward operator is two launches --
    ffno_spectral_x3_mix_pair + ffno_infer_sum: the mixed spectra between them, no branch image, the branch sum in registers --
    and the backward the fused single-axis launches; both against fp64 torch.fft + autograd."""
    lib, p = be.lib, be.ptr
    C = 64
    assert lib.ffno_spectral2d_path(B, M, N, C, K) == 1 and lib.ffno_layer_infer_supported(B, M, N, C, 4 * C, K, K) == 1
    rs = np.random.RandomState(B + M + N + K)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    gy = rs.standard_normal((B, M, N, C)).astype(np.float32)
    w0, w1 = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32), (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
    ws = be.zeros(lib.ffno_spectral2d_ws_floats(B, M, N, C, K))
    twn, twm = be.twiddle(N), be.twiddle(M)
    dx, dw0, dw1, y = be.put(x), be.put(w0), be.put(w1), be.empty(x.shape)
    assert lib.ffno_spectral2d_fwd(p(dx), p(dw0), p(dw1), p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 0, None) == 0
    assert rel_l2(be.get(y), _fp64_forward_fourier(x, w0, w1, K)) < TOL
    # backward on the same workspace (its packs are shared with the forward's)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    t0, t1 = torch.tensor(w0, dtype=torch.float64, requires_grad=True), torch.tensor(w1, dtype=torch.float64, requires_grad=True)
    xp = xt.permute(0, 3, 1, 2)
    fy = torch.fft.rfft(xp, dim=-1, norm="ortho")
    oy = torch.zeros(B, C, M, N // 2 + 1, dtype=torch.complex128)
    oy[..., :K] = torch.einsum("bixy,ioy->boxy", fy[..., :K], torch.view_as_complex(t0))
    fx = torch.fft.rfft(xp, dim=-2, norm="ortho")
    ox = torch.zeros(B, C, M // 2 + 1, N, dtype=torch.complex128)
    ox[:, :, :K] = torch.einsum("bixy,iox->boxy", fx[:, :, :K], torch.view_as_complex(t1))
    out = (torch.fft.irfft(oy, n=N, dim=-1, norm="ortho") + torch.fft.irfft(ox, n=M, dim=-2, norm="ortho")).permute(0, 2, 3, 1)
    out.backward(torch.tensor(gy, dtype=torch.float64))
    gx, gw0, gw1 = be.empty(x.shape), be.zeros(w0.shape), be.zeros(w1.shape)
    assert lib.ffno_spectral2d_bwd(p(dx), p(dw0), p(dw1), p(be.put(gy)), p(gx), p(gw0), p(gw1), p(ws), p(twn), p(twm), B, M, N, C, K, 0, 0, 0,
                                   None) == 0
    assert rel_l2(be.get(gx), xt.grad.numpy()) < TOL
    assert rel_l2(be.get(gw0), t0.grad.numpy()) < 2e-5 and rel_l2(be.get(gw1), t1.grad.numpy()) < 2e-5


def test_spectral2d_stage_path_still_serves_what_the_fused_kernels_refuse(be):
    """Width 32 with more than 16 modes: outside ffno_spectral_x3_supported -> the stage sequence, same results."""
    lib, p = be.lib, be.ptr
    B, M, N, C, K = 1, 40, 40, 32, 18
    assert lib.ffno_spectral2d_path(B, M, N, C, K) == 0
    rs = np.random.RandomState(3)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    w0, w1 = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32), (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
    ws = be.zeros(lib.ffno_spectral2d_ws_floats(B, M, N, C, K))
    twn, twm = be.twiddle(N), be.twiddle(M)
    y = be.empty(x.shape)
    assert lib.ffno_spectral2d_fwd(p(be.put(x)), p(be.put(w0)), p(be.put(w1)), p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 0, None) == 0
    assert rel_l2(be.get(y), _fp64_forward_fourier(x, w0, w1, K)) < TOL


@pytest.mark.gpu
def test_spectral2d_full_size_golden_and_properties():
    """BASELINE size (64x64, C=64, K=16): golden of the reference + size-independent properties:
    linearity in x, and low-pass idempotence (projecting twice == once when W = identity-per-mode)."""
    from backend_util import Backend
    be = Backend("gpu")
    lib, p = be.lib, be.ptr
    g = gu.load_golden("spectral_c64_k16")
    B, M, N, C, K, seed = [int(v) for v in g["meta"]]
    x, w0, w1, gy = gu.make_spectral_io(seed, B, M, N, C, K)
    ws = be.zeros(lib.ffno_spectral2d_ws_floats(B, M, N, C, K))
    twn, twm = be.twiddle(N), be.twiddle(M)
    dx, dw0, dw1, y = be.put(x), be.put(w0), be.put(w1), be.empty(x.shape)
    assert lib.ffno_spectral2d_fwd(p(dx), p(dw0), p(dw1), p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 0, None) == 0
    y1 = be.get(y).copy()
    assert gu.compare_packed(g, "y", y1, TOL) < TOL
    # linearity: F(2x + x') = 2F(x) + F(x')
    x2 = np.random.RandomState(1).standard_normal(x.shape).astype(np.float32)
    dx2, dmix = be.put(x2), be.put(2 * x + x2)
    assert lib.ffno_spectral2d_fwd(p(dx2), p(dw0), p(dw1), p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 0, None) == 0
    y2 = be.get(y).copy()
    assert lib.ffno_spectral2d_fwd(p(dmix), p(dw0), p(dw1), p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 0, None) == 0
    assert rel_l2(be.get(y), 2 * y1 + y2) < TOL
    # low-pass mode is a projection per branch: lowpass(lowpass_y-only) ... check P(P(x)) == 2 P(x) - cross terms
    # simpler exact property: for low-pass, out = Py x + Px x with Py, Px orthogonal projections, so
    # <x, out> = |Py x|^2 + |Px x|^2 >= 0 and |out| <= 2|x|.
    assert lib.ffno_spectral2d_fwd(p(dx), None, None, p(y), p(ws), p(twn), p(twm), B, M, N, C, K, 1, None) == 0
    lp = be.get(y).astype(np.float64)
    assert float((lp * x).sum()) > 0 and np.linalg.norm(lp) <= 2 * np.linalg.norm(x) * (1 + 1e-6)


@pytest.mark.parametrize("B,M,N,K,C", [(1, 8, 12, 3, 64), (2, 6, 10, 5, 64), (1, 20, 64, 16, 64), (1, 13, 9, 4, 64), (3, 5, 7, 2, 64),
                                       (2, 16, 32, 8, 64), (2, 6, 10, 5, 32), (1, 13, 40, 16, 32), (3, 5, 7, 2, 32)])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("direction", ["fwd", "adj", "lowpass"])
def test_spectral_fused_branch_equals_three_stage_path(be, B, M, N, K, C, axis, direction):
    """The fused branch kernel (DFT -> mix -> iDFT in LDS) against fp64 torch.fft, incl. the saved spectrum,
    ragged line counts (R % 8 != 0), accumulate + residual epilogue, and the adjoint configuration; widths 64 and 32."""
    L = N if axis == 0 else M
    if K > L // 2 + 1:
        pytest.skip("modes exceed axis")
    lib, p = be.lib, be.ptr
    assert lib.ffno_spectral_fused_supported(C, K, L) == 1
    rs = np.random.RandomState(B + 10 * M + 100 * N + K + axis)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
    R = B * M if axis == 0 else B * N
    dx, dw, tw = be.put(x), be.put(w), be.twiddle(L)
    wp, wpt = be.zeros((K, 2, C, C)), be.zeros((K, 2, C, C))
    assert lib.ffno_fw_pack(p(dw), p(wp), p(wpt), C, K, None) == 0
    out, spec = be.empty(x.shape), be.empty((K, R, 2, C))
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    planes = None if direction == "lowpass" else (wpt if direction == "adj" else wp)
    word = be.zeros(1, np.uint32)
    assert lib.ffno_spectral_fused(p(dx), p(out), None, p(spec), p(planes), p(tw), B, M, N, C, K, axis,
                                   fwd_ck, inv_ck, conj, 0, p(word), None) == 0
    assert np.asarray(be.get(word)).view(np.float32)[0] == np.abs(be.get(out)).max()     # the kernel recorded its output maximum
    # fp64 reference
    xt = torch.tensor(x, dtype=torch.float64)
    dim = 2 if axis == 0 else 1
    f = torch.fft.rfft(xt, dim=dim, norm="ortho").narrow(dim, 0, K)
    ck = torch.tensor([1.0 if (k == 0 or 2 * k == L) else 2.0 for k in range(K)], dtype=torch.float64)
    shape = [1, 1, 1, 1]
    shape[dim] = K
    wc = torch.tensor(w[..., 0].astype(np.float64) + 1j * w[..., 1])
    if direction == "adj":
        f = f * ck.view(shape)                                    # adjoint of the zero-padded irfft
        y = torch.einsum("bmko,iok->bmki" if axis == 0 else "bkno,iok->bkni", f, wc.conj())
    elif direction == "fwd":
        y = torch.einsum("bmki,iok->bmko" if axis == 0 else "bkni,iok->bkno", f, wc)
    else:
        y = f
    if direction == "adj":                                        # adjoint of the truncated rfft: no c_k, plain sum
        n = torch.arange(L, dtype=torch.float64)
        kk = torch.arange(K, dtype=torch.float64)
        ang = 2 * np.pi * torch.outer(n, kk) / L
        Gr, Gi = torch.cos(ang) / np.sqrt(L), -torch.sin(ang) / np.sqrt(L)
        ref = (torch.einsum("nk,bmkc->bmnc", Gr, y.real) + torch.einsum("nk,bmkc->bmnc", Gi, y.imag)) if axis == 0 \
            else (torch.einsum("nk,bknc->bnkc", Gr, y.real).permute(0, 1, 2, 3) if False else
                  torch.einsum("mk,bknc->bmnc", Gr, y.real) + torch.einsum("mk,bknc->bmnc", Gi, y.imag))
    else:
        full_shape = list(y.shape)
        full_shape[dim] = L // 2 + 1
        full = torch.zeros(full_shape, dtype=torch.complex128)
        full.narrow(dim, 0, K).copy_(y)
        ref = torch.fft.irfft(full, n=L, dim=dim, norm="ortho")
    assert rel_l2(be.get(out), ref.numpy()) < TOL
    # saved spectrum == stage-A output (with the same c_k convention)
    fs = f.permute(2, 0, 1, 3).reshape(K, R, C) if axis == 0 else f.permute(1, 0, 2, 3).reshape(K, R, C)
    assert rel_l2(be.get(spec), torch.stack([fs.real, fs.imag], dim=2).numpy()) < TOL
    # accumulate + residual epilogue, no spectrum save
    resid = rs.standard_normal(x.shape).astype(np.float32)
    dres = be.put(resid)
    assert lib.ffno_spectral_fused(p(dx), p(out), p(dres), None, p(planes), p(tw), B, M, N, C, K, axis,
                                   fwd_ck, inv_ck, conj, 1, None, None) == 0
    assert rel_l2(be.get(out), 2 * ref.numpy() + resid) < TOL


def test_spectral_fused_support_matrix(be):
    assert be.lib.ffno_spectral_fused_supported(64, 16, 64) == 1
    assert be.lib.ffno_spectral_fused_supported(64, 17, 64) == 0
    assert be.lib.ffno_spectral_fused_supported(32, 8, 64) == 1
    assert be.lib.ffno_spectral_fused_supported(32, 17, 64) == 0
    assert be.lib.ffno_spectral_fused_supported(48, 8, 64) == 0
    x, tw = be.zeros((1, 4, 64, 64)), be.twiddle(64)
    assert be.lib.ffno_spectral_fused(be.ptr(x), be.ptr(x), None, None, None, be.ptr(tw), 1, 4, 64, 64, 17, 0, 0, 1, 0, 0, None, None) == -2


@pytest.mark.parametrize("B,M,N,K,C", [(2, 10, 12, 5, 64), (1, 40, 48, 20, 64), (2, 9, 14, 4, 32), (1, 34, 36, 17, 32)])
@pytest.mark.parametrize("direction", ["fwd", "adj", "lowpass"])
def test_paired_launches_equal_the_single_branch_kernels(be, B, M, N, K, C, direction):
    """ffno_spectral_staged_pair (and, where the shape fits its LDS tile, ffno_spectral_fused_pair) run the two axes of a
    layer side by side; results must be bit-identical to the single-branch stage kernels: same code, other grid."""
    from fourierflow_amd._capi import FusedBranch
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + K)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    resid = rs.standard_normal(x.shape).astype(np.float32)
    base = rs.standard_normal(x.shape).astype(np.float32)
    dx, dres = be.put(x), be.put(resid)
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br = []
    for axis in (0, 1):
        L = N if axis == 0 else M
        R = B * M if axis == 0 else B * N
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        wp, wpt = be.zeros((K, 2, C, C)), be.zeros((K, 2, C, C))
        assert lib.ffno_fw_pack(p(be.put(w)), p(wp), p(wpt), C, K, None) == 0
        planes = None if direction == "lowpass" else (wpt if direction == "adj" else wp)
        br.append(dict(axis=axis, L=L, R=R, tw=be.twiddle(L), planes=planes, keep=(wp, wpt)))
    # reference: the single-branch stage kernels; branch 0 accumulates onto `base` with a residual, branch 1 overwrites
    refs, specs = [], []
    for i, b in enumerate(br):
        spec, mix = be.empty((K, b["R"], 2, C)), be.empty((K, b["R"], 2, C))
        out = be.put(base) if i == 0 else be.empty(x.shape)
        assert lib.ffno_dft_fwd(p(dx), p(spec), p(b["tw"]), B, M, N, C, K, b["axis"], fwd_ck, None) == 0
        src = spec
        if b["planes"] is not None:
            assert lib.ffno_mode_mix(p(spec), p(b["planes"]), p(mix), b["R"], C, K, conj, None) == 0
            src = mix
        assert lib.ffno_dft_inv(p(src), p(out), p(dres) if i == 0 else None, p(b["tw"]), B, M, N, C, K, b["axis"], inv_ck,
                                int(i == 0), None) == 0
        refs.append(be.get(out).copy())
        specs.append(be.get(spec).copy())

    def run(fn, *extra):
        outs = [be.put(base), be.empty(x.shape)]
        sv = [be.empty((K, b["R"], 2, C)) for b in br]
        args = [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"]), p(b["tw"]), B, M, N, K,
                            b["axis"], int(i == 0)) for i, b in enumerate(br)]
        assert fn(ctypes.byref(args[0]), ctypes.byref(args[1]), *extra, C, fwd_ck, inv_ck, conj, None) == 0
        return [be.get(o) for o in outs], [be.get(s_) for s_ in sv]

    mixes = [be.empty((K, b["R"], 2, C)) for b in br]
    outs, sv = run(lib.ffno_spectral_staged_pair, p(mixes[0]), p(mixes[1]))
    for i in range(2):
        np.testing.assert_array_equal(outs[i], refs[i])
        np.testing.assert_array_equal(sv[i], specs[i])
    if all(lib.ffno_spectral_fused_supported(C, K, b["L"]) for b in br):
        outs, sv = run(lib.ffno_spectral_fused_pair)
        for i in range(2):
            assert rel_l2(outs[i], refs[i]) < TOL and rel_l2(sv[i], specs[i]) < TOL
    # argument checks: shared outputs / scratch are refused
    a = FusedBranch(p(dx), p(dres), None, p(mixes[0]), None, p(br[0]["tw"]), B, M, N, K, 0, 0)
    assert lib.ffno_spectral_staged_pair(ctypes.byref(a), ctypes.byref(a), p(mixes[0]), p(mixes[1]), C, 0, 1, 0, None) == -1


# ---- the split-bf16 fused branch (spectral_x3.hip): 16 lines per workgroup, packed pre-split weights ------------------------
def _branch_reference(x, w, K, axis, direction):
    """fp64 reference of one branch (forward / adjoint / low-pass) -> (out [B,M,N,C], stage-A spectrum [K,R,2,C])."""
    B, M, N, C = x.shape
    L = N if axis == 0 else M
    R = B * M if axis == 0 else B * N
    xt = torch.tensor(x, dtype=torch.float64)
    dim = 2 if axis == 0 else 1
    f = torch.fft.rfft(xt, dim=dim, norm="ortho").narrow(dim, 0, K)
    ck = torch.tensor([1.0 if (k == 0 or 2 * k == L) else 2.0 for k in range(K)], dtype=torch.float64)
    shape = [1, 1, 1, 1]
    shape[dim] = K
    wc = torch.tensor(w[..., 0].astype(np.float64) + 1j * w[..., 1])
    if direction == "adj":
        f = f * ck.view(shape)                                    # adjoint of the zero-padded irfft
        y = torch.einsum("bmko,iok->bmki" if axis == 0 else "bkno,iok->bkni", f, wc.conj())
        n = torch.arange(L, dtype=torch.float64)
        ang = 2 * np.pi * torch.outer(n, torch.arange(K, dtype=torch.float64)) / L
        Gr, Gi = torch.cos(ang) / np.sqrt(L), -torch.sin(ang) / np.sqrt(L)
        eq = "nk,bmkc->bmnc" if axis == 0 else "mk,bknc->bmnc"
        ref = torch.einsum(eq, Gr, y.real) + torch.einsum(eq, Gi, y.imag)   # adjoint of the truncated rfft: plain sum
    else:
        y = torch.einsum("bmki,iok->bmko" if axis == 0 else "bkni,iok->bkno", f, wc) if direction == "fwd" else f
        full_shape = list(y.shape)
        full_shape[dim] = L // 2 + 1
        full = torch.zeros(full_shape, dtype=torch.complex128)
        full.narrow(dim, 0, K).copy_(y)
        ref = torch.fft.irfft(full, n=L, dim=dim, norm="ortho")
    fs = f.permute(2, 0, 1, 3).reshape(K, R, C) if axis == 0 else f.permute(1, 0, 2, 3).reshape(K, R, C)
    return ref.numpy(), torch.stack([fs.real, fs.imag], dim=2).numpy()


def _x3_pack(be, w, K, C=64, fmt=0):
    """(forward pack, adjoint pack) of a [C, C, K, 2] Fourier weight through ffno_fw_pack + ffno_spectral_x3_pack
    (fmt 1: fp16x2 planes for the mix of the fused kernel)."""
    from fourierflow_amd._capi import X3PackDesc
    lib, p = be.lib, be.ptr
    wp, wpt = be.zeros((K, 2, C, C)), be.zeros((K, 2, C, C))
    assert lib.ffno_fw_pack(p(be.put(w)), p(wp), p(wpt), C, K, None) == 0
    nbytes = int(lib.ffno_spectral_x3_pack_bytes(C, K))
    assert nbytes == K * (16 if C == 64 else 4) * 3 * 64 * 16
    pk = [be.zeros((nbytes // 4,), np.uint32) for _ in range(2)]
    descs = (X3PackDesc * 2)(X3PackDesc(p(wp), p(pk[0]), K, fmt), X3PackDesc(p(wpt), p(pk[1]), K, fmt))
    dtab = be.put(np.frombuffer(bytes(descs), dtype=np.uint8).copy())
    assert lib.ffno_spectral_x3_pack(p(dtab), 2, C, K, None) == 0
    return pk[0], pk[1], (wp, wpt, dtab)


X3_SHAPES = [(1, 8, 12, 3), (2, 6, 10, 5), (1, 20, 64, 16), (1, 13, 9, 4), (3, 5, 7, 2), (2, 16, 32, 8), (1, 3, 72, 16), (1, 100, 4, 2)]


@pytest.fixture(params=[16, 8], ids=["tile16", "tile8"])
def x3_tile(request):
    """Lines per workgroup of the fused x3 kernels, forced per call through ffno_fused_branch.tile_lines (16 or 8)."""
    return request.param


@pytest.mark.parametrize("B,M,N,K", X3_SHAPES)
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("direction", ["fwd", "adj", "lowpass"])
def test_spectral_x3_branch(be, x3_tile, B, M, N, K, axis, direction):
    if be.kind == "emu" and x3_tile == 8 and (B, M, N, K) not in ((1, 8, 12, 3), (1, 20, 64, 16), (1, 3, 72, 16)):
        pytest.skip("8-line tiles on the emulator: three shapes (the GPU run covers all)")
    if be.kind == "emu" and x3_tile == 16 and (B, M, N, K) in ((2, 6, 10, 5), (2, 16, 32, 8), (1, 100, 4, 2)):
        pytest.skip("emulator time budget: five shapes (the GPU run covers all)")
    """The split-bf16 fused branch against fp64 torch.fft at the fp32 tolerance: forward / adjoint / low-pass, the saved
    spectrum, ragged line counts (R % 16 != 0), lines longer than one 64-sample chunk (72) and odd lengths, accumulate +
    residual epilogue."""
    from fourierflow_amd._capi import FusedBranch
    C = 64
    L = N if axis == 0 else M
    if K > L // 2 + 1:
        pytest.skip("modes exceed axis")
    lib, p = be.lib, be.ptr
    assert lib.ffno_spectral_x3_supported(C, K, L) == 1
    rs = np.random.RandomState(B + 10 * M + 100 * N + K + axis)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
    R = B * M if axis == 0 else B * N
    ref, ref_spec_ = _branch_reference(x, w, K, axis, direction)
    dx, tw = be.put(x), be.twiddle(L)
    pk_f, pk_a, keep = _x3_pack(be, w, K)
    out, spec = be.empty(x.shape), be.empty((K, R, 2, C))
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    planes = None if direction == "lowpass" else (pk_a if direction == "adj" else pk_f)
    word = be.zeros(1, np.uint32)
    br = FusedBranch(p(dx), p(out), None, p(spec), p(planes), p(tw), B, M, N, K, axis, 0, 0, x3_tile, None, p(word))
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    got = be.get(out)
    assert not np.isnan(got).any()
    assert rel_l2(got, ref) < TOL
    assert rel_l2(be.get(spec), ref_spec_) < TOL
    assert np.asarray(be.get(word)).view(np.float32)[0] == np.abs(got).max()        # the branch recorded its output maximum
    resid = rs.standard_normal(x.shape).astype(np.float32)       # accumulate + residual epilogue, no spectrum save
    dres = be.put(resid)
    br = FusedBranch(p(dx), p(out), p(dres), None, p(planes), p(tw), B, M, N, K, axis, 1, 0, x3_tile)
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    assert rel_l2(be.get(out), 2 * ref + resid) < TOL


@pytest.mark.parametrize("B,M,N,K,axis", [(1, 8, 12, 3, 0), (1, 20, 64, 16, 0), (2, 16, 32, 8, 1), (32, 64, 64, 16, 1)])
@pytest.mark.parametrize("direction,mag", [("fwd", 1.0), ("adj", 1e-6), ("fwd", 1e6), ("adj", 3e5), ("fwd", 1e-9)])
def test_spectral_x3_branch_fp16x2_mix(be, x3_tile, B, M, N, K, axis, direction, mag):
    """The same branch with the per-mode channel mix on fp16x2 packs (FFNO_PLANES_FP16X2): fp32 tolerance for data of ANY
    magnitude -- O(1), a 1e-6 gradient, 1e6 activations (a spectrum of 1e6 * sqrt(L) would overflow the half format unscaled:
    VERDICT r02 weak #1) -- because the spectrum tile is held scaled by the power of two the kernel derives on the device from
    the range word of its input; the saved spectrum and the outputs come back unscaled."""
    from fourierflow_amd._capi import FusedBranch
    if be.kind == "emu" and (B > 2 or (x3_tile == 8 and K == 8) or (mag not in (1.0, 1e-6, 1e6) and K != 3)):
        pytest.skip("emulator time budget (the GPU run covers all)")
    C = 64
    L = N if axis == 0 else M
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + 10 * M + 100 * N + K + axis)
    x = (rs.standard_normal((B, M, N, C)) * mag).astype(np.float32)
    w = (rs.standard_normal((C, C, K, 2)) * 0.02).astype(np.float32)         # the magnitude of xavier-initialised weights
    R = B * M if axis == 0 else B * N
    ref, ref_spec_ = _branch_reference(x, w, K, axis, direction)
    dx, tw = be.put(x), be.twiddle(L)
    pk_f, pk_a, keep = _x3_pack(be, w, K, fmt=1)
    out, spec = be.empty(x.shape), be.empty((K, R, 2, C))
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    xw, ow = be.zeros(1, np.uint32), be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(xw), None) == 0
    pk = pk_a if direction == "adj" else pk_f
    br = FusedBranch(p(dx), p(out), None, p(spec), p(pk), p(tw), B, M, N, K, axis, 0, 1, x3_tile, p(xw), p(ow))
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    got = be.get(out)
    assert np.all(np.isfinite(got)) and rel_l2(got, ref) < TOL
    assert rel_l2(be.get(spec), ref_spec_) < TOL
    assert np.asarray(be.get(ow)).view(np.float32)[0] == np.abs(got).max()
    resid = (rs.standard_normal(x.shape) * mag).astype(np.float32)
    dres = be.put(resid)
    br = FusedBranch(p(dx), p(out), p(dres), None, p(pk), p(tw), B, M, N, K, axis, 1, 1, x3_tile, p(xw), None)
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    assert rel_l2(be.get(out), 2 * ref + resid) < TOL
    # unknown plane formats / tile sizes are refused
    bad = FusedBranch(p(dx), p(out), None, p(spec), p(pk_f), p(tw), B, M, N, K, axis, 0, 3, 0, None, None)
    assert lib.ffno_spectral_x3(ctypes.byref(bad), C, 0, 1, 0, None) == -1
    bad = FusedBranch(p(dx), p(out), None, p(spec), p(pk_f), p(tw), B, M, N, K, axis, 0, 2, 0, None, None)
    assert lib.ffno_spectral_x3(ctypes.byref(bad), C, 0, 1, 0, None) == -2      # the 16-row mix: many-mode axes with a table only
    bad = FusedBranch(p(dx), p(out), None, p(spec), p(pk_f), p(tw), B, M, N, K, axis, 0, 1, 12, None, None)
    assert lib.ffno_spectral_x3(ctypes.byref(bad), C, 0, 1, 0, None) == -1


@pytest.mark.parametrize("B,M,N,K,interleave", [(2, 10, 12, 5, 0), (2, 10, 12, 5, 1), (1, 16, 16, 8, 0), (1, 16, 16, 8, 1),
                                                (1, 40, 34, 16, 0), (1, 40, 34, 16, 1), (8, 16, 16, 8, 3), (16, 32, 32, 4, 2),
                                                (8, 24, 24, 4, 3)])
@pytest.mark.parametrize("direction", ["fwd", "adj"])
def test_spectral_x3_pair_equals_single_branches(be, x3_tile, B, M, N, K, direction, interleave):
    """Both axes of a layer in one launch (any workgroup -> branch map: contiguous, even / odd, image-local where the shapes allow
    it -- batch a multiple of 8, square images of whole tiles; (8, 24, 24) falls back with 16-line tiles): bit-identical to the
    single-branch launches."""
    if be.kind == "emu" and (B >= 16 or (B == 8 and (direction == "adj" or M == 24 or x3_tile == 8))):
        pytest.skip("emulator time budget (the GPU run covers it)")
    from fourierflow_amd._capi import FusedBranch
    C = 64
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + K)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    resid = rs.standard_normal(x.shape).astype(np.float32)
    base = rs.standard_normal(x.shape).astype(np.float32)
    dx, dres = be.put(x), be.put(resid)
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br, keep = [], []
    for axis in (0, 1):
        L = N if axis == 0 else M
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        pk_f, pk_a, kp = _x3_pack(be, w, K)
        keep.append(kp)
        br.append(dict(axis=axis, R=B * M if axis == 0 else B * N, tw=be.twiddle(L), planes=pk_a if direction == "adj" else pk_f))

    def branches(outs, sv, tile=None):
        return [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"]), p(b["tw"]), B, M, N, K,
                            b["axis"], int(i == 0), 0, x3_tile if tile is None else tile) for i, b in enumerate(br)]

    outs1, sv1 = [be.put(base), be.empty(x.shape)], [be.empty((K, b["R"], 2, C)) for b in br]
    for a in branches(outs1, sv1):
        assert lib.ffno_spectral_x3(ctypes.byref(a), C, fwd_ck, inv_ck, conj, None) == 0
    outs2, sv2 = [be.put(base), be.empty(x.shape)], [be.empty((K, b["R"], 2, C)) for b in br]
    a2 = branches(outs2, sv2)
    assert lib.ffno_spectral_x3_pair(ctypes.byref(a2[0]), ctypes.byref(a2[1]), C, fwd_ck, inv_ck, conj, interleave, None) == 0
    for i in range(2):
        np.testing.assert_array_equal(be.get(outs2[i]), be.get(outs1[i]))
        np.testing.assert_array_equal(be.get(sv2[i]), be.get(sv1[i]))
    # ... and to the other tile size (rows of the per-mode mix are independent: same products, same order)
    outs3, sv3 = [be.put(base), be.empty(x.shape)], [be.empty((K, b["R"], 2, C)) for b in br]
    a3 = branches(outs3, sv3, 24 - x3_tile)
    assert lib.ffno_spectral_x3_pair(ctypes.byref(a3[0]), ctypes.byref(a3[1]), C, fwd_ck, inv_ck, conj, interleave, None) == 0
    for i in range(2):
        np.testing.assert_array_equal(be.get(outs3[i]), be.get(outs1[i]))
        np.testing.assert_array_equal(be.get(sv3[i]), be.get(sv1[i]))
    assert lib.ffno_spectral_x3_pair(ctypes.byref(a2[0]), ctypes.byref(a2[0]), C, fwd_ck, inv_ck, conj, 0, None) == -1   # shared output


@pytest.mark.parametrize("B,M,N,K,axis", [(1, 40, 48, 20, 0), (1, 40, 48, 20, 1), (1, 70, 8, 34, 1), (2, 66, 70, 32, 0), (2, 66, 70, 32, 1),
                                          (1, 130, 136, 64, 0), (2, 256, 256, 32, 1)])
@pytest.mark.parametrize("direction", ["fwd", "adj", "lowpass"])
@pytest.mark.parametrize("fmt", [0, 1, 2], ids=["bf16x3", "fp16x2", "fp16x2-mix16"])
def test_spectral_x3_fused_many_modes(be, B, M, N, K, axis, direction, fmt):
    """17..64 modes per axis on the FUSED split kernel (spectral_x3k: four lines per workgroup, the spectrum tile in LDS) -- the
    256 x 256 regime of torus_kochkov (32 modes; the reference's own config runs 64) that used to go through three stage
    launches and HBM spectra.  Against the fp64 reference at the fp32 tolerance: forward / adjoint / low-pass, the saved
    spectrum, ragged line counts (R % 4 != 0), lengths that are not multiples of 32 or 64, all pack formats (with the range
    word for fp16x2; format 2 = the 16-row mix, which ships with the DFT-fragment table), accumulate + residual epilogue, and
    the recorded output maximum."""
    from fourierflow_amd._capi import FusedBranch
    if fmt == 2 and direction == "lowpass":
        pytest.skip("no mix in the low-pass: the pack format does not enter")
    if be.kind == "emu" and (K > 34 or B > 1 or (direction == "lowpass" and fmt) or (K == 20 and fmt == 1 and direction == "adj") or
                             (fmt == 2 and K == 34)):
        pytest.skip("emulator time budget (the GPU run covers all)")
    C = 64
    L = N if axis == 0 else M
    lib, p = be.lib, be.ptr
    assert lib.ffno_spectral_x3_supported(C, K, L) == 1
    rs = np.random.RandomState(B + 10 * M + 100 * N + K + axis)
    mag = 1.0 if fmt == 0 else 3e4          # fp16x2: a spectrum of 3e4 * sqrt(L) needs the device-side range scale
    x = (rs.standard_normal((B, M, N, C)) * mag).astype(np.float32)
    w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
    R = B * M if axis == 0 else B * N
    ref, ref_spec_ = _branch_reference(x, w, K, axis, direction)
    dx, tw = be.put(x), be.twiddle(L)
    pk_f, pk_a, keep = _x3_pack(be, w, K, fmt=fmt)
    out, spec = be.empty(x.shape), be.empty((K, R, 2, C))
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    planes = None if direction == "lowpass" else (pk_a if direction == "adj" else pk_f)
    xw, ow = be.zeros(1, np.uint32), be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(xw), None) == 0
    tab = None
    if fmt == 2:
        br = FusedBranch(p(dx), p(out), None, p(spec), p(planes), p(tw), B, M, N, K, axis, 0, fmt, 0, p(xw), p(ow))
        assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == -2      # FFNO_EUNSUPPORTED without the table
        tab = be.zeros((int(lib.ffno_spectral_x3_dft_frags_bytes(L, K)) // 4,), np.uint32)
        assert lib.ffno_spectral_x3_dft_frags(p(tw), L, K, fwd_ck, inv_ck, p(tab), None) == 0
    br = FusedBranch(p(dx), p(out), None, p(spec), p(planes), p(tw), B, M, N, K, axis, 0, fmt, 0, p(xw), p(ow), 0, 0, p(tab))
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    got = be.get(out)
    assert np.all(np.isfinite(got)) and rel_l2(got, ref) < TOL
    assert rel_l2(be.get(spec), ref_spec_) < TOL
    assert np.asarray(be.get(ow)).view(np.float32)[0] == np.abs(got).max()
    resid = (rs.standard_normal(x.shape) * mag).astype(np.float32)       # accumulate + residual epilogue, no spectrum save
    dres = be.put(resid)          # (kept alive across the call: the descriptor holds a raw pointer)
    br = FusedBranch(p(dx), p(out), p(dres), None, p(planes), p(tw), B, M, N, K, axis, 1, fmt, 0, p(xw), None, 0, 0, p(tab))
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    assert rel_l2(be.get(out), 2 * ref + resid) < TOL


@pytest.mark.parametrize("B,M,N,Ka,Kb", [(1, 40, 48, 20, 18), (1, 36, 70, 34, 17), (2, 256, 256, 32, 32), (1, 130, 136, 64, 40), (1, 44, 40, 20, 12)])
@pytest.mark.parametrize("direction", ["fwd", "adj"])
def test_spectral_x3_many_modes_dft_table_is_bit_identical(be, B, M, N, Ka, Kb, direction):
    """ffno_spectral_x3_dft_frags: the many-mode kernel with its DFT-matrix fragments LOADED from the table built once per (L, K,
    direction) against the same kernel building them from the twiddle table for every line -- outputs, saved spectra and range
    words bit for bit, single launches of both axes and the paired launch (also a pair whose axes need different tile heights,
    or one axis with <= 16 modes: the library then falls back to building on the fly for both)."""
    from fourierflow_amd._capi import FusedBranch
    if be.kind == "emu" and (B > 1 or M > 100 or Ka > 30 or (direction == "adj" and Kb > 16)):
        pytest.skip("emulator time budget (the GPU run covers all)")
    C = 64
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + Ka)
    x, resid = (rs.standard_normal((B, M, N, C)).astype(np.float32) for _ in range(2))
    dx, dres = be.put(x), be.put(resid)
    xw = be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(xw), None) == 0
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br, keep = [], []
    for axis, K in ((0, Ka), (1, Kb)):
        L = N if axis == 0 else M
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        pk_f, pk_a, kp = _x3_pack(be, w, K, fmt=1)
        tw = be.twiddle(L)
        nb = int(lib.ffno_spectral_x3_dft_frags_bytes(L, K))
        assert nb > 0
        tab = be.zeros((nb // 4,), np.uint32)
        assert lib.ffno_spectral_x3_dft_frags(p(tw), L, K, fwd_ck, inv_ck, p(tab), None) == 0
        keep += [kp, tw, tab]
        br.append(dict(axis=axis, K=K, R=B * M if axis == 0 else B * N, tw=tw, tab=tab, planes=pk_a if direction == "adj" else pk_f))
    assert lib.ffno_spectral_x3_dft_frags(p(br[0]["tw"]), N, N // 2 + 2, 0, 1, p(keep[2]), None) == -3

    def run(with_tab, paired):
        outs, sv = [be.empty(x.shape), be.empty(x.shape)], [be.empty((b["K"], b["R"], 2, C)) for b in br]
        words = [be.zeros(1, np.uint32), be.zeros(1, np.uint32)]
        ds = [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"]), p(b["tw"]), B, M, N, b["K"],
                          b["axis"], 0, 1, 0, p(xw), p(words[i]), 0, 0, p(b["tab"]) if with_tab else None) for i, b in enumerate(br)]
        if paired:
            assert lib.ffno_spectral_x3_pair(ctypes.byref(ds[0]), ctypes.byref(ds[1]), C, fwd_ck, inv_ck, conj, 3, None) == 0
        else:
            for d in ds:
                assert lib.ffno_spectral_x3(ctypes.byref(d), C, fwd_ck, inv_ck, conj, None) == 0
        return [be.get(t).copy() for t in outs + sv] + [np.asarray(be.get(t)).copy() for t in words]

    ref = run(False, False)
    assert all(np.all(np.isfinite(a)) for a in ref[:4])
    for paired in (False, True):
        for a, b in zip(run(True, paired), ref):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("B,M,N,Ka,Kb", [(1, 40, 48, 20, 18), (2, 256, 256, 32, 32), (1, 130, 136, 64, 40), (1, 36, 70, 34, 17)])
@pytest.mark.parametrize("direction", ["fwd", "adj"])
def test_spectral_x3_many_modes_mix16_pair(be, B, M, N, Ka, Kb, direction):
    """FFNO_PLANES_FP16X2_M16 (the mix of a 4-line tile on 16-row MFMA tiles: 8 live rows of 16 instead of 8 of 32): the paired
    launch equals the two single launches bit for bit (outputs, saved spectra, range words), and both stay within fp32 rounding
    of the 32-row mix on the same inputs (the products are the same; only the accumulation order inside the MFMA differs).  A
    pair whose axes need different tile heights has no common table layout: FFNO_EUNSUPPORTED, the engine keeps format 1 there."""
    from fourierflow_amd._capi import FusedBranch
    if be.kind == "emu" and (B > 1 or M > 100 or direction == "adj" or Ka > 30):
        pytest.skip("emulator time budget (the GPU run covers all)")
    C = 64
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + Ka)
    x, resid = (rs.standard_normal((B, M, N, C)).astype(np.float32) for _ in range(2))
    dx, dres = be.put(x), be.put(resid)
    xw = be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(xw), None) == 0
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br, keep = [], []
    for axis, K in ((0, Ka), (1, Kb)):
        L = N if axis == 0 else M
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        packs = {fmt: _x3_pack(be, w, K, fmt=fmt) for fmt in (1, 2)}
        tw = be.twiddle(L)
        tab = be.zeros((int(lib.ffno_spectral_x3_dft_frags_bytes(L, K)) // 4,), np.uint32)
        assert lib.ffno_spectral_x3_dft_frags(p(tw), L, K, fwd_ck, inv_ck, p(tab), None) == 0
        keep += [packs, tw, tab]
        br.append(dict(axis=axis, K=K, R=B * M if axis == 0 else B * N, tw=tw, tab=tab,
                       planes={fmt: pk[1] if direction == "adj" else pk[0] for fmt, pk in packs.items()}))
    same_tile = (Ka <= 32) == (Kb <= 32)

    def run(fmt, paired):
        outs, sv = [be.empty(x.shape), be.empty(x.shape)], [be.empty((b["K"], b["R"], 2, C)) for b in br]
        words = [be.zeros(1, np.uint32), be.zeros(1, np.uint32)]
        ds = [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"][fmt]), p(b["tw"]), B, M, N, b["K"],
                          b["axis"], 0, fmt, 0, p(xw), p(words[i]), 0, 0, p(b["tab"])) for i, b in enumerate(br)]
        if paired:
            rc = lib.ffno_spectral_x3_pair(ctypes.byref(ds[0]), ctypes.byref(ds[1]), C, fwd_ck, inv_ck, conj, 3, None)
            if fmt == 2 and not same_tile:
                assert rc == -2
                return None
            assert rc == 0
        else:
            for d in ds:
                assert lib.ffno_spectral_x3(ctypes.byref(d), C, fwd_ck, inv_ck, conj, None) == 0
        return [be.get(t).copy() for t in outs + sv] + [np.asarray(be.get(t)).copy() for t in words]

    single = run(2, False)
    assert all(np.all(np.isfinite(a)) for a in single[:4])
    pair = run(2, True)
    if pair is not None:
        for a, b in zip(pair, single):
            np.testing.assert_array_equal(a, b)
    for a, b in zip(single[:2], run(1, False)[:2]):
        assert rel_l2(a, b) < 3e-7
    for a, b in zip(single[2:4], run(1, False)[2:4]):      # saved spectra: before the mix -- the same bits
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("B,M,N,K", [(1, 8, 12, 3), (1, 64, 64, 16), (2, 6, 10, 5), (1, 3, 72, 16), (3, 5, 7, 2), (1, 13, 9, 4)])
@pytest.mark.parametrize("direction,fmt", [("fwd", 1), ("adj", 1), ("fwd", 0), ("lowpass", 1)])
def test_spectral_x3_latency_tiles_are_bit_identical(be, B, M, N, K, direction, fmt):
    """FFNO_X3_TILE_LATENCY (4-line tiles, two waves per line, a whole mode of weight fragments in flight: the rollout's batch-1
    launches) against the 8-line kernel: the same products in the same order per output element -- outputs, saved spectra and
    range words are bit-identical; single launches of both axes (with the accumulate + residual epilogue) and the paired launch."""
    from fourierflow_amd._capi import FusedBranch
    if be.kind == "emu" and ((B, M, N, K) == (1, 64, 64, 16) or ((B, M, N, K) == (2, 6, 10, 5) and (direction, fmt) != ("fwd", 1))):
        pytest.skip("emulator time budget (the GPU run covers all)")
    C = 64
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + 10 * M + 100 * N + K)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    resid = rs.standard_normal(x.shape).astype(np.float32)
    dx, dres = be.put(x), be.put(resid)
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    word_in = be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(word_in), None) == 0
    keep, br = [], {}
    for axis in (0, 1):
        L = N if axis == 0 else M
        if K > L // 2 + 1:
            pytest.skip("modes exceed axis")
        R = B * M if axis == 0 else B * N
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        pk_f, pk_a, k_ = _x3_pack(be, w, K, C, fmt)
        planes = None if direction == "lowpass" else (pk_a if direction == "adj" else pk_f)
        tw = be.twiddle(L)
        keep += [k_, pk_f, pk_a, tw]
        # (tile "1t": the latency kernel loading its DFT-matrix fragments from the table of ffno_spectral_x3_dft_frags)
        tab = be.zeros((int(lib.ffno_spectral_x3_dft_frags_bytes(L, K)) // 4,), np.uint32)
        assert lib.ffno_spectral_x3_dft_frags(p(tw), L, K, fwd_ck, inv_ck, p(tab), None) == 0
        keep.append(tab)
        got = {}
        # (tiles 2 / "2t": FFNO_X3_TILE_LATENCY_SPLIT -- two workgroups per tile, one per output-channel parity; fp16x2 mix only)
        split_ok = fmt == 1 and planes is not None
        tags = (8, 1, "1t") + ((2, "2t") if split_ok else ())
        if not split_ok:
            d = FusedBranch(p(dx), p(be.empty(x.shape)), None, None, p(planes), p(tw), B, M, N, K, axis, 0, fmt, 2, p(word_in), None)
            assert lib.ffno_spectral_x3(ctypes.byref(d), C, fwd_ck, inv_ck, conj, None) == -2
        for tag in tags:
            tile, dft = (int(tag[0]), p(tab)) if isinstance(tag, str) else (tag, None)
            out, spec, word = be.empty(x.shape), be.empty((K, R, 2, C)), be.zeros(1, np.uint32)
            d = FusedBranch(p(dx), p(out), None, p(spec), p(planes), p(tw), B, M, N, K, axis, 0, fmt, tile, p(word_in), p(word), 0, 0, dft)
            assert lib.ffno_spectral_x3(ctypes.byref(d), C, fwd_ck, inv_ck, conj, None) == 0
            first = be.get(out).copy()
            d = FusedBranch(p(dx), p(out), p(dres), None, p(planes), p(tw), B, M, N, K, axis, 1, fmt, tile, p(word_in), None, 0, 0, dft)
            assert lib.ffno_spectral_x3(ctypes.byref(d), C, fwd_ck, inv_ck, conj, None) == 0
            got[tag] = (first, be.get(spec).copy(), np.asarray(be.get(word)).copy(), be.get(out).copy())
        assert not np.isnan(got[1][0]).any()
        for tag in tags[1:]:
            for a, b in zip(got[8], got[tag]):
                np.testing.assert_array_equal(a, b)
        br[axis] = (planes, tw, got[1][0], tab, got[1][1])
    for tile in (1, 2) if split_ok else (1,):
        for with_tab in (False, True):
            oa, ob = be.empty(x.shape), be.empty(x.shape)
            sa, sb = be.empty(br[0][4].shape), be.empty(br[1][4].shape)
            da = FusedBranch(p(dx), p(oa), None, p(sa), p(br[0][0]), p(br[0][1]), B, M, N, K, 0, 0, fmt, tile, p(word_in), None, 0, 0,
                             p(br[0][3]) if with_tab else None)
            db = FusedBranch(p(dx), p(ob), None, p(sb), p(br[1][0]), p(br[1][1]), B, M, N, K, 1, 0, fmt, tile, p(word_in), None, 0, 0,
                             p(br[1][3]) if with_tab else None)
            assert lib.ffno_spectral_x3_pair(ctypes.byref(da), ctypes.byref(db), C, fwd_ck, inv_ck, conj, 3, None) == 0
            np.testing.assert_array_equal(be.get(oa), br[0][2])
            np.testing.assert_array_equal(be.get(ob), br[1][2])
            np.testing.assert_array_equal(be.get(sa), br[0][4])
            np.testing.assert_array_equal(be.get(sb), br[1][4])


@pytest.mark.parametrize("B,M,N,Ka,Kb", [(1, 40, 48, 20, 18), (1, 70, 36, 12, 34), (2, 256, 256, 32, 32), (1, 130, 136, 64, 40)])
@pytest.mark.parametrize("direction", ["fwd", "adj"])
def test_spectral_x3_fused_many_modes_pair_equals_single_branches(be, B, M, N, Ka, Kb, direction):
    """Both axes in ONE launch of the 4-line kernel (also when only one axis has more than 16 modes, and with unequal tile
    counts): bit-identical to the single-branch launches."""
    from fourierflow_amd._capi import FusedBranch
    if be.kind == "emu" and (B > 1 or M > 100 or direction == "adj"):
        pytest.skip("emulator time budget (the GPU run covers all)")
    C = 64
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + Ka)
    x, resid, base = (rs.standard_normal((B, M, N, C)).astype(np.float32) for _ in range(3))
    dx, dres = be.put(x), be.put(resid)
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br, keep = [], []
    for axis, K in ((0, Ka), (1, Kb)):
        L = N if axis == 0 else M
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        pk_f, pk_a, kp = _x3_pack(be, w, K)
        keep.append(kp)
        br.append(dict(axis=axis, K=K, R=B * M if axis == 0 else B * N, tw=be.twiddle(L), planes=pk_a if direction == "adj" else pk_f))

    def branches(outs, sv):
        return [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"]), p(b["tw"]), B, M, N, b["K"],
                            b["axis"], int(i == 0)) for i, b in enumerate(br)]

    outs1, sv1 = [be.put(base), be.empty(x.shape)], [be.empty((b["K"], b["R"], 2, C)) for b in br]
    for a in branches(outs1, sv1):
        assert lib.ffno_spectral_x3(ctypes.byref(a), C, fwd_ck, inv_ck, conj, None) == 0
    outs2, sv2 = [be.put(base), be.empty(x.shape)], [be.empty((b["K"], b["R"], 2, C)) for b in br]
    a2 = branches(outs2, sv2)
    assert lib.ffno_spectral_x3_pair(ctypes.byref(a2[0]), ctypes.byref(a2[1]), C, fwd_ck, inv_ck, conj, 1, None) == 0
    for i in range(2):
        if br[i]["K"] > 16 or max(Ka, Kb) <= 16:      # (an axis with <= 16 modes runs on another tile shape when launched alone)
            np.testing.assert_array_equal(be.get(outs2[i]), be.get(outs1[i]))
            np.testing.assert_array_equal(be.get(sv2[i]), be.get(sv1[i]))
        else:
            assert rel_l2(be.get(outs2[i]), be.get(outs1[i])) < 1e-6 and rel_l2(be.get(sv2[i]), be.get(sv1[i])) < 1e-6


@pytest.mark.parametrize("B,M,N,K", [(1, 8, 12, 3), (2, 6, 10, 5), (1, 13, 9, 4), (1, 21, 72, 8), (2, 72, 17, 8), (1, 3, 130, 16), (4, 72, 72, 8)])
@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("direction", ["fwd", "adj", "lowpass"])
@pytest.mark.parametrize("fmt", [0, 1], ids=["bf16x3", "fp16x2"])
def test_spectral_x3_width32(be, B, M, N, K, axis, direction, fmt):
    """The split fused branch at width 32 (spectral_x3c32: the 3-D mesh operators of the reference run width 32 -- BASELINE
    config 5 used to stay on the fp32-MFMA kernels, VERDICT r02 missing #2).  A wave transforms its two lines side by side as the
    two column tiles.  fp64 reference at the fp32 tolerance: forward / adjoint / low-pass, saved spectrum, odd line counts (the
    last wave's second line is dead), lengths beyond one 64-sample chunk, both pack formats (1e4-sized data with the range word
    for fp16x2), accumulate + residual, recorded output maximum."""
    from fourierflow_amd._capi import FusedBranch
    C = 32
    L = N if axis == 0 else M
    if K > L // 2 + 1:
        pytest.skip("modes exceed axis")
    if be.kind == "emu" and (B * M * N > 1600 or (fmt and direction == "lowpass")):
        pytest.skip("emulator time budget (the GPU run covers all)")
    lib, p = be.lib, be.ptr
    assert lib.ffno_spectral_x3_supported(C, K, L) == 1
    rs = np.random.RandomState(B + 10 * M + 100 * N + K + axis)
    mag = 1.0 if fmt == 0 else 1e4
    x = (rs.standard_normal((B, M, N, C)) * mag).astype(np.float32)
    w = (rs.standard_normal((C, C, K, 2)) / 6).astype(np.float32)
    R = B * M if axis == 0 else B * N
    ref, ref_spec_ = _branch_reference(x, w, K, axis, direction)
    dx, tw = be.put(x), be.twiddle(L)
    pk_f, pk_a, keep = _x3_pack(be, w, K, C=32, fmt=fmt)
    out, spec = be.empty(x.shape), be.empty((K, R, 2, C))
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    planes = None if direction == "lowpass" else (pk_a if direction == "adj" else pk_f)
    xw, ow = be.zeros(1, np.uint32), be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(xw), None) == 0
    br = FusedBranch(p(dx), p(out), None, p(spec), p(planes), p(tw), B, M, N, K, axis, 0, fmt, 0, p(xw), p(ow))
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    got = be.get(out)
    assert np.all(np.isfinite(got)) and rel_l2(got, ref) < TOL
    assert rel_l2(be.get(spec), ref_spec_) < TOL
    assert np.asarray(be.get(ow)).view(np.float32)[0] == np.abs(got).max()
    resid = (rs.standard_normal(x.shape) * mag).astype(np.float32)
    dres = be.put(resid)
    br = FusedBranch(p(dx), p(out), p(dres), None, p(planes), p(tw), B, M, N, K, axis, 1, fmt, 0, p(xw), None)
    assert lib.ffno_spectral_x3(ctypes.byref(br), C, fwd_ck, inv_ck, conj, None) == 0
    assert rel_l2(be.get(out), 2 * ref + resid) < TOL


@pytest.mark.parametrize("B,M,N,K", [(1, 10, 12, 5), (2, 9, 7, 3), (2, 72, 72, 8)])
@pytest.mark.parametrize("direction", ["fwd", "adj"])
def test_spectral_x3_width32_pair_equals_single_branches(be, B, M, N, K, direction):
    from fourierflow_amd._capi import FusedBranch
    if be.kind == "emu" and M > 20:
        pytest.skip("emulator time budget (the GPU run covers it)")
    C = 32
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + K)
    x, resid, base = (rs.standard_normal((B, M, N, C)).astype(np.float32) for _ in range(3))
    dx, dres = be.put(x), be.put(resid)
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br, keep = [], []
    for axis in (0, 1):
        L = N if axis == 0 else M
        w = (rs.standard_normal((C, C, K, 2)) / 6).astype(np.float32)
        pk_f, pk_a, kp = _x3_pack(be, w, K, C=32)
        keep.append(kp)
        br.append(dict(axis=axis, R=B * M if axis == 0 else B * N, tw=be.twiddle(L), planes=pk_a if direction == "adj" else pk_f))

    def branches(outs, sv):
        return [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"]), p(b["tw"]), B, M, N, K,
                            b["axis"], int(i == 0)) for i, b in enumerate(br)]

    outs1, sv1 = [be.put(base), be.empty(x.shape)], [be.empty((K, b["R"], 2, C)) for b in br]
    for a in branches(outs1, sv1):
        assert lib.ffno_spectral_x3(ctypes.byref(a), C, fwd_ck, inv_ck, conj, None) == 0
    outs2, sv2 = [be.put(base), be.empty(x.shape)], [be.empty((K, b["R"], 2, C)) for b in br]
    a2 = branches(outs2, sv2)
    assert lib.ffno_spectral_x3_pair(ctypes.byref(a2[0]), ctypes.byref(a2[1]), C, fwd_ck, inv_ck, conj, 1, None) == 0
    for i in range(2):
        np.testing.assert_array_equal(be.get(outs2[i]), be.get(outs1[i]))
        np.testing.assert_array_equal(be.get(sv2[i]), be.get(sv1[i]))


@pytest.mark.parametrize("B,M,N,K", [(1, 12, 16, 5), (2, 72, 10, 4), (1, 14, 72, 8)])
@pytest.mark.parametrize("direction", ["fwd", "adj"])
def test_spectral_x3_width32_dft_table_is_bit_identical(be, B, M, N, K, direction):
    """The width-32 kernel loading its DFT-matrix fragments from the table of ffno_spectral_x3_dft_frags (round 4: a wave owns two
    lines there, so the on-the-fly construction is a quarter of its vector work) against the on-the-fly path: outputs, saved
    spectra, range words bit for bit; single launches and the paired launch."""
    from fourierflow_amd._capi import FusedBranch
    if be.kind == "emu" and B * M * N > 1100:
        pytest.skip("emulator time budget (the GPU run covers it)")
    C = 32
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + K)
    x, resid = (rs.standard_normal((B, M, N, C)).astype(np.float32) for _ in range(2))
    dx, dres = be.put(x), be.put(resid)
    xw = be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(xw), None) == 0
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br, keep = [], []
    for axis in (0, 1):
        L = N if axis == 0 else M
        w = (rs.standard_normal((C, C, K, 2)) / 6).astype(np.float32)
        pk_f, pk_a, kp = _x3_pack(be, w, K, C=32, fmt=1)
        tw = be.twiddle(L)
        tab = be.zeros((int(lib.ffno_spectral_x3_dft_frags_bytes(L, K)) // 4,), np.uint32)
        assert lib.ffno_spectral_x3_dft_frags(p(tw), L, K, fwd_ck, inv_ck, p(tab), None) == 0
        keep += [kp, tw, tab]
        br.append(dict(axis=axis, R=B * M if axis == 0 else B * N, tw=tw, tab=tab, planes=pk_a if direction == "adj" else pk_f))

    def run(with_tab, paired):
        outs, sv = [be.empty(x.shape), be.empty(x.shape)], [be.empty((K, b["R"], 2, C)) for b in br]
        words = [be.zeros(1, np.uint32), be.zeros(1, np.uint32)]
        ds = [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"]), p(b["tw"]), B, M, N, K,
                          b["axis"], 0, 1, 0, p(xw), p(words[i]), 0, 0, p(b["tab"]) if with_tab else None) for i, b in enumerate(br)]
        if paired:
            assert lib.ffno_spectral_x3_pair(ctypes.byref(ds[0]), ctypes.byref(ds[1]), C, fwd_ck, inv_ck, conj, 1, None) == 0
        else:
            for d in ds:
                assert lib.ffno_spectral_x3(ctypes.byref(d), C, fwd_ck, inv_ck, conj, None) == 0
        return [be.get(t).copy() for t in outs + sv] + [np.asarray(be.get(t)).copy() for t in words]

    ref = run(False, False)
    assert all(np.all(np.isfinite(a)) for a in ref[:4])
    for paired in (False, True):
        for a, b in zip(run(True, paired), ref):
            np.testing.assert_array_equal(a, b)


def test_spectral_x3_support_matrix(be):
    lib = be.lib
    assert lib.ffno_spectral_x3_supported(64, 16, 64) == 1
    assert lib.ffno_spectral_x3_supported(64, 17, 64) == 1 and lib.ffno_spectral_x3_supported(64, 64, 256) == 1   # the 4-line tile
    assert lib.ffno_spectral_x3_supported(64, 65, 256) == 0
    assert lib.ffno_spectral_x3_supported(32, 8, 64) == 1 and lib.ffno_spectral_x3_supported(32, 17, 64) == 0     # width 32: K <= 16
    assert lib.ffno_spectral_x3_pack_bytes(32, 8) == 8 * 4 * 3 * 64 * 16 and lib.ffno_spectral_x3_pack_bytes(48, 8) == 0


@pytest.mark.parametrize("B,M,N,K", [(2, 10, 12, 5), (1, 40, 48, 20), (1, 66, 70, 32), (3, 7, 9, 3)])
@pytest.mark.parametrize("direction", ["fwd", "adj", "lowpass"])
def test_spectral_x3_staged_pair(be, B, M, N, K, direction):
    if be.kind == "emu" and (B, M, N, K) == (1, 66, 70, 32) and direction != "adj":
        pytest.skip("emulator time budget: the 32-mode case runs once on the emulator (all directions on the GPU)")
    """The split-bf16 STAGE kernels (17..32 modes: the 256 x 256 regime) as three paired launches: both axes against the
    fp64 reference, with the saved spectra, accumulate + residual on the first branch, ragged line counts and lengths."""
    from fourierflow_amd._capi import FusedBranch
    C = 64
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + K)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    resid = rs.standard_normal(x.shape).astype(np.float32)
    base = rs.standard_normal(x.shape).astype(np.float32)
    dx, dres = be.put(x), be.put(resid)
    fwd_ck, inv_ck, conj = (0, 1, 0) if direction != "adj" else (1, 0, 1)
    br, keep, refs = [], [], []
    for axis in (0, 1):
        L = N if axis == 0 else M
        assert lib.ffno_spectral_x3_staged_supported(C, K, L) == 1
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        pk_f, pk_a, kp = _x3_pack(be, w, K)
        keep.append(kp)
        planes = None if direction == "lowpass" else (pk_a if direction == "adj" else pk_f)
        br.append(dict(axis=axis, R=B * M if axis == 0 else B * N, tw=be.twiddle(L), planes=planes))
        refs.append(_branch_reference(x, w, K, axis, direction))
    outs, sv = [be.put(base), be.empty(x.shape)], [be.empty((K, b["R"], 2, C)) for b in br]
    mixes = [be.empty((K, b["R"], 2, C)) for b in br]
    args = [FusedBranch(p(dx), p(outs[i]), p(dres) if i == 0 else None, p(sv[i]), p(b["planes"]), p(b["tw"]), B, M, N, K,
                        b["axis"], int(i == 0)) for i, b in enumerate(br)]
    assert lib.ffno_spectral_x3_staged_pair(ctypes.byref(args[0]), ctypes.byref(args[1]), p(mixes[0]), p(mixes[1]), C, fwd_ck,
                                            inv_ck, conj, None) == 0
    assert rel_l2(be.get(outs[0]), base + resid + refs[0][0]) < TOL
    assert rel_l2(be.get(outs[1]), refs[1][0]) < TOL
    for i in range(2):
        assert rel_l2(be.get(sv[i]), refs[i][1]) < TOL
    assert lib.ffno_spectral_x3_staged_pair(ctypes.byref(args[0]), ctypes.byref(args[0]), p(mixes[0]), p(mixes[1]), C, 0, 1, 0, None) == -1
    assert lib.ffno_spectral_x3_staged_supported(64, 33, 128) == 0 and lib.ffno_spectral_x3_staged_supported(32, 8, 64) == 0
