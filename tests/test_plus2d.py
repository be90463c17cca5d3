"""FNOPlus2DBlock (the non-factorized "fno++" ablation, zongyi_fno/grid_plus_2d.py): the new first-axis
complex DFT / weight layout kernels against numpy, the oracle vs the reference's golden vectors, and the HIP block vs both."""
import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import be, host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc

TAGS = ["c32_small", "c64_shared"]


@pytest.mark.parametrize("B,M,C,K", [(2, 10, 32, 3), (1, 16, 64, 8), (3, 7, 32, 2), (1, 256, 64, 32)])
def test_cdft_rows_forward_inverse_and_adjoint(be, B, M, C, K):
    if be.kind == "emu" and M > 64:
        pytest.skip("large case (dynamic LDS beyond 64 KB) runs on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(M + K)
    S = rs.standard_normal((K, B, M, 2, C)).astype(np.float32)          # [ky][b][m][re/im][c]
    Z = be.empty((K, 2 * K, B, 2, C))
    assert lib.ffno_cdft_rows(p(be.put(S)), p(Z), B, M, C, K, 0, None) == 0
    Sc = S[:, :, :, 0].astype(np.float64) + 1j * S[:, :, :, 1]           # [ky][b][m][c]
    rows = np.r_[0:K, M - K:M]
    full = np.fft.fft(Sc, axis=2, norm="ortho")[:, :, rows]              # [ky][b][kx'][c]
    ref = np.stack([full.real, full.imag], axis=3).transpose(0, 2, 1, 3, 4)   # [ky][kx'][b][ri][c]
    assert rel_l2(be.get(Z), ref) < 1e-5
    # zero-padded inverse
    Zin = rs.standard_normal((K, 2 * K, B, 2, C)).astype(np.float32)
    Sout = be.empty((K, B, M, 2, C))
    assert lib.ffno_cdft_rows(p(be.put(Zin)), p(Sout), B, M, C, K, 1, None) == 0
    Zc = (Zin[:, :, :, 0].astype(np.float64) + 1j * Zin[:, :, :, 1]).transpose(0, 2, 1, 3)   # [ky][b][kx'][c]
    pad = np.zeros((K, B, M, C), np.complex128)
    pad[:, :, rows] = Zc
    inv = np.fft.ifft(pad, axis=2, norm="ortho")
    assert rel_l2(be.get(Sout), np.stack([inv.real, inv.imag], axis=3)) < 1e-5
    # <forward(S), Zin> == <S, inverse(Zin)> : the inverse is the adjoint of the forward
    assert abs((be.get(Z).astype(np.float64) * Zin).sum() - (S.astype(np.float64) * be.get(Sout)).sum()) < 1e-3
    assert lib.ffno_cdft_rows(p(be.put(S)), p(Z), B, 2 * K - 1, C, K, 0, None) == -3     # 2K > M


def test_fw2d_pack_and_grad_reduce_layouts(be):
    lib, p = be.lib, be.ptr
    C, K, nsplit = 32, 3, 2
    rs = np.random.RandomState(1)
    w0, w1 = (rs.standard_normal((C, C, K, K, 2)).astype(np.float32) for _ in range(2))
    wp, wpt = be.empty((K * 2 * K, 2, C, C)), be.empty((K * 2 * K, 2, C, C))
    assert lib.ffno_fw2d_pack(p(be.put(w0)), p(be.put(w1)), p(wp), p(wpt), C, K, None) == 0
    W = np.concatenate([w0, w1], axis=2)                                 # [i][o][kx'][ky][ri]
    ref = W.transpose(3, 2, 4, 0, 1).reshape(K * 2 * K, 2, C, C)         # [ky][kx'][ri][i][o]
    np.testing.assert_array_equal(be.get(wp), ref)
    np.testing.assert_array_equal(be.get(wpt), ref.transpose(0, 1, 3, 2))
    part = rs.standard_normal((nsplit, K * 2 * K, 2, C, C)).astype(np.float32)
    g0, g1 = be.put(np.ones_like(w0)), be.put(np.ones_like(w1))
    assert lib.ffno_fw2d_grad_reduce(p(be.put(part)), p(g0), p(g1), C, K, nsplit, 1, None) == 0
    tot = part.sum(0).reshape(K, 2 * K, 2, C, C).transpose(3, 4, 1, 0, 2)   # [i][o][kx'][ky][ri]
    assert rel_l2(be.get(g0), 1 + tot[:, :, :K]) < 1e-6 and rel_l2(be.get(g1), 1 + tot[:, :, K:]) < 1e-6


def oracle_run(kw, seed, B, M, N, dtype=torch.float32):
    import oracle_util as ou
    sd, uniq = ou.torch_state_dict(gu.make_block_state_dict(kw, seed, plus=True), dtype)
    x_np, t_np = gu.make_block_io(kw, seed, B, M, N)
    out = orc.ffno2d_block(sd, torch.tensor(x_np, dtype=dtype), modes=kw["modes"], n_layers=kw["n_layers"],
                           spectral="plus")["forecast"]
    loss = orc.lp_rel_loss(out, torch.tensor(t_np, dtype=dtype))
    loss.backward()
    return out, loss, {k: p.grad.detach().numpy() for k, p in uniq.items()}


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_plus_matches_reference_golden(tag):
    g = gu.load_golden("plus_" + tag)
    kw = gu.golden_kwargs(g)
    B, M, N, seed = [int(v) for v in g["meta"]]
    out, loss, grads = oracle_run(kw, seed, B, M, N)
    assert gu.compare_packed(g, "forecast", out.detach().numpy(), 2e-5) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    for n in [k for k in gu.packed_names(g) if k.startswith("grad.")]:
        assert gu.compare_packed(g, n, grads[n[5:]], 2e-5) < 5e-5, n


@pytest.mark.parametrize("tag", TAGS)
def test_plus_hip_path_matches_reference_golden(tag, host_device):
    from fourierflow_amd.modules import FNOPlus2DBlock
    g = gu.load_golden("plus_" + tag)
    kw = gu.golden_kwargs(g)
    B, M, N, seed = [int(v) for v in g["meta"]]
    blk = FNOPlus2DBlock(**kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in gu.make_block_state_dict(kw, seed, plus=True).items()}
    assert set(blk.state_dict().keys()) == set(sd.keys())
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(host_device)
    x_np, t_np = gu.make_block_io(kw, seed, B, M, N)
    out = blk(torch.from_numpy(x_np).to(host_device))["forecast"]
    assert gu.compare_packed(g, "forecast", out.detach().cpu().numpy(), 1e-5) < 1e-5
    loss = orc.lp_rel_loss(out, torch.from_numpy(t_np).to(host_device))
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    named = dict(blk.named_parameters())
    errs = {n: gu.compare_packed(g, n, named[n[5:]].grad.cpu().numpy(), 1e-5)
            for n in gu.packed_names(g) if n.startswith("grad.")}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 3e-3, (worst, errs[worst])       # ReLU bit-flip discontinuity, see tests/test_block.py
    assert float(np.median(list(errs.values()))) < 3e-4, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.gpu
def test_plus_kochkov_ablation_shape_on_gpu():
    """torus_kochkov/ffno/ablation/fno++ shape (64 x 64, modes 16, width 64): forward vs the oracle, two layers."""
    from fourierflow_amd.modules import FNOPlus2DBlock
    import oracle_util as ou
    kw = dict(modes=16, width=64, n_layers=2, input_dim=5, share_weight=False, factor=4, ff_weight_norm=True, gain=0.5)
    seed, B, M, N = 8, 4, 64, 64
    sd_np = gu.make_block_state_dict(kw, seed, plus=True)
    blk = FNOPlus2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    blk = blk.cuda()
    x_np, _ = gu.make_block_io(kw, seed, B, M, N)
    with torch.no_grad():
        out = blk(torch.from_numpy(x_np).cuda())["forecast"]
        sd, _ = ou.torch_state_dict(sd_np, torch.float32, requires_grad=False)
        ref = orc.ffno2d_block(sd, torch.from_numpy(x_np), modes=16, n_layers=2, spectral="plus")["forecast"]
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("B,M,C,Kx,Ky", [(2, 12, 32, 3, 5), (1, 20, 64, 6, 2)])
def test_rectangular_mode_blocks(be, B, M, C, Kx, Ky):
    """modes1 != modes2 (FNOMesh2D, zongyi_fno/mesh_2d.py:46-49): Kx retained rows per corner block, Ky retained columns."""
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(M + Kx)
    S = rs.standard_normal((Ky, B, M, 2, C)).astype(np.float32)          # [ky][b][m][re/im][c]
    Z = be.empty((Ky, 2 * Kx, B, 2, C))
    hS = be.put(S)
    assert lib.ffno_cdft_rows2(p(hS), p(Z), B, M, C, Kx, Ky, 0, None) == 0
    Sc = S[:, :, :, 0].astype(np.float64) + 1j * S[:, :, :, 1]
    rows = np.r_[0:Kx, M - Kx:M]
    full = np.fft.fft(Sc, axis=2, norm="ortho")[:, :, rows]              # [ky][b][kx'][c]
    ref = np.stack([full.real, full.imag], axis=3).transpose(0, 2, 1, 3, 4)
    assert rel_l2(be.get(Z), ref) < 1e-5
    Zin = rs.standard_normal((Ky, 2 * Kx, B, 2, C)).astype(np.float32)
    Sout, hZ = be.empty((Ky, B, M, 2, C)), be.put(Zin)
    assert lib.ffno_cdft_rows2(p(hZ), p(Sout), B, M, C, Kx, Ky, 1, None) == 0
    Zc = (Zin[:, :, :, 0].astype(np.float64) + 1j * Zin[:, :, :, 1]).transpose(0, 2, 1, 3)
    pad = np.zeros((Ky, B, M, C), np.complex128)
    pad[:, :, rows] = Zc
    inv = np.fft.ifft(pad, axis=2, norm="ortho")
    assert rel_l2(be.get(Sout), np.stack([inv.real, inv.imag], axis=3)) < 1e-5
    # weight layouts [i][o][Kx][Ky][2]
    w0, w1 = (rs.standard_normal((C, C, Kx, Ky, 2)).astype(np.float32) for _ in range(2))
    wp, wpt = be.empty((Ky * 2 * Kx, 2, C, C)), be.empty((Ky * 2 * Kx, 2, C, C))
    h0, h1 = be.put(w0), be.put(w1)
    assert lib.ffno_fw2d_pack2(p(h0), p(h1), p(wp), p(wpt), C, Kx, Ky, None) == 0
    W = np.concatenate([w0, w1], axis=2)                                 # [i][o][kx'][ky][ri]
    refp = W.transpose(3, 2, 4, 0, 1).reshape(Ky * 2 * Kx, 2, C, C)      # [ky][kx'][ri][i][o]
    np.testing.assert_array_equal(be.get(wp), refp)
    np.testing.assert_array_equal(be.get(wpt), refp.transpose(0, 1, 3, 2))
    part = rs.standard_normal((2, Ky * 2 * Kx, 2, C, C)).astype(np.float32)
    g0, g1, hp = be.put(np.ones_like(w0)), be.put(np.ones_like(w1)), be.put(part)
    assert lib.ffno_fw2d_grad_reduce2(p(hp), p(g0), p(g1), C, Kx, Ky, 2, 1, None) == 0
    tot = part.sum(0).reshape(Ky, 2 * Kx, 2, C, C).transpose(3, 4, 1, 0, 2)
    assert rel_l2(be.get(g0), 1 + tot[:, :, :Kx]) < 1e-6 and rel_l2(be.get(g1), 1 + tot[:, :, Kx:]) < 1e-6


@pytest.mark.parametrize("B,M,C,Kx,Ky", [(2, 12, 32, 3, 5), (1, 20, 64, 6, 2), (2, 8, 32, 4, 3), (1, 13, 64, 5, 4), (3, 9, 32, 2, 2)])
def test_cdft_rows_on_the_matrix_cores_equals_the_vector_version(be, B, M, C, Kx, Ky):
    """ffno_cdft_rows_mfma (two truncated real DFTs + an element-wise combination) vs ffno_cdft_rows2, forward and
    zero-padded inverse, incl. 2 Kx == M (the Nyquist row) and odd M; and inverse == adjoint of forward."""
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(M + Kx + Ky)
    S = rs.standard_normal((Ky, B, M, 2, C)).astype(np.float32)
    Zin = rs.standard_normal((Ky, 2 * Kx, B, 2, C)).astype(np.float32)
    hS, hZ, tw = be.put(S), be.put(Zin), be.twiddle(M)
    ws = be.empty(int(lib.ffno_cdft_rows_ws_floats(B, C, Kx, Ky)))
    Zv, Zm = be.empty(Zin.shape), be.empty(Zin.shape)
    assert lib.ffno_cdft_rows2(p(hS), p(Zv), B, M, C, Kx, Ky, 0, None) == 0
    assert lib.ffno_cdft_rows_mfma(p(hS), p(Zm), p(ws), p(tw), B, M, C, Kx, Ky, 0, None) == 0
    assert rel_l2(be.get(Zm), be.get(Zv)) < 2e-6
    Sv, Sm = be.empty(S.shape), be.empty(S.shape)
    assert lib.ffno_cdft_rows2(p(hZ), p(Sv), B, M, C, Kx, Ky, 1, None) == 0
    assert lib.ffno_cdft_rows_mfma(p(hZ), p(Sm), p(ws), p(tw), B, M, C, Kx, Ky, 1, None) == 0
    assert rel_l2(be.get(Sm), be.get(Sv)) < 2e-6
    lhs = (be.get(Zm).astype(np.float64) * Zin).sum()
    rhs = (S.astype(np.float64) * be.get(Sm)).sum()
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    assert lib.ffno_cdft_rows_mfma(p(hS), p(Zm), None, p(tw), B, M, C, Kx, Ky, 0, None) == -1
    assert lib.ffno_cdft_rows_mfma(p(hS), p(Zm), p(ws), p(tw), B, 2 * Kx - 1, C, Kx, Ky, 0, None) == -3
