"""Layer-level C entry points (include/ffno.h: ffno_layer_fwd / ffno_layer_bwd, SURVEY 8b "signature level 2") against the
kernel-level calls they sequence: bit-identical outputs, and the engine really drives a block through them."""
import ctypes

import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import be, host_device, rel_l2  # noqa: F401
from test_kernels_ffx import pack_weights
from test_kernels_spectral import _x3_pack


@pytest.mark.parametrize("kernel", ["x3", "fused"])
def test_layer_fwd_bwd_equal_the_kernel_level_sequence(be, kernel):
    from fourierflow_amd._capi import FusedBranch, LayerBwdDesc, LayerFwdDesc
    lib, p = be.lib, be.ptr
    B, M, N, K, C, H = 2, 10, 12, 5, 64, 256
    P = B * M * N
    rs = np.random.RandomState(3)
    x, g, g2 = (rs.standard_normal((B, M, N, C)).astype(np.float32) for _ in range(3))
    W1 = (rs.standard_normal((H, C)) / 8).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / 16).astype(np.float32)
    b1, b2 = (rs.standard_normal(H) * 0.1).astype(np.float32), (rs.standard_normal(C) * 0.1).astype(np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights(be, W1, W2)
    db1, db2 = be.put(b1), be.put(b2)
    planes, keep = [], []
    for axis in (0, 1):
        w = (rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32)
        pk_f, pk_a, kp = _x3_pack(be, w, K)
        keep.append(kp)
        planes.append((pk_f, pk_a) if kernel == "x3" else (kp[0], kp[1]))
    tw = [be.twiddle(N), be.twiddle(M)]
    R = [B * M, B * N]
    nsplit = 3
    res = {}
    for mode in ("kernels", "layer"):
        dx = be.put(x)
        s, t, out = be.empty(x.shape), be.empty(x.shape), be.empty(x.shape)
        sx = [be.empty((K, R[i], 2, C)) for i in range(2)]
        mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
        br = [FusedBranch(p(dx), p(s if i == 0 else t), None, p(sx[i]), p(planes[i][0]), p(tw[i]), B, M, N, K, i, 0) for i in range(2)]
        if mode == "kernels":
            fn = lib.ffno_spectral_x3_pair if kernel == "x3" else lib.ffno_spectral_fused_pair
            extra = (1,) if kernel == "x3" else ()
            assert fn(ctypes.byref(br[0]), ctypes.byref(br[1]), C, 0, 1, 0, *extra, None) == 0
            assert lib.ffno_ffx_fwd2(p(s), p(t), p(s), p(dx), p(a1), p(db1), p(a2), p(db2), p(out), p(mask), P, C, H, None, None) == 0
        else:
            d = LayerFwdDesc(br[0], br[1], int(kernel == "x3"), 1, p(a1), p(db1), p(a2), p(db2), p(s), p(dx), p(out), p(mask), P, C, H, 0)
            assert lib.ffno_layer_fwd(ctypes.byref(d), None) == 0
        # backward of the same layer
        dg, dg2, ds, gout, g1 = be.put(g), be.put(g2), be.empty(x.shape), be.empty(x.shape), be.empty(x.shape)
        sd = [be.empty((K, R[i], 2, C)) for i in range(2)]
        part = be.zeros(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit))
        ab = [FusedBranch(p(ds), p(gout if i == 0 else g1), p(dg) if i == 0 else None, p(sd[i]), p(planes[i][1]), p(tw[i]), B, M, N, K,
                          i, 0) for i in range(2)]
        if mode == "kernels":
            assert lib.ffno_ffx_bwd_data2(p(dg), p(dg2), p(dg), p(mask), p(a1b), p(a2b), p(ds), P, C, H, None, None) == 0
            assert lib.ffno_ffx_bwd_weights_partial(p(s), p(dg), p(a1), p(db1), p(a1b), p(part), P, C, H, nsplit, None) == 0
            fn = lib.ffno_spectral_x3_pair if kernel == "x3" else lib.ffno_spectral_fused_pair
            extra = (1,) if kernel == "x3" else ()
            assert fn(ctypes.byref(ab[0]), ctypes.byref(ab[1]), C, 1, 0, 1, *extra, None) == 0
        else:
            d = LayerBwdDesc(ab[0], ab[1], int(kernel == "x3"), 1, p(dg), p(dg2), p(dg), p(mask), p(a1b), p(a2b), p(ds), p(s), p(a1),
                             p(db1), p(part), nsplit, P, C, H)
            assert lib.ffno_layer_bwd(ctypes.byref(d), None) == 0
        res[mode] = [np.array(be.get(t_)).copy() for t_ in (s, t, out, mask, sx[0], sx[1], dg, ds, gout, g1, sd[0], sd[1], part)]
    for a, b in zip(res["kernels"], res["layer"]):
        np.testing.assert_array_equal(a, b)
    assert lib.ffno_layer_fwd(None, None) == -1 and lib.ffno_layer_bwd(None, None) == -1


def test_engine_drives_the_block_through_layer_calls(host_device):
    """With no per-kernel timer attached the engine issues ONE C call per layer and direction; results equal the
    kernel-level sequencing bit for bit."""
    from fourierflow_amd.modules import FNOFactorized2DBlock
    kw = dict(modes=4, width=64, input_dim=3, n_layers=2, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)
    x_np, t_np = gu.make_block_io(kw, 5, 1, 8, 8)
    outs = {}
    for layer_calls in (True, False):
        blk = FNOFactorized2DBlock(**kw)
        blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_block_state_dict(kw, 5).items()})
        blk = blk.to(host_device)
        eng = blk.engine()
        eng.use_layer_calls = layer_calls
        seen = []
        orig = eng._k
        eng._k = lambda name, fn, *a, _o=orig: (seen.append(name), _o(name, fn, *a))[1]
        pred = blk(torch.from_numpy(x_np).to(host_device))["forecast"]
        ((pred - torch.from_numpy(t_np).to(host_device)) ** 2).mean().backward()
        assert ("layer_fwd" in seen and "layer_bwd" in seen) == layer_calls
        outs[layer_calls] = [pred.detach().cpu().numpy()] + [p.grad.cpu().numpy() for _, p in blk.engine_parameters()]
    for a, b in zip(outs[True], outs[False]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("P,C", [(37, 64), (300, 32), (5000, 64)])
def test_layernorm_kernels_vs_torch(be, P, C):
    """ffno_layernorm_fwd / _bwd (FeedForward(layer_norm=True), feedforward.py:18-19) against torch.nn.functional.layer_norm
    in fp64: forward with the fused residual, backward with the fused sum of two gradient buffers, accumulate flag."""
    if be.kind == "emu" and P > 1000:
        pytest.skip("large case runs on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P + C)
    t, resid, g, g2 = (rs.standard_normal((P, C)).astype(np.float32) for _ in range(4))
    gamma, beta = rs.uniform(0.5, 1.5, C).astype(np.float32), rs.uniform(-0.2, 0.2, C).astype(np.float32)
    dt_, dga, dbe = be.put(t), be.put(gamma), be.put(beta)
    out, stats = be.empty((P, C)), be.empty((P, 2))
    assert lib.ffno_layernorm_fwd(p(dt_), p(dga), p(dbe), p(be.put(resid)), p(out), p(stats), P, C, 1e-5, None) == 0
    tt = torch.tensor(t, dtype=torch.float64, requires_grad=True)
    gt, bt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True), torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    ref = torch.nn.functional.layer_norm(tt, (C,), gt, bt, 1e-5)
    assert rel_l2(be.get(out), ref.detach().numpy() + resid) < 1e-6
    ref.backward(torch.tensor((g + g2).astype(np.float64)))
    gsum, dx = be.empty((P, C)), be.empty((P, C))
    part = be.zeros(2 * C * lib.ffno_layernorm_nsplit(P))
    dgm, dbt = be.zeros(C), be.zeros(C)
    for acc in (0, 1):
        assert lib.ffno_layernorm_bwd(p(dt_), p(stats), p(dga), p(be.put(g)), p(be.put(g2)), p(gsum), p(dx), p(part), p(dgm), p(dbt),
                                      P, C, acc, None) == 0
    np.testing.assert_array_equal(be.get(gsum), g + g2)
    assert rel_l2(be.get(dx), tt.grad.numpy()) < 1e-5
    assert rel_l2(be.get(dgm), 2 * gt.grad.numpy()) < 1e-5 and rel_l2(be.get(dbt), 2 * bt.grad.numpy()) < 1e-5
    assert lib.ffno_layernorm_fwd(p(dt_), p(dga), p(dbe), None, p(out), p(stats), P, 48, 1e-5, None) == -2
