"""Feed-forward / pointwise kernel sources through the CPU wave emulator vs fp64 numpy references."""
import ctypes

import numpy as np
import pytest

from emu_util import emu_lib, ptr, rel_l2
from fourierflow_amd._capi import WnDesc

pytestmark = pytest.mark.emu
TOL = 1e-5


def ff_ref(s, resid, W1, b1, W2, b2):
    s64 = s.astype(np.float64)
    hpre = s64 @ W1.astype(np.float64).T + b1
    h = np.maximum(hpre, 0)
    out = h @ W2.astype(np.float64).T + b2
    if resid is not None:
        out = out + resid
    return out, h


@pytest.mark.parametrize("P,C,H", [(70, 64, 256), (33, 32, 128), (64, 64, 128), (40, 32, 64)])
def test_ff_fwd_bwd(P, C, H):
    lib = emu_lib()
    rs = np.random.RandomState(P + C + H)
    s = rs.standard_normal((P, C)).astype(np.float32)
    resid = rs.standard_normal((P, C)).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = (rs.standard_normal(H) * 0.1).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = (rs.standard_normal(C) * 0.1).astype(np.float32)
    out = np.full((P, C), np.nan, np.float32)
    h = np.full((P, H), np.nan, np.float32)
    mask = np.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    assert lib.ffno_ff_fwd(ptr(s), ptr(resid), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(out), ptr(h), ptr(mask),
                           P, C, H, None) == 0
    ref_out, ref_h = ff_ref(s, resid, W1, b1, W2, b2)
    assert rel_l2(out, ref_out) < TOL
    assert rel_l2(h, ref_h) < TOL
    # no residual, no h/mask outputs, in-place-safe
    out2 = np.full((P, C), np.nan, np.float32)
    assert lib.ffno_ff_fwd(ptr(s), None, ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(out2), None, None, P, C, H, None) == 0
    assert rel_l2(out2, ref_out - resid) < TOL

    # backward (data)
    db = rs.standard_normal((P, C)).astype(np.float32)
    dh = np.full((P, H), np.nan, np.float32)
    ds = np.full((P, C), np.nan, np.float32)
    assert lib.ffno_ff_bwd_data(ptr(db), ptr(mask), ptr(W1), ptr(W2), ptr(dh), ptr(ds), P, C, H, None) == 0
    ref_dh = (db.astype(np.float64) @ W2.astype(np.float64)) * (ref_h > 0)
    ref_ds = ref_dh @ W1.astype(np.float64)
    assert rel_l2(dh, ref_dh) < TOL
    assert rel_l2(ds, ref_ds) < TOL

    # backward (weights)
    nsplit = 3
    partial = np.zeros(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit), np.float32)
    assert lib.ffno_ff_bwd_weights_partial(ptr(s), ptr(db), ptr(h), ptr(dh), ptr(partial), P, C, H, nsplit, None) == 0
    dW1, dW2 = np.zeros((H, C), np.float32), np.zeros((C, H), np.float32)
    db1, db2 = np.zeros(H, np.float32), np.zeros(C, np.float32)
    assert lib.ffno_ff_bwd_weights_reduce(ptr(partial), ptr(dW1), ptr(dW2), ptr(db1), ptr(db2), C, H, nsplit, 0, None) == 0
    assert rel_l2(dW1, ref_dh.T @ s.astype(np.float64)) < TOL
    assert rel_l2(dW2, db.astype(np.float64).T @ ref_h) < TOL
    assert rel_l2(db1, ref_dh.sum(0)) < TOL
    assert rel_l2(db2, db.astype(np.float64).sum(0)) < TOL
    assert lib.ffno_ff_bwd_weights_reduce(ptr(partial), ptr(dW1), ptr(dW2), ptr(db1), ptr(db2), C, H, nsplit, 1, None) == 0
    assert rel_l2(dW1, 2 * ref_dh.T @ s.astype(np.float64)) < TOL


def test_weightnorm_batched():
    lib = emu_lib()
    rs = np.random.RandomState(3)
    shapes = [(64, 3), (256, 64), (64, 256), (1, 128), (5, 70)]
    keep, descs = [], (WnDesc * len(shapes))()
    for i, (r, c) in enumerate(shapes):
        g = rs.uniform(0.5, 1.5, r).astype(np.float32)
        v = rs.standard_normal((r, c)).astype(np.float32)
        w = np.full((r, c), np.nan, np.float32)
        dw = rs.standard_normal((r, c)).astype(np.float32)
        dg = np.full(r, np.nan, np.float32)
        dv = np.full((r, c), np.nan, np.float32)
        keep.append((g, v, w, dw, dg, dv))
        descs[i] = WnDesc(ptr(g), ptr(v), ptr(w), ptr(dw), ptr(dg), ptr(dv), r, c)
    mr = max(r for r, _ in shapes)
    assert lib.ffno_weightnorm_fwd(ctypes.byref(descs), len(shapes), mr, None) == 0
    assert lib.ffno_weightnorm_bwd(ctypes.byref(descs), len(shapes), mr, None) == 0
    for g, v, w, dw, dg, dv in keep:
        v64, dw64 = v.astype(np.float64), dw.astype(np.float64)
        nrm = np.linalg.norm(v64, axis=1, keepdims=True)
        assert rel_l2(w, g[:, None] * v64 / nrm) < TOL
        rdg = (dw64 * v64 / nrm).sum(1)
        assert rel_l2(dg, rdg) < TOL
        assert rel_l2(dv, g[:, None] / nrm * (dw64 - rdg[:, None] * v64 / nrm)) < TOL


@pytest.mark.parametrize("P,Cin,C", [(100, 3, 64), (77, 5, 32), (50, 37, 64)])
def test_lift(P, Cin, C):
    lib = emu_lib()
    rs = np.random.RandomState(P + Cin)
    x = rs.standard_normal((P, Cin)).astype(np.float32)
    W = rs.standard_normal((C, Cin)).astype(np.float32)
    b = rs.standard_normal(C).astype(np.float32)
    out = np.full((P, C), np.nan, np.float32)
    assert lib.ffno_lift_fwd(ptr(x), ptr(W), ptr(b), ptr(out), P, Cin, C, None) == 0
    assert rel_l2(out, x.astype(np.float64) @ W.T + b) < TOL
    g = rs.standard_normal((P, C)).astype(np.float32)
    nsplit = 3
    partial = np.zeros(nsplit * C * (Cin + 1), np.float32)
    dW, db = np.zeros((C, Cin), np.float32), np.zeros(C, np.float32)
    assert lib.ffno_lift_bwd(ptr(x), ptr(g), ptr(partial), ptr(dW), ptr(db), P, Cin, C, nsplit, 0, None) == 0
    assert rel_l2(dW, g.astype(np.float64).T @ x) < TOL
    assert rel_l2(db, g.astype(np.float64).sum(0)) < TOL


@pytest.mark.parametrize("P,C", [(300, 64), (77, 32)])
def test_head(P, C):
    lib = emu_lib()
    D = 128
    rs = np.random.RandomState(P)
    Wa = (rs.standard_normal((D, C)) / 8).astype(np.float32)
    ca = rs.standard_normal(D).astype(np.float32)
    Wb = (rs.standard_normal((1, D)) / 8).astype(np.float32)
    cb = rs.standard_normal(1).astype(np.float32)
    bfeat = rs.standard_normal((P, C)).astype(np.float32)
    fold = np.zeros(C + 1, np.float32)
    assert lib.ffno_head_fold(ptr(Wa), ptr(ca), ptr(Wb), ptr(cb), ptr(fold), C, D, None) == 0
    y = np.full(P, np.nan, np.float32)
    assert lib.ffno_head_fwd(ptr(bfeat), ptr(fold), ptr(y), P, C, 0, None) == 0
    ref = ((bfeat.astype(np.float64) @ Wa.T + ca) @ Wb.T + cb)[:, 0]
    assert rel_l2(y, ref) < TOL
    y2 = y.copy()
    assert lib.ffno_head_fwd(ptr(bfeat), ptr(fold), ptr(y2), P, C, 1, None) == 0
    assert rel_l2(y2, 2 * ref) < TOL
    gy = rs.standard_normal(P).astype(np.float32)
    gb = np.full((P, C), np.nan, np.float32)
    nsplit = 4
    partial = np.zeros(nsplit * (C + 1), np.float32)
    red = np.zeros(C + 1, np.float32)
    assert lib.ffno_head_bwd(ptr(bfeat), ptr(gy), ptr(fold), ptr(gb), ptr(partial), ptr(red), P, C, nsplit, None) == 0
    weff = (Wb.astype(np.float64) @ Wa)[0]
    assert rel_l2(gb, gy[:, None] * weff[None]) < TOL
    G = gy.astype(np.float64) @ bfeat
    S = gy.astype(np.float64).sum()
    assert rel_l2(red[:C], G) < TOL and abs(red[C] - S) < 1e-4
    dWa, dca, dWb, dcb = np.zeros_like(Wa), np.zeros_like(ca), np.zeros_like(Wb), np.zeros_like(cb)
    assert lib.ffno_head_param_grads(ptr(red), ptr(Wa), ptr(ca), ptr(Wb), ptr(dWa), ptr(dca), ptr(dWb), ptr(dcb),
                                     C, D, 0, None) == 0
    hid = bfeat.astype(np.float64) @ Wa.T + ca
    assert rel_l2(dWb[0], gy.astype(np.float64) @ hid) < 1e-4
    assert rel_l2(dWa, np.outer(Wb[0], G)) < TOL
    assert rel_l2(dca, Wb[0] * S) < 1e-4
    assert abs(dcb[0] - S) < 1e-4


def test_lploss_and_adamw_and_axpy():
    import torch
    lib = emu_lib()
    rs = np.random.RandomState(9)
    B, n = 3, 500
    pred = rs.standard_normal((B, n)).astype(np.float32)
    tgt = rs.standard_normal((B, n)).astype(np.float32)
    loss = np.zeros(1, np.float32)
    gp = np.zeros((B, n), np.float32)
    tmp = np.zeros(2 * B, np.float32)
    assert lib.ffno_lploss_fwd_bwd(ptr(pred), ptr(tgt), ptr(loss), ptr(gp), ptr(tmp), B, n, 1.0, None) == 0
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    tt = torch.tensor(tgt, dtype=torch.float64)
    l = ((pt - tt).norm(dim=1) / tt.norm(dim=1)).mean()
    l.backward()
    assert abs(loss[0] - l.item()) < 1e-6
    assert rel_l2(gp, pt.grad.numpy()) < TOL
    # AdamW vs torch
    n = 1000
    p0 = rs.standard_normal(n).astype(np.float32)
    p = p0.copy()
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.AdamW([tp], lr=2.5e-3, weight_decay=1e-4)
    for step in range(1, 4):
        g = rs.standard_normal(n).astype(np.float32)
        assert lib.ffno_adamw_flat(ptr(p), ptr(g), ptr(m), ptr(v), n, 2.5e-3, 0.9, 0.999, 1e-8, 1e-4, step, 1.0, None) == 0
        tp.grad = torch.tensor(g)
        opt.step()
    assert rel_l2(p, tp.detach().numpy()) < 1e-6
    yv = rs.standard_normal(100).astype(np.float32)
    xv = rs.standard_normal(100).astype(np.float32)
    y0 = yv.copy()
    assert lib.ffno_axpy(ptr(yv), ptr(xv), 0.5, 100, None) == 0
    np.testing.assert_allclose(yv, y0 + 0.5 * xv, rtol=1e-6)
