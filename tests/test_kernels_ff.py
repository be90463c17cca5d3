"""Feed-forward / pointwise kernels through the C ABI vs fp64 numpy references -- on the CPU wave
emulator (-m "not gpu") and on the MI355X (-m gpu)."""
import ctypes

import numpy as np
import pytest

from backend_util import be, rel_l2  # noqa: F401
from fourierflow_amd._capi import TrDesc, WnDesc

TOL = 1e-5


def ff_ref(s, resid, W1, b1, W2, b2):
    s64 = s.astype(np.float64)
    h = np.maximum(s64 @ W1.astype(np.float64).T + b1, 0)
    out = h @ W2.astype(np.float64).T + b2
    if resid is not None:
        out = out + resid
    return out, h


@pytest.mark.parametrize("P,C,H", [(70, 64, 256), (33, 32, 128), (64, 64, 128), (40, 32, 64), (3000, 64, 256)])
def test_ff_fwd_bwd(be, P, C, H):
    if be.kind == "emu" and P > 1000:
        pytest.skip("large case runs on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P + C + H)
    s = rs.standard_normal((P, C)).astype(np.float32)
    resid = rs.standard_normal((P, C)).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = (rs.standard_normal(H) * 0.1).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = (rs.standard_normal(C) * 0.1).astype(np.float32)
    ds_, dres, dW1_, db1_, dW2_, db2_ = map(be.put, (s, resid, W1, b1, W2, b2))
    out, h = be.empty((P, C)), be.empty((P, H))
    mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    assert lib.ffno_ff_fwd(p(ds_), p(dres), p(dW1_), p(db1_), p(dW2_), p(db2_), p(out), p(h), p(mask), P, C, H, None) == 0
    ref_out, ref_h = ff_ref(s, resid, W1, b1, W2, b2)
    assert rel_l2(be.get(out), ref_out) < TOL
    assert rel_l2(be.get(h), ref_h) < TOL
    # no residual, no h/mask outputs
    out2 = be.empty((P, C))
    assert lib.ffno_ff_fwd(p(ds_), None, p(dW1_), p(db1_), p(dW2_), p(db2_), p(out2), None, None, P, C, H, None) == 0
    assert rel_l2(be.get(out2), ref_out - resid) < TOL
    # in place: out aliases resid
    assert lib.ffno_ff_fwd(p(ds_), p(dres), p(dW1_), p(db1_), p(dW2_), p(db2_), p(dres), None, None, P, C, H, None) == 0
    assert rel_l2(be.get(dres), ref_out) < TOL

    # backward (data)
    db = rs.standard_normal((P, C)).astype(np.float32)
    ddb, dh, ds = be.put(db), be.empty((P, H)), be.empty((P, C))
    dW1t, dW2t = be.put(W1.T.copy()), be.put(W2.T.copy())   # the backward chain takes the transposed weights
    assert lib.ffno_ff_bwd_data(p(ddb), p(mask), p(dW1t), p(dW2t), p(dh), p(ds), P, C, H, None) == 0
    ref_dh = (db.astype(np.float64) @ W2.astype(np.float64)) * (ref_h > 0)
    ref_ds = ref_dh @ W1.astype(np.float64)
    assert rel_l2(be.get(dh), ref_dh) < TOL
    assert rel_l2(be.get(ds), ref_ds) < TOL

    # backward (weights), deterministic two-step reduction
    nsplit = 3 if P < 1000 else 64
    partial = be.zeros(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit))
    assert lib.ffno_ff_bwd_weights_partial(p(ds_), p(ddb), p(h), p(dh), p(partial), P, C, H, nsplit, None) == 0
    gW1, gW2, gb1, gb2 = be.zeros((H, C)), be.zeros((C, H)), be.zeros(H), be.zeros(C)
    assert lib.ffno_ff_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, nsplit, 0, None) == 0
    assert rel_l2(be.get(gW1), ref_dh.T @ s.astype(np.float64)) < TOL
    assert rel_l2(be.get(gW2), db.astype(np.float64).T @ ref_h) < TOL
    assert rel_l2(be.get(gb1), ref_dh.sum(0)) < TOL
    assert rel_l2(be.get(gb2), db.astype(np.float64).sum(0)) < TOL
    assert lib.ffno_ff_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, nsplit, 1, None) == 0
    assert rel_l2(be.get(gW1), 2 * ref_dh.T @ s.astype(np.float64)) < TOL


def test_ff_rejects_unsupported_shapes(be):
    z = be.zeros(64)
    p = be.ptr
    assert be.lib.ffno_ff_fwd(p(z), None, p(z), p(z), p(z), p(z), p(z), None, None, 1, 48, 192, None) == -2
    assert be.lib.ffno_ff_fwd(None, None, p(z), p(z), p(z), p(z), p(z), None, None, 1, 64, 256, None) == -1


def test_weightnorm_batched(be):
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(3)
    shapes = [(64, 3), (256, 64), (64, 256), (1, 128), (5, 70)]
    host, dev, descs = [], [], (WnDesc * len(shapes))()
    for i, (r, c) in enumerate(shapes):
        g = rs.uniform(0.5, 1.5, r).astype(np.float32)
        v = rs.standard_normal((r, c)).astype(np.float32)
        dw = rs.standard_normal((r, c)).astype(np.float32)
        d = [be.put(g), be.put(v), be.empty((r, c)), be.put(dw), be.empty(r), be.empty((r, c))]
        host.append((g, v, dw))
        dev.append(d)
        descs[i] = WnDesc(*[p(t) for t in d], r, c)
    table = be.put(np.frombuffer(bytes(descs), dtype=np.uint8))
    mr = max(r for r, _ in shapes)
    assert lib.ffno_weightnorm_fwd(p(table), len(shapes), mr, None) == 0
    assert lib.ffno_weightnorm_bwd(p(table), len(shapes), mr, None) == 0
    for (g, v, dw), d in zip(host, dev):
        v64, dw64 = v.astype(np.float64), dw.astype(np.float64)
        nrm = np.linalg.norm(v64, axis=1, keepdims=True)
        assert rel_l2(be.get(d[2]), g[:, None] * v64 / nrm) < TOL
        rdg = (dw64 * v64 / nrm).sum(1)
        assert rel_l2(be.get(d[4]), rdg) < TOL
        assert rel_l2(be.get(d[5]), g[:, None] / nrm * (dw64 - rdg[:, None] * v64 / nrm)) < TOL


def test_transpose_batched(be):
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(4)
    shapes = [(256, 64), (64, 256), (33, 70), (1, 5)]
    src = [rs.standard_normal(sh).astype(np.float32) for sh in shapes]
    dsrc = [be.put(a) for a in src]
    ddst = [be.empty((sh[1], sh[0])) for sh in shapes]
    descs = (TrDesc * len(shapes))(*[TrDesc(p(a), p(b), sh[0], sh[1]) for a, b, sh in zip(dsrc, ddst, shapes)])
    table = be.put(np.frombuffer(bytes(descs), dtype=np.uint8))
    assert lib.ffno_transpose_batched(p(table), len(shapes), 256, 256, None) == 0
    for a, b in zip(src, ddst):
        np.testing.assert_array_equal(be.get(b), a.T)


def padmap(be, size, padded):
    from fourierflow_amd._capi import PadMap
    pm = PadMap((ctypes.c_int32 * 3)(*size), (ctypes.c_int32 * 3)(*padded))
    return pm


def pad_index(Bn, size, padded):
    """unpadded pixel p -> padded pixel q for [B][s0][s1][s2] inside [B][p0][p1][p2] (pad at the END of each axis)."""
    b, x, y, z = np.meshgrid(np.arange(Bn), np.arange(size[0]), np.arange(size[1]), np.arange(size[2]), indexing="ij")
    return (((b * padded[0] + x) * padded[1] + y) * padded[2] + z).reshape(-1)


@pytest.mark.parametrize("P,Cin,C", [(100, 3, 64), (77, 5, 32), (50, 37, 64)])
def test_lift(be, P, Cin, C):
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P + Cin)
    x = rs.standard_normal((P, Cin)).astype(np.float32)
    W = rs.standard_normal((C, Cin)).astype(np.float32)
    b = rs.standard_normal(C).astype(np.float32)
    dx, dW_, db_, out = be.put(x), be.put(W), be.put(b), be.empty((P, C))
    assert lib.ffno_lift_fwd(p(dx), p(dW_), p(db_), p(out), P, Cin, C, None, None, None) == 0
    assert rel_l2(be.get(out), x.astype(np.float64) @ W.T + b) < TOL
    g = rs.standard_normal((P, C)).astype(np.float32)
    nsplit = 3
    dg, partial, gW, gb = be.put(g), be.zeros(nsplit * C * (Cin + 1)), be.zeros((C, Cin)), be.zeros(C)
    assert lib.ffno_lift_bwd(p(dx), p(dg), p(partial), p(gW), p(gb), P, Cin, C, nsplit, 0, None, None) == 0
    assert rel_l2(be.get(gW), g.astype(np.float64).T @ x) < TOL
    assert rel_l2(be.get(gb), g.astype(np.float64).sum(0)) < TOL


def test_lift_and_head_through_pad_map(be):
    """mesh_3d.py:165,173: the lift writes into / the head reads from the zero-padded activation buffer."""
    lib, p = be.lib, be.ptr
    Bn, size, padded, Cin, C, O, D = 2, (3, 4, 5), (5, 7, 6), 4, 64, 4, 128
    P, Pp = Bn * int(np.prod(size)), Bn * int(np.prod(padded))
    q = pad_index(Bn, size, padded)
    pm = padmap(be, size, padded)
    rs = np.random.RandomState(0)
    x = rs.standard_normal((P, Cin)).astype(np.float32)
    W = rs.standard_normal((C, Cin)).astype(np.float32)
    b = rs.standard_normal(C).astype(np.float32)
    dx, dW_, db_, out = be.put(x), be.put(W), be.put(b), be.zeros((Pp, C))
    assert lib.ffno_lift_fwd(p(dx), p(dW_), p(db_), p(out), P, Cin, C, ctypes.byref(pm), None, None) == 0
    ref = np.zeros((Pp, C))
    ref[q] = x.astype(np.float64) @ W.T + b
    assert rel_l2(be.get(out), ref) < TOL            # pad region untouched (stays zero)
    g = rs.standard_normal((Pp, C)).astype(np.float32)
    dg, partial, gW, gb = be.put(g), be.zeros(2 * C * (Cin + 1)), be.zeros((C, Cin)), be.zeros(C)
    assert lib.ffno_lift_bwd(p(dx), p(dg), p(partial), p(gW), p(gb), P, Cin, C, 2, 0, ctypes.byref(pm), None) == 0
    assert rel_l2(be.get(gW), g[q].astype(np.float64).T @ x) < TOL
    # head with O outputs reading the padded buffer
    Wa = (rs.standard_normal((D, C)) / 8).astype(np.float32)
    ca = rs.standard_normal(D).astype(np.float32)
    Wb = (rs.standard_normal((O, D)) / 8).astype(np.float32)
    cb = rs.standard_normal(O).astype(np.float32)
    feat = rs.standard_normal((Pp, C)).astype(np.float32)
    dWa, dca_, dWb, dcb_, dft = map(be.put, (Wa, ca, Wb, cb, feat))
    fold, y = be.zeros(O * (C + 1)), be.empty((P, O))
    assert lib.ffno_head_fold(p(dWa), p(dca_), p(dWb), p(dcb_), p(fold), C, D, O, None) == 0
    assert lib.ffno_head_fwd(p(dft), p(fold), p(y), P, C, O, 0, ctypes.byref(pm), None) == 0
    refy = (feat[q].astype(np.float64) @ Wa.T + ca) @ Wb.T + cb
    assert rel_l2(be.get(y), refy) < TOL
    gy = rs.standard_normal((P, O)).astype(np.float32)
    nsplit = 3
    dgy, gbuf, part, red = be.put(gy), be.zeros((Pp, C)), be.zeros(nsplit * O * (C + 1)), be.zeros(O * (C + 1))
    assert lib.ffno_head_bwd(p(dft), p(dgy), p(fold), p(gbuf), p(part), p(red), P, C, O, nsplit, ctypes.byref(pm), None, None) == 0
    weff = Wb.astype(np.float64) @ Wa
    refg = np.zeros((Pp, C))
    refg[q] = gy.astype(np.float64) @ weff
    assert rel_l2(be.get(gbuf), refg) < TOL
    gWa, gca, gWb, gcb = be.zeros(Wa.shape), be.zeros(ca.shape), be.zeros(Wb.shape), be.zeros(cb.shape)
    assert lib.ffno_head_param_grads(p(red), p(dWa), p(dca_), p(dWb), p(gWa), p(gca), p(gWb), p(gcb), C, D, O, 0, None) == 0
    hid = feat[q].astype(np.float64) @ Wa.T + ca
    assert rel_l2(be.get(gWb), gy.astype(np.float64).T @ hid) < 1e-4
    assert rel_l2(be.get(gWa), (gy.astype(np.float64) @ Wb).T @ feat[q]) < 1e-4
    assert rel_l2(be.get(gca), (gy.astype(np.float64) @ Wb).sum(0)) < 1e-4
    assert rel_l2(be.get(gcb), gy.astype(np.float64).sum(0)) < 1e-4


@pytest.mark.parametrize("P,C", [(300, 64), (77, 32), (1100, 64), (2100, 32)])      # (the last two: head_bwd on 1024-thread workgroups)
def test_head(be, P, C):
    lib, p = be.lib, be.ptr
    D = 128
    rs = np.random.RandomState(P)
    Wa = (rs.standard_normal((D, C)) / 8).astype(np.float32)
    ca = rs.standard_normal(D).astype(np.float32)
    Wb = (rs.standard_normal((1, D)) / 8).astype(np.float32)
    cb = rs.standard_normal(1).astype(np.float32)
    bfeat = rs.standard_normal((P, C)).astype(np.float32)
    dWa, dca_, dWb, dcb_, dbf = map(be.put, (Wa, ca, Wb, cb, bfeat))
    fold, y = be.zeros(C + 1), be.empty(P)
    assert lib.ffno_head_fold(p(dWa), p(dca_), p(dWb), p(dcb_), p(fold), C, D, 1, None) == 0
    assert lib.ffno_head_fwd(p(dbf), p(fold), p(y), P, C, 1, 0, None, None) == 0
    ref = ((bfeat.astype(np.float64) @ Wa.T + ca) @ Wb.T + cb)[:, 0]
    assert rel_l2(be.get(y), ref) < TOL
    assert lib.ffno_head_fwd(p(dbf), p(fold), p(y), P, C, 1, 1, None, None) == 0
    assert rel_l2(be.get(y), 2 * ref) < TOL
    gy = rs.standard_normal(P).astype(np.float32)
    nsplit = 4
    dgy, gb, partial, red = be.put(gy), be.empty((P, C)), be.zeros(nsplit * (C + 1)), be.zeros(C + 1)
    assert lib.ffno_head_bwd(p(dbf), p(dgy), p(fold), p(gb), p(partial), p(red), P, C, 1, nsplit, None, None, None) == 0
    weff = (Wb.astype(np.float64) @ Wa)[0]
    assert rel_l2(be.get(gb), gy[:, None] * weff[None]) < TOL
    G = gy.astype(np.float64) @ bfeat
    S = gy.astype(np.float64).sum()
    r = be.get(red)
    assert rel_l2(r[:C], G) < TOL and abs(r[C] - S) < 1e-4
    gWa, gca, gWb, gcb = be.zeros(Wa.shape), be.zeros(ca.shape), be.zeros(Wb.shape), be.zeros(cb.shape)
    assert lib.ffno_head_param_grads(p(red), p(dWa), p(dca_), p(dWb), p(gWa), p(gca), p(gWb), p(gcb), C, D, 1, 0, None) == 0
    hid = bfeat.astype(np.float64) @ Wa.T + ca
    assert rel_l2(be.get(gWb)[0], gy.astype(np.float64) @ hid) < 1e-4
    assert rel_l2(be.get(gWa), np.outer(Wb[0], G)) < TOL
    assert rel_l2(be.get(gca), Wb[0] * S) < 1e-4
    assert abs(be.get(gcb)[0] - S) < 1e-4


@pytest.mark.parametrize("B,n", [(3, 500), (2, 20000)])     # one slice per sample / several (two-stage reduction)
def test_lploss(be, B, n):
    import torch
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(9)
    pred = rs.standard_normal((B, n)).astype(np.float32)
    tgt = rs.standard_normal((B, n)).astype(np.float32)
    ntmp = lib.ffno_lploss_tmp_floats(B, n)
    assert ntmp == 2 * B * max(1, min(64, (n + 1023) // 1024))
    dpred, dtgt, loss, gp, tmp = be.put(pred), be.put(tgt), be.zeros(1), be.zeros((B, n)), be.zeros(ntmp)
    assert lib.ffno_lploss_fwd_bwd(p(dpred), p(dtgt), p(loss), p(gp), p(tmp), B, n, 1.0, None, None) == 0
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    tt = torch.tensor(tgt, dtype=torch.float64)
    l = ((pt - tt).norm(dim=1) / tt.norm(dim=1)).mean()   # LpLoss.rel, loss.py:33-46
    l.backward()
    assert abs(be.get(loss)[0] - l.item()) < 1e-6
    assert rel_l2(be.get(gp), pt.grad.numpy()) < TOL


def test_adamw_and_axpy(be):
    import torch
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(9)
    # AdamW vs torch.optim.AdamW
    n = 1000
    p0 = rs.standard_normal(n).astype(np.float32)
    dp, m, v = be.put(p0), be.zeros(n), be.zeros(n)
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.AdamW([tp], lr=2.5e-3, weight_decay=1e-4)
    for step in range(1, 4):
        g = rs.standard_normal(n).astype(np.float32)
        dg = be.put(g)
        assert lib.ffno_adamw_flat(p(dp), p(dg), p(m), p(v), n, 2.5e-3, 0.9, 0.999, 1e-8, 1e-4, step, 1.0, None) == 0
        tp.grad = torch.tensor(g)
        opt.step()
    assert rel_l2(be.get(dp), tp.detach().numpy()) < 1e-6
    # Adam with L2 weight decay vs torch.optim.Adam (the geo-FNO baselines' optimiser)
    dp, m, v = be.put(p0), be.zeros(n), be.zeros(n)
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.Adam([tp], lr=1e-3, weight_decay=1e-4)
    for step in range(1, 4):
        g = rs.standard_normal(n).astype(np.float32)
        dg = be.put(g)
        assert lib.ffno_adam_flat(p(dp), p(dg), p(m), p(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step, 1.0, None) == 0
        tp.grad = torch.tensor(g)
        opt.step()
    assert rel_l2(be.get(dp), tp.detach().numpy()) < 1e-6
    yv = rs.standard_normal(100).astype(np.float32)
    xv = rs.standard_normal(100).astype(np.float32)
    dy, dxv = be.put(yv), be.put(xv)
    assert lib.ffno_axpy(p(dy), p(dxv), 0.5, 100, None) == 0
    np.testing.assert_allclose(be.get(dy), yv + 0.5 * xv, rtol=1e-6)
