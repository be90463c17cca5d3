"""Pointwise linear kernels of the FNOZongyi2DBlock baseline (csrc/plin.hip; reference zongyi_fno/grid_2d.py:22,45,74-77,
:106, :119-122) against numpy float64 restatements of nn.Linear / ReLU and their gradients.  Tolerance: 2e-6 rel-L2
(fp32 dot products in a different summation order than torch)."""
import ctypes

import numpy as np
import pytest

from backend_util import be, rel_l2  # noqa: F401
from fourierflow_amd import _capi

TOL = 2e-6

# (P, Cin, ldx, Cout, ldo): in_proj, the per-layer linear, the two head linears, a ragged tail
SHAPES = [(2 * 64 * 64, 12, 12, 32, 32), (300, 20, 32, 20, 32), (1000, 32, 32, 128, 128), (777, 128, 128, 1, 1),
          (65, 5, 8, 7, 16), (150, 64, 64, 128, 128), (100, 64, 64, 64, 64)]


def _data(P, Cin, ldx, Cout, ldo, seed=0):
    rs = np.random.RandomState(seed)
    x = np.zeros((P, ldx), np.float32)
    x[:, :Cin] = rs.standard_normal((P, Cin))
    W = (rs.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    add = np.zeros((P, ldo), np.float32)
    add[:, :Cout] = rs.standard_normal((P, Cout))
    return x, W, b, add


@pytest.mark.parametrize("P,Cin,ldx,Cout,ldo", SHAPES)
def _gelu(v):
    from scipy.special import erf
    return 0.5 * v * (1 + erf(v / np.sqrt(2)))


def _dgelu(v):
    from scipy.special import erf
    return 0.5 * (1 + erf(v / np.sqrt(2))) + v * np.exp(-0.5 * v * v) / np.sqrt(2 * np.pi)


@pytest.mark.parametrize("P,Cin,ldx,Cout,ldo", SHAPES)
@pytest.mark.parametrize("act_mode,with_add", [(1, True), (0, False), (2, True)])
def test_plin_forward(be, P, Cin, ldx, Cout, ldo, act_mode, with_add):
    x, W, b, add = _data(P, Cin, ldx, Cout, ldo)
    out, out2, pre = be.empty((P, ldo)), be.empty((P, ldo)), be.empty((P, ldo))
    hx, hW, hb, hadd = be.put(x), be.put(W), be.put(b), be.put(add)      # handles stay alive across the launch
    rc = be.lib.ffno_plin_fwd(be.ptr(hx), ldx, be.ptr(hW), be.ptr(hb), be.ptr(hadd) if with_add else None, be.ptr(out), ldo,
                              be.ptr(hadd), be.ptr(out2), be.ptr(pre), P, Cin, Cout, act_mode, None)
    assert rc == 0
    ref_pre = x[:, :Cin].astype(np.float64) @ W.astype(np.float64).T + b
    if with_add:
        ref_pre = ref_pre + add[:, :Cout]
    ref = np.maximum(ref_pre, 0) if act_mode == 1 else (_gelu(ref_pre) if act_mode == 2 else ref_pre)
    got = be.get(out)
    assert rel_l2(got[:, :Cout], ref) < TOL
    assert rel_l2(be.get(pre)[:, :Cout], ref_pre) < TOL        # pre-activation copy (what the GELU backward reads)
    assert not got[:, Cout:].any() and not be.get(pre)[:, Cout:].any()          # the pad channels are exact zeros
    np.testing.assert_array_equal(be.get(out2), got + add)      # second output: out + res


@pytest.mark.parametrize("P,Cin,ldx,Cout,ldo", SHAPES)
@pytest.mark.parametrize("masked", [True, False, "gelu"])
def test_plin_backward(be, P, Cin, ldx, Cout, ldo, masked):
    x, W, b, _ = _data(P, Cin, ldx, Cout, ldo, seed=1)
    rs = np.random.RandomState(2)
    g = np.zeros((P, ldo), np.float32)
    g[:, :Cout] = rs.standard_normal((P, Cout))
    act = np.zeros((P, ldo), np.float32)
    act[:, :Cout] = np.maximum(rs.standard_normal((P, Cout)), 0)
    mode = 0
    if masked == "gelu":          # `act` = the kept PRE-activation, dpre = g * gelu'(pre)
        act[:, :Cout] = rs.standard_normal((P, Cout)) * 1.5
        dpre, mode = g[:, :Cout].astype(np.float64) * _dgelu(act[:, :Cout].astype(np.float64)), 2
    else:
        dpre, mode = g[:, :Cout].astype(np.float64) * ((act[:, :Cout] > 0) if masked else 1.0), int(bool(masked))

    hg, hact, hW, hx = be.put(g), be.put(act), be.put(W), be.put(x)
    base = rs.standard_normal((P, ldx)).astype(np.float32)
    for accumulate in (0, 1):
        dx = be.put(base)
        dp = be.empty((P, ldo))
        rc = be.lib.ffno_plin_bwd_data(be.ptr(hg), ldo, be.ptr(hact) if masked else None, be.ptr(hW), be.ptr(dx), ldx,
                                       be.ptr(dp), P, Cin, Cout, accumulate, mode, None)
        assert rc == 0
        ref = np.zeros((P, ldx))
        ref[:, :Cin] = dpre @ W.astype(np.float64)
        if accumulate:
            ref = ref + base
        assert rel_l2(be.get(dx), ref) < TOL
        got_dp = be.get(dp)
        if masked == "gelu":
            assert rel_l2(got_dp[:, :Cout], dpre) < TOL
        else:
            np.testing.assert_array_equal(got_dp[:, :Cout], dpre.astype(np.float32))
        assert not got_dp[:, Cout:].any()

    part = be.empty(int(be.lib.ffno_plin_wgrad_partial_floats(P, Cin, Cout)))
    dW0 = rs.standard_normal((Cout, Cin)).astype(np.float32)
    db0 = rs.standard_normal(Cout).astype(np.float32)
    for accumulate in (0, 1):
        dW, db = be.put(dW0), be.put(db0)
        rc = be.lib.ffno_plin_bwd_weights(be.ptr(hg), ldo, be.ptr(hact) if masked else None, be.ptr(hx), ldx, be.ptr(part),
                                          be.ptr(dW), be.ptr(db), P, Cin, Cout, accumulate, mode, None)
        assert rc == 0
        rW = dpre.T @ x[:, :Cin].astype(np.float64) + (dW0 if accumulate else 0)
        rb = dpre.sum(0) + (db0 if accumulate else 0)
        assert rel_l2(be.get(dW), rW) < TOL
        assert rel_l2(be.get(db), rb) < TOL


def test_pad_copy_round_trip(be):
    rs = np.random.RandomState(3)
    shapes = [(20, 20, 288, 32, 32), (1, 20, 1, 1, 32), (128, 20, 1, 128, 32)]     # (R, Cc, inner, Rp, Cp)
    plains = [rs.standard_normal((R, Cc, inner)).astype(np.float32) for R, Cc, inner, _, _ in shapes]
    hp = [be.put(a) for a in plains]
    hq = [be.zeros((Rp, Cp, inner)) for _, _, inner, Rp, Cp in shapes]
    arr = (_capi.PadDesc * len(shapes))(*[_capi.PadDesc(be.ptr(a), be.ptr(q), R, Cc, inner, Cp)
                                          for a, q, (R, Cc, inner, _, Cp) in zip(hp, hq, shapes)])
    table = be.put(np.frombuffer(bytes(arr), dtype=np.uint8))
    assert be.lib.ffno_pad_copy(be.ptr(table), len(shapes), 1, None) == 0
    for a, q, (R, Cc, inner, Rp, Cp) in zip(plains, hq, shapes):
        full = be.get(q)
        np.testing.assert_array_equal(full[:R, :Cc], a)
        assert not full[R:].any() and not full[:, Cc:].any()
    back = [be.zeros(a.shape) for a in plains]
    arr2 = (_capi.PadDesc * len(shapes))(*[_capi.PadDesc(be.ptr(a), be.ptr(q), R, Cc, inner, Cp)
                                           for a, q, (R, Cc, inner, _, Cp) in zip(back, hq, shapes)])
    table2 = be.put(np.frombuffer(bytes(arr2), dtype=np.uint8))
    assert be.lib.ffno_pad_copy(be.ptr(table2), len(shapes), 0, None) == 0
    for a, h in zip(plains, back):
        np.testing.assert_array_equal(be.get(h), a)


def test_plin_rejects_bad_arguments(be):
    x = be.zeros((4, 8))
    assert be.lib.ffno_plin_fwd(None, 8, be.ptr(x), None, None, be.ptr(x), 8, None, None, None, 4, 8, 8, 0, None) == -1
    assert be.lib.ffno_plin_fwd(be.ptr(x), 4, be.ptr(x), None, None, be.ptr(x), 8, None, None, None, 4, 8, 8, 0, None) == -1     # ldx < Cin
    assert be.lib.ffno_plin_fwd(be.ptr(x), 200, be.ptr(x), None, None, be.ptr(x), 8, None, None, None, 4, 200, 8, 0, None) == -2
    assert be.lib.ffno_plin_supported(128, 128) == 0 and be.lib.ffno_plin_supported(64, 128) == 1
    assert be.lib.ffno_plin_fwd(be.ptr(x), 8, be.ptr(x), None, None, be.ptr(x), 8, None, None, None, 4, 8, 8, 3, None) == -1   # act_mode
