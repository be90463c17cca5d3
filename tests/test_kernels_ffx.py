"""bf16x3 feed-forward kernels (fourierflow_amd/csrc/ffx.hip) through the C ABI vs fp64 numpy references -- on the CPU
wave emulator (-m "not gpu") and on the MI355X (-m gpu).  Same operator and the same tolerance as the fp32-MFMA kernels
of test_kernels_ff.py: the split arithmetic must be fp32-grade."""
import ctypes

import numpy as np
import pytest

from backend_util import be, rel_l2  # noqa: F401
from fourierflow_amd._capi import FfOpts, FxPackDesc
from test_kernels_ff import ff_ref

TOL = 1e-5


def pack_weights(be, W1, W2):
    """-> device buffers (fwd A1, fwd A2, bwd A1, bwd A2) for W1 [H][C], W2 [C][H] (include/ffno.h "packed")."""
    lib, p = be.lib, be.ptr
    H, C = W1.shape
    nbytes = lib.ffno_ffx_pack_bytes(C, H)
    assert nbytes == C * H * 6
    dW1, dW2 = be.put(W1), be.put(W2)
    bufs = [be.zeros(nbytes // 4, np.uint32) for _ in range(4)]
    spec = [(dW1, C, 1, 1), (dW2, 1, H, 2), (dW2, 1, H, 1), (dW1, C, 1, 2)]
    descs = (FxPackDesc * 4)(*[FxPackDesc(p(src), p(dst), sh, sc, ty, 0) for (src, sh, sc, ty), dst in zip(spec, bufs)])
    table = be.put(np.frombuffer(bytes(descs), dtype=np.uint8))
    assert lib.ffno_ffx_pack(p(table), 4, C, H, None) == 0
    be.get(bufs[0])   # sync
    return bufs, (dW1, dW2, table)


def test_split3_is_exact_and_pack_layout(be):
    """The three bf16 planes of a packed weight add up to the fp32 value bit for bit, at the documented positions."""
    C, H = 64, 256
    rs = np.random.RandomState(0)
    W1 = (rs.standard_normal((H, C)) * np.exp(rs.uniform(-20, 20, (H, C)))).astype(np.float32)
    W2 = rs.standard_normal((C, H)).astype(np.float32)
    bufs, _keep = pack_weights(be, W1, W2)

    def planes(buf, nfrag):
        raw = be.get(buf).view(np.uint16).reshape(nfrag, 3, 64, 8)
        f = (raw.astype(np.uint32) << 16).view(np.float32)
        return f[:, 0].astype(np.float64) + f[:, 1] + f[:, 2]            # [frag][lane][e], exact in fp64

    lane = np.arange(64)
    j, half, e = (lane & 31)[:, None], (lane >> 5)[:, None], np.arange(8)[None, :]
    a1 = planes(bufs[0], (H // 32) * (C // 16)).reshape(H // 32, C // 16, 64, 8)
    for w in (0, 3, 7):
        for st in range(C // 16):
            np.testing.assert_array_equal(a1[w, st], W1[32 * w + j, 16 * st + 8 * half + e].astype(np.float64))
    a2 = planes(bufs[1], (H // 32) * (C // 32) * 2).reshape(H // 32, C // 32, 2, 64, 8)
    for w, mt, s2 in ((0, 0, 0), (5, 1, 1), (7, 0, 1)):
        hid = 32 * w + (e & 3) + 8 * (2 * s2 + (e >> 2)) + 4 * half
        np.testing.assert_array_equal(a2[w, mt, s2], W2[32 * mt + j, hid].astype(np.float64))


@pytest.mark.parametrize("P,C,H", [(70, 64, 256), (33, 32, 128), (64, 64, 128), (40, 32, 64), (5000, 64, 256)])
def test_ffx_fwd_bwd(be, P, C, H):
    if be.kind == "emu" and P > 1000:
        pytest.skip("large case runs on the GPU only")
    lib, p = be.lib, be.ptr
    assert lib.ffno_ffx_supported(C, H) == 1
    rs = np.random.RandomState(P + C + H)
    s = rs.standard_normal((P, C)).astype(np.float32)
    resid = rs.standard_normal((P, C)).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = (rs.standard_normal(H) * 0.1).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = (rs.standard_normal(C) * 0.1).astype(np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights(be, W1, W2)
    ds_, dres, db1_, db2_ = map(be.put, (s, resid, b1, b2))
    out = be.empty((P, C))
    mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    assert lib.ffno_ffx_fwd(p(ds_), p(dres), p(a1), p(db1_), p(a2), p(db2_), p(out), p(mask), P, C, H, None) == 0
    ref_out, ref_h = ff_ref(s, resid, W1, b1, W2, b2)
    assert rel_l2(be.get(out), ref_out) < TOL
    # no residual, no mask
    out2 = be.empty((P, C))
    assert lib.ffno_ffx_fwd(p(ds_), None, p(a1), p(db1_), p(a2), p(db2_), p(out2), None, P, C, H, None) == 0
    assert rel_l2(be.get(out2), ref_out - resid) < TOL
    # in place: out aliases resid
    res2 = be.put(resid)
    assert lib.ffno_ffx_fwd(p(ds_), p(res2), p(a1), p(db1_), p(a2), p(db2_), p(res2), None, P, C, H, None) == 0
    assert rel_l2(be.get(res2), ref_out) < TOL

    # backward (data) from the saved sign bits
    db = rs.standard_normal((P, C)).astype(np.float32)
    ddb, ds = be.put(db), be.empty((P, C))
    assert lib.ffno_ffx_bwd_data(p(ddb), p(mask), p(a1b), p(a2b), p(ds), P, C, H, None) == 0
    ref_dh = (db.astype(np.float64) @ W2.astype(np.float64)) * (ref_h > 0)
    ref_ds = ref_dh @ W1.astype(np.float64)
    assert rel_l2(be.get(ds), ref_ds) < TOL

    # backward (weights) with recomputed h / dh, deterministic two-step reduction
    nsplit = 3 if P < 1000 else 64
    partial = be.zeros(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit))
    assert lib.ffno_ffx_bwd_weights_partial(p(ds_), p(ddb), p(a1), p(db1_), p(a1b), p(partial), P, C, H, nsplit, None) == 0
    gW1, gW2, gb1, gb2 = be.zeros((H, C)), be.zeros((C, H)), be.zeros(H), be.zeros(C)
    assert lib.ffno_ffx_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, nsplit, 0, None) == 0
    assert rel_l2(be.get(gW1), ref_dh.T @ s.astype(np.float64)) < TOL
    assert rel_l2(be.get(gW2), db.astype(np.float64).T @ ref_h) < TOL
    assert rel_l2(be.get(gb1), ref_dh.sum(0)) < TOL
    assert rel_l2(be.get(gb2), db.astype(np.float64).sum(0)) < TOL
    assert lib.ffno_ffx_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, nsplit, 1, None) == 0
    assert rel_l2(be.get(gW1), 2 * ref_dh.T @ s.astype(np.float64)) < TOL


def test_ffx_wide_dynamic_range(be):
    """Tiny gradients and large activations in one call: bf16 keeps the fp32 exponent range, so no scaling is needed."""
    lib, p = be.lib, be.ptr
    P, C, H = 96, 64, 256
    rs = np.random.RandomState(5)
    s = (rs.standard_normal((P, C)) * np.exp(rs.uniform(-3, 6, (P, 1)))).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = (rs.standard_normal(H) * 0.1).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = np.zeros(C, np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights(be, W1, W2)
    ds_, db1_, db2_ = map(be.put, (s, b1, b2))
    out = be.empty((P, C))
    mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    assert lib.ffno_ffx_fwd(p(ds_), None, p(a1), p(db1_), p(a2), p(db2_), p(out), p(mask), P, C, H, None) == 0
    ref_out, ref_h = ff_ref(s, None, W1, b1, W2, b2)
    err_rows = np.linalg.norm(be.get(out) - ref_out, axis=1) / np.linalg.norm(ref_out, axis=1)
    assert err_rows.max() < TOL                       # per pixel, whatever its scale
    db = (rs.standard_normal((P, C)) * 1e-9).astype(np.float32)
    ddb, ds = be.put(db), be.empty((P, C))
    assert lib.ffno_ffx_bwd_data(p(ddb), p(mask), p(a1b), p(a2b), p(ds), P, C, H, None) == 0
    ref_ds = ((db.astype(np.float64) @ W2.astype(np.float64)) * (ref_h > 0)) @ W1.astype(np.float64)
    assert rel_l2(be.get(ds), ref_ds) < TOL


def test_ffx_rejects_unsupported_shapes(be):
    z = be.zeros(64)
    p = be.ptr
    assert be.lib.ffno_ffx_supported(48, 192) == 0
    assert be.lib.ffno_ffx_fwd(p(z), None, p(z), p(z), p(z), p(z), p(z), None, 1, 48, 192, None) == -2
    assert be.lib.ffno_ffx_fwd(None, None, p(z), p(z), p(z), p(z), p(z), None, 1, 64, 256, None) == -1


def test_ffx_two_input_variants_equal_the_presummed_call(be):
    """ffno_ffx_fwd2 / ffno_ffx_bwd_data2 (input = sum of the two spectral branch buffers, optionally stored back) are
    bit-identical to summing first and calling the one-input entry points."""
    lib, p = be.lib, be.ptr
    P, C, H = 100, 64, 256
    rs = np.random.RandomState(11)
    sa, sb = (rs.standard_normal((P, C)).astype(np.float32) for _ in range(2))
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b1, b2 = (rs.standard_normal(H) * 0.1).astype(np.float32), (rs.standard_normal(C) * 0.1).astype(np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights(be, W1, W2)
    db1_, db2_ = be.put(b1), be.put(b2)
    ssum_host = sa + sb
    mask_a = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    mask_b = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    out_a, out_b, ssum = be.empty((P, C)), be.empty((P, C)), be.empty((P, C))
    assert lib.ffno_ffx_fwd(p(be.put(ssum_host)), None, p(a1), p(db1_), p(a2), p(db2_), p(out_a), p(mask_a), P, C, H, None) == 0
    assert lib.ffno_ffx_fwd2(p(be.put(sa)), p(be.put(sb)), p(ssum), None, p(a1), p(db1_), p(a2), p(db2_), p(out_b), p(mask_b),
                             P, C, H, None, None) == 0
    np.testing.assert_array_equal(be.get(ssum), ssum_host)
    np.testing.assert_array_equal(be.get(out_a), be.get(out_b))
    np.testing.assert_array_equal(be.get(mask_a), be.get(mask_b))
    ds_a, ds_b, gsum = be.empty((P, C)), be.empty((P, C)), be.empty((P, C))
    assert lib.ffno_ffx_bwd_data(p(be.put(ssum_host)), p(mask_a), p(a1b), p(a2b), p(ds_a), P, C, H, None) == 0
    assert lib.ffno_ffx_bwd_data2(p(be.put(sa)), p(be.put(sb)), p(gsum), p(mask_a), p(a1b), p(a2b), p(ds_b), P, C, H, None, None) == 0
    np.testing.assert_array_equal(be.get(ds_a), be.get(ds_b))
    np.testing.assert_array_equal(be.get(gsum), ssum_host)
    assert lib.ffno_ffx_fwd2(p(be.put(sa)), None, p(ssum), None, p(a1), p(db1_), p(a2), p(db2_), p(out_b), None, P, C, H, None, None) == -1


@pytest.mark.parametrize("P,C,H,wgs", [(200, 64, 256, 2), (150, 32, 128, 1), (97, 64, 128, 3), (400, 64, 256, 1), (384, 64, 256, 2), (5000, 64, 256, 0), (131072, 64, 256, 0)])
def test_ffx_chain_schedules_are_bit_identical(be, P, C, H, wgs):
    """The role-split forward schedule (the default) against the in-phase kernel (FFNO_FF_SCHED_IN_PHASE): same products in the
    same order, so outputs, stored sums and sign words must be IDENTICAL -- with several tiles per persistent workgroup (per-call
    ``max_workgroups``: loads two tiles ahead, residual rows / sign words one tile ahead), ragged last tile, in-place residual.
    The options are arguments of the call: nothing process-wide is switched (re-entrant)."""
    if be.kind == "emu" and P > 1000:
        pytest.skip("large case runs on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P + H)
    sa, sb, resid, db = (rs.standard_normal((P, C)).astype(np.float32) for _ in range(4))
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b1, b2 = (rs.standard_normal(H) * 0.1).astype(np.float32), (rs.standard_normal(C) * 0.1).astype(np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights(be, W1, W2)
    db1_, db2_ = be.put(b1), be.put(b2)
    res = {}
    for sched in (0, 1):
        mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
        ssum, x = be.empty((P, C)), be.put(resid)          # out aliases resid (the layer's x <- x + b update)
        word = be.zeros(1, np.uint32)
        o = FfOpts(None, p(word), wgs, sched)
        assert lib.ffno_ffx_fwd2(p(be.put(sa)), p(be.put(sb)), p(ssum), p(x), p(a1), p(db1_), p(a2), p(db2_), p(x), p(mask),
                                 P, C, H, ctypes.byref(o), None) == 0
        gsum, ds = be.empty((P, C)), be.empty((P, C))
        o2 = FfOpts(None, None, wgs, 0)
        assert lib.ffno_ffx_bwd_data2(p(be.put(db)), p(be.put(sa)), p(gsum), p(mask), p(a1b), p(a2b), p(ds), P, C, H,
                                      ctypes.byref(o2), None) == 0
        res[sched] = [np.array(be.get(t)).copy() for t in (ssum, x, mask, gsum, ds, word)]
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)
    ref_out, _ = ff_ref(sa + sb, resid, W1, b1, W2, b2)
    assert rel_l2(res[0][1], ref_out) < TOL
    # the bf16x3 family needs no range word but records its output maximum like the fp16x2 one (mixed configurations)
    assert res[0][5].view(np.float32)[0] == np.abs(res[0][1]).max()
