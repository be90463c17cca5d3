"""fp16x2 feed-forward kernels (ffno_ffh_*, fourierflow_amd/csrc/ffx.hip with the SplitHf2 policy) through the C ABI vs fp64
numpy references -- on the CPU wave emulator (-m "not gpu") and on the MI355X (-m gpu).  Same operator, masks and tolerance as
the bf16x3 family (test_kernels_ffx.py); what differs is the operand format, so the range behaviour is tested as well."""
import ctypes

import numpy as np
import pytest

from backend_util import be, rel_l2  # noqa: F401
from fourierflow_amd._capi import FfOpts, FxPackDesc
from test_kernels_ff import ff_ref

FFNO_FF_SCHED_IN_PHASE = 1

TOL = 1e-5


def pack_weights_h(be, W1, W2):
    lib, p = be.lib, be.ptr
    H, C = W1.shape
    nbytes = lib.ffno_ffh_pack_bytes(C, H)
    assert nbytes == C * H * 4
    dW1, dW2 = be.put(W1), be.put(W2)
    bufs = [be.zeros(nbytes // 4, np.uint32) for _ in range(4)]
    spec = [(dW1, C, 1, 1), (dW2, 1, H, 2), (dW2, 1, H, 1), (dW1, C, 1, 2)]
    descs = (FxPackDesc * 4)(*[FxPackDesc(p(src), p(dst), sh, sc, ty, 0) for (src, sh, sc, ty), dst in zip(spec, bufs)])
    table = be.put(np.frombuffer(bytes(descs), dtype=np.uint8))
    assert lib.ffno_ffh_pack(p(table), 4, C, H, None) == 0
    be.get(bufs[0])   # sync
    return bufs, (dW1, dW2, table)


def test_split2_pack_layout_and_accuracy(be):
    """hi + lo / 2^11 of a packed weight reproduces the fp32 value to max(2^-24 |x|, 2^-36), at the documented
    fragment positions (the bf16x3 layout with two planes)."""
    C, H = 64, 256
    rs = np.random.RandomState(0)
    W1 = (rs.standard_normal((H, C)) * np.exp(rs.uniform(-6, 6, (H, C)))).astype(np.float32)     # 2.5e-3 .. 4e2
    W2 = rs.standard_normal((C, H)).astype(np.float32)
    bufs, _keep = pack_weights_h(be, W1, W2)
    raw = be.get(bufs[0]).view(np.float16).reshape((H // 32) * (C // 16), 2, 64, 8)
    val = raw[:, 0].astype(np.float64) + raw[:, 1].astype(np.float64) / 2048.0
    lane = np.arange(64)
    j, half, e = (lane & 31)[:, None], (lane >> 5)[:, None], np.arange(8)[None, :]
    a1 = val.reshape(H // 32, C // 16, 64, 8)
    for w in (0, 3, 7):
        for st in range(C // 16):
            ref = W1[32 * w + j, 16 * st + 8 * half + e].astype(np.float64)
            err, big = np.abs(a1[w, st] - ref), np.abs(ref) >= 2.0 ** -12      # both planes are normal halves there
            assert np.max(err[big] / np.abs(ref[big])) <= 2.0 ** -23
            assert np.all(err[~big] <= 2.0 ** -35)


def amax_word(be, *tensors):
    """A range word holding max |x| over the given host tensors, folded on the device by ffno_amax."""
    w = be.zeros(1, np.uint32)
    for t in tensors:
        assert be.lib.ffno_amax(be.ptr(be.put(t)), t.size, be.ptr(w), None) == 0
    got = np.asarray(be.get(w)).view(np.uint32)[0]
    assert got == np.float32(max(np.abs(t).max() for t in tensors)).view(np.uint32)
    return w


def opts(be, in_amax=None, out_amax=None, max_workgroups=0, schedule=0):
    return FfOpts(be.ptr(in_amax), be.ptr(out_amax), max_workgroups, schedule)


def word_value(be, w):
    return float(np.asarray(be.get(w)).view(np.float32)[0])


def test_amax_folds_and_accumulates(be):
    """ffno_amax: word = max(word, bits(max |x|)) for any length / alignment tail; a second call only raises it."""
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(3)
    w = be.zeros(1, np.uint32)
    for n, mag in ((7, 1.0), (4099, 0.5), (20000, 3.0)):
        x = (rs.standard_normal(n) * mag).astype(np.float32)
        before = word_value(be, w)
        assert lib.ffno_amax(p(be.put(x)), n, p(w), None) == 0
        assert word_value(be, w) == max(before, float(np.abs(x).max()))
    assert lib.ffno_amax(None, 4, p(w), None) == -1 and lib.ffno_amax(p(w), 0, p(w), None) == -1


def test_amax_batched_folds_every_tensor(be):
    """ffno_amax_batched: n tensors (any lengths) folded into one word by one launch -- what the engine's weight-range check issues."""
    from fourierflow_amd._capi import AmaxDesc
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(7)
    xs = [rs.standard_normal(n).astype(np.float32) * sc for n, sc in ((5, 1.0), (4099, 3.0), (70000, 0.5), (1, 2.0))]
    xs[1][77] = -41.5
    devs = [be.put(x) for x in xs]
    descs = (AmaxDesc * len(xs))(*[AmaxDesc(p(d), x.size) for d, x in zip(devs, xs)])
    table = be.put(np.frombuffer(bytes(descs), dtype=np.uint8))
    w = be.zeros(1, np.uint32)
    assert lib.ffno_amax_batched(p(table), len(xs), max(x.size for x in xs), p(w), None) == 0
    assert np.asarray(be.get(w)).view(np.float32)[0] == 41.5
    # the maximum at ANY position is found: last element of a long tensor (its tail past the 16-byte loads), of a short one, and in a
    # tensor that does not start on a 16-byte boundary (a view one float into a buffer) -- one launch each, fresh word
    big = be.put(np.zeros(70001, np.float32))
    for which, pos, off, n in ((2, 69999, 0, 70000), (1, 4098, 0, 4099), (2, 69996, 1, 69999), (0, 4, 0, 5)):
        ys = [x.copy() for x in xs]
        ys[which][:] *= 1e-3
        ys[which] = ys[which][:n]
        ys[which][pos] = 123.25
        devs2 = [be.put(y) for y in ys]
        ptrs = [p(d) for d in devs2]
        if off:      # unaligned start: the tensor lives one float into a device buffer
            full = np.zeros(n + off, np.float32)
            full[off:] = ys[which]
            devs2[which] = be.put(full)
            base = p(devs2[which])
            ptrs[which] = (base.value if hasattr(base, "value") else int(base)) + 4 * off
        descs2 = (AmaxDesc * len(ys))(*[AmaxDesc(pt, y.size) for pt, y in zip(ptrs, ys)])
        table2 = be.put(np.frombuffer(bytes(descs2), dtype=np.uint8))
        w2 = be.zeros(1, np.uint32)
        assert lib.ffno_amax_batched(p(table2), len(ys), max(y.size for y in ys), p(w2), None) == 0
        assert np.asarray(be.get(w2)).view(np.float32)[0] == 123.25, (which, pos, off)
    assert lib.ffno_amax_batched(None, 1, 4, p(w), None) == -1 and lib.ffno_amax_batched(p(table), 0, 4, p(w), None) == -1


@pytest.mark.parametrize("sched", [0, FFNO_FF_SCHED_IN_PHASE])
@pytest.mark.parametrize("P,C,H", [(70, 64, 256), (33, 32, 128), (64, 64, 128), (40, 32, 64), (5000, 64, 256), (200, 64, 256)])
def test_ffh_fwd_bwd(be, P, C, H, sched):
    if be.kind == "emu" and (P > 1000 or (sched and (C, H) != (64, 256))):
        pytest.skip("large case / most of the schedule sweep run on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P + C + H)
    sa, sb = (rs.standard_normal((P, C)).astype(np.float32) for _ in range(2))
    s = sa + sb
    resid = rs.standard_normal((P, C)).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = (rs.standard_normal(H) * 0.1).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = (rs.standard_normal(C) * 0.1).astype(np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights_h(be, W1, W2)
    db1_, db2_ = be.put(b1), be.put(b2)
    out, ssum = be.put(resid), be.empty((P, C))              # out aliases the residual
    mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    s_word, out_word = amax_word(be, sa, sb), be.zeros(1, np.uint32)
    o = opts(be, s_word, out_word, schedule=sched)
    assert lib.ffno_ffh_fwd2(p(be.put(sa)), p(be.put(sb)), p(ssum), p(out), p(a1), p(db1_), p(a2), p(db2_), p(out), p(mask),
                             P, C, H, ctypes.byref(o), None) == 0
    ref_out, ref_h = ff_ref(s, resid, W1, b1, W2, b2)
    assert rel_l2(be.get(out), ref_out) < TOL
    np.testing.assert_array_equal(be.get(ssum), s)
    assert word_value(be, out_word) == float(np.abs(be.get(out)).max())     # the producer recorded its output maximum
    # one addend, no residual, no mask, no range word (the data is O(1): in range as it is)
    out2 = be.empty((P, C))
    assert lib.ffno_ffh_fwd2(p(be.put(s)), None, None, None, p(a1), p(db1_), p(a2), p(db2_), p(out2), None, P, C, H, None, None) == 0
    assert rel_l2(be.get(out2), ref_out - resid) < TOL

    # backward: tiny gradients, brought into the half range through the range word of their producer
    ga, gb = ((rs.standard_normal((P, C)) * 4e-6).astype(np.float32) for _ in range(2))
    db = ga + gb
    g_word, ds_word = amax_word(be, ga, gb), be.zeros(1, np.uint32)
    o = opts(be, g_word, ds_word)
    gsum, ds = be.empty((P, C)), be.empty((P, C))
    assert lib.ffno_ffh_bwd_data2(p(be.put(ga)), p(be.put(gb)), p(gsum), p(mask), p(a1b), p(a2b), p(ds), P, C, H, ctypes.byref(o), None) == 0
    np.testing.assert_array_equal(be.get(gsum), db)           # the stored sum is NOT scaled
    ref_dh = (db.astype(np.float64) @ W2.astype(np.float64)) * (ref_h > 0)
    ref_ds = ref_dh @ W1.astype(np.float64)
    assert rel_l2(be.get(ds), ref_ds) < TOL
    assert word_value(be, ds_word) == float(np.abs(be.get(ds)).max())
    nsplit = 2 if P == 200 else 3 if P < 1000 else 64      # (200 pixels on 2 workgroups: 4 + 3 tiles each, ragged last tile)
    partial = be.zeros(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit))
    assert lib.ffno_ffh_bwd_weights_partial(p(ssum), p(gsum), p(a1), p(db1_), p(a1b), p(partial), P, C, H, nsplit, p(s_word), p(g_word), 0, None) == 0
    gW1, gW2, gb1, gb2 = be.zeros((H, C)), be.zeros((C, H)), be.zeros(H), be.zeros(C)
    assert lib.ffno_ffx_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, nsplit, 0, None) == 0
    assert rel_l2(be.get(gW1), ref_dh.T @ s.astype(np.float64)) < TOL
    assert rel_l2(be.get(gW2), db.astype(np.float64).T @ ref_h) < TOL
    assert rel_l2(be.get(gb1), ref_dh.sum(0)) < TOL
    assert rel_l2(be.get(gb2), db.astype(np.float64).sum(0)) < TOL
    # without a range word the same call still works, only less accurately (documented: gradual below 6e-5)
    assert lib.ffno_ffh_bwd_data2(p(be.put(db)), None, None, p(mask), p(a1b), p(a2b), p(ds), P, C, H, None, None) == 0
    assert rel_l2(be.get(ds), ref_ds) < 3e-5


FFNO_FF_SCHED_WAVE_TILES, FFNO_FF_SCHED_ROLE_SPLIT = 2, 3


@pytest.mark.parametrize("P,two", [(70, True), (200, False), (32 * 21, True), (5000, True)])
def test_ffh_wave_tiles_against_shared_tiles(be, P, two):
    """FFNO_FF_SCHED_WAVE_TILES (one wave per 32-pixel tile, both weight maps in LDS; what schedule 0 picks for large launches at
    64 / 256) against the shared-tile kernels on the same operands: the stored input sum and the ReLU sign words are the same
    bits (the first product chain is the same, product for product), outputs and data gradients agree to fp32 rounding (same
    products; the hidden chunks of an output are summed in one accumulator pair instead of eight partial tiles), both against
    fp64.  Ragged last tile, one and two addends, more tiles than waves of a workgroup, fewer tiles than waves."""
    if be.kind == "emu" and P > 1000:
        pytest.skip("large case on the GPU only")
    C, H = 64, 256
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P)
    sa, sb = (rs.standard_normal((P, C)).astype(np.float32) * 30 for _ in range(2))
    s = sa + sb if two else sa
    resid = rs.standard_normal((P, C)).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = (rs.standard_normal(H) * 0.1).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = (rs.standard_normal(C) * 0.1).astype(np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights_h(be, W1, W2)
    db1_, db2_ = be.put(b1), be.put(b2)
    s_word = amax_word(be, sa, sb) if two else amax_word(be, sa)
    ga, gb = ((rs.standard_normal((P, C)) * 3e-3).astype(np.float32) for _ in range(2))
    g = ga + gb if two else ga
    g_word = amax_word(be, ga, gb) if two else amax_word(be, ga)
    got = {}
    dsa, dsb, dres, dga, dgb = be.put(sa), be.put(sb), be.put(resid), be.put(ga), be.put(gb)      # (kept alive across the launches)
    for sched in (FFNO_FF_SCHED_ROLE_SPLIT, FFNO_FF_SCHED_WAVE_TILES):
        out, ssum = be.empty((P, C)), be.empty((P, C))
        mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
        ow, dw = be.zeros(1, np.uint32), be.zeros(1, np.uint32)
        o = opts(be, s_word, ow, schedule=sched)
        assert lib.ffno_ffh_fwd2(p(dsa), p(dsb) if two else None, p(ssum) if two else None, p(dres), p(a1),
                                 p(db1_), p(a2), p(db2_), p(out), p(mask), P, C, H, ctypes.byref(o), None) == 0
        gsum, ds = be.empty((P, C)), be.empty((P, C))
        o = opts(be, g_word, dw, schedule=sched)
        assert lib.ffno_ffh_bwd_data2(p(dga), p(dgb) if two else None, p(gsum) if two else None, p(mask), p(a1b),
                                      p(a2b), p(ds), P, C, H, ctypes.byref(o), None) == 0
        got[sched] = (be.get(out).copy(), be.get(ssum).copy(), np.asarray(be.get(mask)).copy(), be.get(ds).copy(),
                      be.get(gsum).copy(), word_value(be, ow), word_value(be, dw))
    ref_out, ref_h = ff_ref(s, resid, W1, b1, W2, b2)
    ref_ds = ((g.astype(np.float64) @ W2.astype(np.float64)) * (ref_h > 0)) @ W1.astype(np.float64)
    old, new = got[FFNO_FF_SCHED_ROLE_SPLIT], got[FFNO_FF_SCHED_WAVE_TILES]
    assert rel_l2(new[0], ref_out) < TOL and rel_l2(new[3], ref_ds) < TOL
    assert rel_l2(new[0], old[0]) < 5e-7 and rel_l2(new[3], old[3]) < 5e-7
    np.testing.assert_array_equal(new[2], old[2])
    if two:
        np.testing.assert_array_equal(new[1], old[1])
        np.testing.assert_array_equal(new[4], old[4])
        np.testing.assert_array_equal(new[1], s)
    assert new[5] == float(np.abs(new[0]).max()) and new[6] == float(np.abs(new[3]).max())
    # a shape without wave tiles, an unknown schedule
    o = opts(be, s_word, None, schedule=FFNO_FF_SCHED_WAVE_TILES)
    scratch = be.empty((P, C))
    assert lib.ffno_ffh_fwd2(p(dsa), None, None, None, p(a1), p(db1_), p(a2), p(db2_), p(scratch), None, P, 64, 128,
                             ctypes.byref(o), None) == -2
    o = opts(be, s_word, None, schedule=7)
    assert lib.ffno_ffh_fwd2(p(dsa), None, None, None, p(a1), p(db1_), p(a2), p(db2_), p(scratch), None, P, C, H,
                             ctypes.byref(o), None) == -1


@pytest.mark.parametrize("P,C,H,nsplit", [(200, 64, 256, 2), (131, 32, 128, 3), (4100, 64, 256, 5)])
def test_ffh_weight_gradient_of_several_blocks_in_one_launch(be, P, C, H, nsplit):
    """ffno_ffh_bwd_weights_partial_multi: the slices of n feed-forward blocks (own inputs, gradients, weights, range words) from
    ONE launch equal the n per-block launches with the same slice count bit for bit -- the same kernel body, the workgroup only
    finds its block and slice from the table.  Ragged last tile, a slice count that does not divide the tile count."""
    from fourierflow_amd._capi import FfWgDesc
    if be.kind == "emu" and P > 1000:
        pytest.skip("large case on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P + nsplit)
    n = 3
    nfl = int(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit))
    keep, descs, singles, descs2, twos, presum_g = [], [], [], [], [], []
    for i in range(n):
        s = (rs.standard_normal((P, C)) * 10.0 ** (i - 1)).astype(np.float32)
        g = (rs.standard_normal((P, C)) * 10.0 ** (-3 * i)).astype(np.float32)
        W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
        W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
        b1 = be.put((rs.standard_normal(H) * 0.1).astype(np.float32))
        (a1, a2, a1b, a2b), k_ = pack_weights_h(be, W1, W2)
        ds_, dg = be.put(s), be.put(g)
        sw, gw = amax_word(be, s), amax_word(be, g)
        one, multi = be.zeros(nfl), be.zeros(nfl)
        assert lib.ffno_ffh_bwd_weights_partial(p(ds_), p(dg), p(a1), p(b1), p(a1b), p(one), P, C, H, nsplit, p(sw), p(gw), 0, None) == 0
        singles.append((one, multi))
        descs.append(FfWgDesc(p(ds_), p(dg), p(a1), p(b1), p(a1b), p(multi), p(sw), p(gw)))
        keep += [ds_, dg, b1, a1, a2, a1b, a2b, k_, sw, gw]
        # the same block as the sum of two addends (two_addends launch): s = sa + sb, g = ga + gb formed while staging
        sa = (s * rs.uniform(0.2, 0.8, s.shape)).astype(np.float32)
        ga = (g * rs.uniform(0.2, 0.8, g.shape)).astype(np.float32)
        sb, gb = (s - sa).astype(np.float32), (g - ga).astype(np.float32)
        if i == n - 1:
            ga, gb = g, np.zeros_like(g)          # a block with one gradient addend passes zeros
        pre_s, pre_g = be.put(sa + sb), be.put(ga + gb)
        one2, multi2 = be.zeros(nfl), be.zeros(nfl)
        assert lib.ffno_ffh_bwd_weights_partial(p(pre_s), p(pre_g), p(a1), p(b1), p(a1b), p(one2), P, C, H, nsplit, p(sw), p(gw), 0, None) == 0
        dsa, dsb, dga, dgb = be.put(sa), be.put(sb), be.put(ga), be.put(gb)
        twos.append((one2, multi2))
        descs2.append(FfWgDesc(p(dsa), p(dga), p(a1), p(b1), p(a1b), p(multi2), p(sw), p(gw), p(dsb), p(dgb)))
        keep += [pre_s, pre_g, dsa, dsb, dga, dgb]
        presum_g.append(pre_g)
    table = (FfWgDesc * n)(*descs)          # (a host array: copied into the kernel arguments at enqueue)
    assert lib.ffno_ffh_bwd_weights_partial_multi(table, n, P, C, H, nsplit, 0, 0, None) == 0
    for one, multi in singles:
        a = be.get(one)
        assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
        np.testing.assert_array_equal(be.get(multi), a)
    table2 = (FfWgDesc * n)(*descs2)
    if C == 64:
        # two_addends = 1: s is a sum formed while staging -- the same blocks with the pre-summed gradient in the first slot
        for d, pg in zip(descs2, presum_g):
            d.g = p(pg)
        table3 = (FfWgDesc * n)(*descs2)
        assert lib.ffno_ffh_bwd_weights_partial_multi(table3, n, P, C, H, nsplit, 0, 1, None) == 0
        for one2, multi2 in twos:
            np.testing.assert_array_equal(be.get(multi2), be.get(one2))
        assert lib.ffno_ffh_bwd_weights_partial_multi(table3, n, P, C, H, nsplit, 0, 2, None) == -1      # (round 4's second gradient addend: removed)
        assert lib.ffno_ffh_bwd_weights_partial_multi(table3, n, P, C, H, nsplit, 0, 3, None) == -1
    else:
        assert lib.ffno_ffh_bwd_weights_partial_multi(table2, n, P, C, H, nsplit, 0, 1, None) == -2
    assert lib.ffno_ffh_bwd_weights_partial_multi(None, n, P, C, H, nsplit, 0, 0, None) == -1
    assert lib.ffno_ffh_bwd_weights_partial_multi(table, n, P, 64, 128, nsplit, 0, 0, None) == -2


@pytest.mark.parametrize("act,grad", [(1e5, 1e4), (1e6, 1e-12), (3e-9, 7e7), (1.0, 1.0)])
def test_ffh_any_fp32_magnitude_is_in_range(be, act, grad):
    """activations of 1e5 / 1e6 (>= 65504 used to become inf in the fp16 planes) and gradients of any
    magnitude -- forward, data gradient and weight gradients stay at the 1e-5 bar of the O(1) case, nothing is inf / nan,
    because every kernel scales its staged rows from the range word of its input (a power of two derived on the device)."""
    lib, p = be.lib, be.ptr
    P, C, H = 70, 64, 256
    rs = np.random.RandomState(11)
    sa, sb = ((rs.standard_normal((P, C)) * act).astype(np.float32) for _ in range(2))
    s = sa + sb
    resid = (rs.standard_normal((P, C)) * act).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = (rs.standard_normal(H) * 0.1 * act).astype(np.float32)       # a bias that matters at this magnitude
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = (rs.standard_normal(C) * 0.1 * act).astype(np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights_h(be, W1, W2)
    db1_, db2_ = be.put(b1), be.put(b2)
    out, ssum = be.empty((P, C)), be.empty((P, C))
    mask = be.zeros(lib.ffno_ff_mask_words(P, H), np.uint32)
    s_word = amax_word(be, sa, sb)
    o = opts(be, s_word, None)
    assert lib.ffno_ffh_fwd2(p(be.put(sa)), p(be.put(sb)), p(ssum), p(be.put(resid)), p(a1), p(db1_), p(a2), p(db2_), p(out), p(mask),
                             P, C, H, ctypes.byref(o), None) == 0
    ref_out, ref_h = ff_ref(s, resid, W1, b1, W2, b2)
    got = be.get(out)
    assert np.all(np.isfinite(got)) and rel_l2(got, ref_out) < TOL
    ga, gb = ((rs.standard_normal((P, C)) * grad).astype(np.float32) for _ in range(2))
    db = ga + gb
    g_word = amax_word(be, ga, gb)
    o = opts(be, g_word, None)
    gsum, ds = be.empty((P, C)), be.empty((P, C))
    assert lib.ffno_ffh_bwd_data2(p(be.put(ga)), p(be.put(gb)), p(gsum), p(mask), p(a1b), p(a2b), p(ds), P, C, H, ctypes.byref(o), None) == 0
    ref_dh = (db.astype(np.float64) @ W2.astype(np.float64)) * (ref_h > 0)
    got = be.get(ds)
    assert np.all(np.isfinite(got)) and rel_l2(got, ref_dh @ W1.astype(np.float64)) < TOL
    partial = be.zeros(lib.ffno_ff_wgrad_partial_floats(C, H, 3))
    assert lib.ffno_ffh_bwd_weights_partial(p(ssum), p(gsum), p(a1), p(db1_), p(a1b), p(partial), P, C, H, 3, p(s_word), p(g_word), 0, None) == 0
    gW1, gW2, gb1, gb2 = be.zeros((H, C)), be.zeros((C, H)), be.zeros(H), be.zeros(C)
    assert lib.ffno_ffx_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, 3, 0, None) == 0
    for got, ref in ((gW1, ref_dh.T @ s.astype(np.float64)), (gW2, db.astype(np.float64).T @ ref_h), (gb1, ref_dh.sum(0)),
                     (gb2, db.astype(np.float64).sum(0))):
        g = be.get(got)
        assert np.all(np.isfinite(g)) and rel_l2(g, ref) < TOL


def test_ffh_accuracy_over_the_half_range(be):
    """Per-pixel relative error of the forward stays at fp32 level for activations from 1e-3 to 1e3 (hi is a normal half
    throughout; the scaled lo plane carries the next 11 bits whatever the magnitude)."""
    lib, p = be.lib, be.ptr
    P, C, H = 96, 64, 256
    rs = np.random.RandomState(5)
    s = (rs.standard_normal((P, C)) * np.exp(rs.uniform(-7, 7, (P, 1)))).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / np.sqrt(C)).astype(np.float32)
    b1 = np.zeros(H, np.float32)
    W2 = (rs.standard_normal((C, H)) / np.sqrt(H)).astype(np.float32)
    b2 = np.zeros(C, np.float32)
    (a1, a2, a1b, a2b), _keep = pack_weights_h(be, W1, W2)
    out = be.empty((P, C))
    assert lib.ffno_ffh_fwd2(p(be.put(s)), None, None, None, p(a1), p(be.put(b1)), p(a2), p(be.put(b2)), p(out), None, P, C, H, None, None) == 0
    ref_out, _ = ff_ref(s, None, W1, b1, W2, b2)
    err_rows = np.linalg.norm(be.get(out) - ref_out, axis=1) / np.linalg.norm(ref_out, axis=1)
    assert err_rows.max() < 2e-6, err_rows.max()


def test_ffh_rejects_bad_arguments(be):
    z = be.zeros(64)
    p = be.ptr
    assert be.lib.ffno_ffh_fwd2(p(z), None, None, None, p(z), p(z), p(z), p(z), p(z), None, 1, 48, 192, None, None) == -2
    assert be.lib.ffno_ffh_bwd_data2(None, None, None, p(z), p(z), p(z), p(z), 1, 64, 256, None, None) == -1


def _tr16_model(image, byte_off):
    """ds_read_b64_tr_b16 as csrc/ffno_platform.h documents it: within each 16-lane group, lane i receives element i & 3 of the
    8-byte pieces addressed by lanes 4 r + (i >> 2), r = 0..3."""
    out = np.zeros((64, 4), np.uint16)
    for lane in range(64):
        grp, i = lane & ~15, lane & 15
        for r in range(4):
            out[lane, r] = image[byte_off[grp + 4 * r + (i >> 2)] // 2 + (i & 3)]
    return out


@pytest.mark.parametrize("pattern", ["linear", "wgrad64", "wgrad32", "random"])
def test_lds_transpose_read_map(be, pattern):
    """The LDS transpose read the weight-gradient kernel takes its channel-major operands through (ffx.hip `tfrag`), lane by lane:
    the hardware (-m gpu) and the emulator (-m "not gpu") against the documented map -- for lane-linear addresses (the canonical
    [4][16] block), for the kernel's own address pattern at both widths (the channels of four pixel rows of the skewed pixel-major
    plane -> MFMA A-operand slots) and for random 8-byte-aligned addresses."""
    lib, p = be.lib, be.ptr
    n16 = 8192
    image = np.arange(n16, dtype=np.uint16) * 7 + 3
    rs = np.random.RandomState(5)
    lane = np.arange(64)
    if pattern == "linear":
        off = 8 * lane
    elif pattern == "random":
        off = 8 * rs.randint(0, n16 * 2 // 8, 64)
    else:
        prow = 192 if pattern == "wgrad64" else 64
        half, g, i = lane >> 5, (lane >> 4) & 1, lane & 15
        off = (4 * half + (i >> 2)) * prow + 16 * half + 32 * g + 8 * (i & 3) + (8 * prow + 32)      # (second read of a fragment)
    off = off.astype(np.int32)
    out = be.zeros((64, 4), np.uint16)
    assert lib.ffno_lds_tr16_probe(p(be.put(image)), n16, p(be.put(off)), p(out), None) == 0
    got = be.get(out).reshape(64, 4)
    np.testing.assert_array_equal(got, _tr16_model(image, off))
    if pattern.startswith("wgrad"):
        # what the kernel relies on: lane l holds channel 16 g + i of pixel rows 8 + 4 half + {0, 1, 2, 3} of the skewed image
        prow = 192 if pattern == "wgrad64" else 64
        for l in range(64):
            half, ch = l >> 5, l & 31
            for r in range(4):
                R = 8 + 4 * half + r
                assert got[l, r] == image[(R * prow + 16 * ((R >> 2) & 3)) // 2 + ch]
