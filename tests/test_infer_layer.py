"""The two-launch INFERENCE layer (include/ffno.h: ffno_spectral_x3_mix_pair + ffno_infer_ff = ffno_layer_infer; csrc/infer.hip)
against an fp64 restatement of one layer of FNOFactorized2DBlock.forward (reference fourierflow/modules/factorized_fno/
grid_2d.py:42-49,51-99,169: rfft(ortho) -> [:K] -> complex einsum -> zero-filled irfft along both axes, summed;
feedforward.py:13-19: Linear -> ReLU -> Linear; + the residual), against the training layer call it replaces (ffno_layer_fwd), and
through the engine: `forward(save_for_backward=False)` -- predict, validation, rollout -- must take it and agree with the oracle.
Tolerance: 1e-5 relative L2 (BASELINE north star); observed 1e-7-level."""
import ctypes

import numpy as np
import pytest
import torch

import golden_util as gu
import oracle_util as ou
from backend_util import be, host_device, rel_l2  # noqa: F401
from test_kernels_ffh import pack_weights_h
from test_kernels_spectral import _x3_pack


def layer_fp64(x, wa, wb, W1, b1, W2, b2, K):
    """x [B, M, N, C]; wa mixes the LAST axis (lines (b, m)), wb the first (grid_2d.py:58-72 / :76-90)."""
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)                      # b i m n
    B, I, M, N = xt.shape
    wA = torch.view_as_complex(torch.tensor(wa, dtype=torch.float64).contiguous())   # [i, o, k]
    wB = torch.view_as_complex(torch.tensor(wb, dtype=torch.float64).contiguous())
    fy = torch.fft.rfft(xt, dim=-1, norm="ortho")
    oy = torch.zeros(B, I, M, N // 2 + 1, dtype=torch.complex128)
    oy[..., :K] = torch.einsum("bixy,ioy->boxy", fy[..., :K], wA)
    sy = torch.fft.irfft(oy, n=N, dim=-1, norm="ortho")
    fx = torch.fft.rfft(xt, dim=-2, norm="ortho")
    ox = torch.zeros(B, I, M // 2 + 1, N, dtype=torch.complex128)
    ox[:, :, :K] = torch.einsum("bixy,iox->boxy", fx[:, :, :K], wB)
    sx = torch.fft.irfft(ox, n=M, dim=-2, norm="ortho")
    s = (sx + sy).permute(0, 2, 3, 1)
    h = torch.relu(s @ torch.tensor(W1, dtype=torch.float64).T + torch.tensor(b1, dtype=torch.float64))
    return (torch.tensor(x, dtype=torch.float64) + h @ torch.tensor(W2, dtype=torch.float64).T + torch.tensor(b2, dtype=torch.float64)).numpy(), s.numpy()


def _setup(be, B, M, N, K, seed, scale=1.0):
    from fourierflow_amd._capi import FusedBranch
    lib, p = be.lib, be.ptr
    C, H = 64, 256
    rs = np.random.RandomState(seed)
    x = (rs.standard_normal((B, M, N, C)) * scale).astype(np.float32)
    W1 = (rs.standard_normal((H, C)) / 8).astype(np.float32)
    W2 = (rs.standard_normal((C, H)) / 16).astype(np.float32)
    b1, b2 = (rs.standard_normal(H) * 0.1).astype(np.float32), (rs.standard_normal(C) * 0.1 * scale).astype(np.float32)
    packs, keep = pack_weights_h(be, W1, W2)
    w = [(rs.standard_normal((C, C, K, 2)) / 8).astype(np.float32) for _ in range(2)]
    xp = [_x3_pack(be, w[i], K, fmt=1) for i in range(2)]
    tw = [be.twiddle(N), be.twiddle(M)]
    L = [N, M]
    tabs = []
    for i in range(2):
        nb = int(lib.ffno_spectral_x3_dft_frags_bytes(L[i], K))
        assert nb > 0
        tab = be.zeros(nb // 4, np.uint32)
        assert lib.ffno_spectral_x3_dft_frags(p(tw[i]), L[i], K, 0, 1, p(tab), None) == 0
        tabs.append(tab)
    dx = be.put(x)
    word = be.zeros(2, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(word), None) == 0
    lines = [B * M, B * N]
    mix = []
    for i in range(2):
        nb = int(lib.ffno_infer_mix_bytes(C, K, lines[i]))
        assert nb >= lines[i] * 8192 + lines[i] * 4
        mix.append(be.zeros(nb // 4, np.uint32))

    def branch(i, out, rout=None):
        return FusedBranch(p(dx), p(out), None, None, p(xp[i][0]), p(tw[i]), B, M, N, K, i, 0, 1, 0, p(word), rout, 0, 0, p(tabs[i]))

    return dict(x=x, W1=W1, W2=W2, b1=b1, b2=b2, w=w, packs=packs, dx=dx, word=word, mix=mix, branch=branch, db1=be.put(b1),
                db2=be.put(b2), keep=(keep, xp, tw, tabs))


SHAPES = [(1, 8, 32, 4), (2, 16, 64, 8), (8, 64, 64, 16), (1, 32, 128, 16), (3, 20, 32, 11)]


@pytest.mark.parametrize("B,M,N,K", SHAPES)
def test_infer_layer_vs_fp64_and_vs_the_training_layer(be, B, M, N, K):
    from fourierflow_amd._capi import LayerFwdDesc, LayerInferDesc
    if be.kind == "emu" and (B, M, N, K) in ((8, 64, 64, 16), (1, 32, 128, 16)):
        pytest.skip("emulator time budget (the GPU run covers all shapes)")
    lib, p = be.lib, be.ptr
    C, H = 64, 256
    assert lib.ffno_layer_infer_supported(B, M, N, C, H, K, K) == 1
    S = _setup(be, B, M, N, K, seed=B * 1000 + M + N + K)
    out = be.empty(S["x"].shape)
    oword = be.zeros(1, np.uint32)
    d = LayerInferDesc(S["branch"](0, S["mix"][0]), S["branch"](1, S["mix"][1]), 2, 0, p(S["packs"][0]), p(S["db1"]),
                       p(S["packs"][1]), p(S["db2"]), p(S["dx"]), p(out), C, H, p(oword))
    assert lib.ffno_layer_infer(ctypes.byref(d), None) == 0
    got = np.array(be.get(out)).copy()
    ref, s_ref = layer_fp64(S["x"], S["w"][0], S["w"][1], S["W1"], S["b1"], S["W2"], S["b2"], K)
    e = rel_l2(got, ref)
    # the update alone (out - x): the residual must not hide an error of the layer's own arithmetic
    e_upd = rel_l2(got.astype(np.float64) - S["x"], ref - S["x"])
    print(f"[infer layer {B}x{M}x{N} K={K}] out rel-L2 {e:.2e}, update rel-L2 {e_upd:.2e}")
    assert e < 1e-6 and e_upd < 1e-5
    # the range word of the output = bits of max |out|
    assert np.array(be.get(oword)).view(np.float32)[0] == np.abs(got).max()
    # ... and the training layer call on the same operands (two branch images through memory): equal to fp32 rounding
    s_img, t_img, out2 = be.empty(S["x"].shape), be.empty(S["x"].shape), be.empty(S["x"].shape)
    sword = be.zeros(1, np.uint32)
    P = B * M * N
    d2 = LayerFwdDesc(S["branch"](0, s_img, p(sword)), S["branch"](1, t_img, p(sword)), 1, 2, p(S["packs"][0]), p(S["db1"]),
                      p(S["packs"][1]), p(S["db2"]), None, p(S["dx"]), p(out2), None, P, C, H, 1, 0, 0, 0, 0, None)
    assert lib.ffno_layer_fwd(ctypes.byref(d2), None) == 0
    assert rel_l2(got, be.get(out2)) < 1e-6
    assert rel_l2(np.array(be.get(s_img)) + np.array(be.get(t_img)), s_ref) < 1e-5


@pytest.mark.parametrize("scale", [1e5, 1e-6])
def test_infer_layer_any_magnitude(be, scale):
    """Inputs far outside the half format's range: the line-wise spectrum scales of K1 and the tile-wise input scale of K2 keep
    every split operand in range (include/ffno.h "Range words")."""
    from fourierflow_amd._capi import LayerInferDesc
    lib, p = be.lib, be.ptr
    B, M, N, K, C, H = 1, 8, 32, 4, 64, 256
    S = _setup(be, B, M, N, K, seed=77, scale=scale)
    out = be.empty(S["x"].shape)
    d = LayerInferDesc(S["branch"](0, S["mix"][0]), S["branch"](1, S["mix"][1]), 2, 0, p(S["packs"][0]), p(S["db1"]),
                       p(S["packs"][1]), p(S["db2"]), p(S["dx"]), p(out), C, H, None)
    assert lib.ffno_layer_infer(ctypes.byref(d), None) == 0
    ref, _ = layer_fp64(S["x"], S["w"][0], S["w"][1], S["W1"], S["b1"], S["W2"], S["b2"], K)
    got = np.array(be.get(out))
    assert rel_l2(got.astype(np.float64) - S["x"], ref - S["x"]) < 1e-5


@pytest.mark.parametrize("B,M,N,K,spread", [(1, 8, 32, 4, 1.0), (2, 16, 64, 8, 1.0), (2, 16, 64, 8, 1e8), (8, 64, 64, 16, 1e4)])
def test_infer_layer_self_ranged_lines(be, B, M, N, K, spread):
    """FFNO_BRANCH_SELF_RANGE (axis lengths <= 64): the first launch scales every LINE from its own maximum -- no range word is read
    (the descriptor carries none), the second launch records none.  `spread`: the images' magnitudes differ by that factor (one
    tensor-wide power of two would push the small images' samples towards the half format's subnormals: per-line scales do not
    care).  Against fp64, per image; and the tensor-ranged launch on the same operands agrees to fp32 rounding when spread = 1."""
    from fourierflow_amd._capi import BRANCH_SELF_RANGE, LayerInferDesc
    if be.kind == "emu" and B * M * N > 2048:
        pytest.skip("emulator time budget (the GPU run covers all shapes)")
    lib, p = be.lib, be.ptr
    C, H = 64, 256
    S = _setup(be, B, M, N, K, seed=4242 + B + M + N + K)
    x = S["x"]
    if spread != 1.0:      # image b scaled by spread^(b / (B - 1)) (biases stay: the residual x dominates, so compare the UPDATE per image)
        f = np.array([spread ** (b / max(1, B - 1)) for b in range(B)], np.float32)
        x = (x * f[:, None, None, None]).astype(np.float32)
    dx = be.put(x)

    def branch(i, sr):
        br = S["branch"](i, S["mix"][i])
        br.in_ = p(dx)
        if sr:
            br.flags, br.in_amax = BRANCH_SELF_RANGE, None
        else:
            br.in_amax = p(word)
        return br

    word = be.zeros(1, np.uint32)
    assert lib.ffno_amax(p(dx), x.size, p(word), None) == 0
    outs = []
    for sr in (True, False):
        out = be.empty(x.shape)
        d = LayerInferDesc(branch(0, sr), branch(1, sr), 2, 0, p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]),
                           p(dx), p(out), C, H, None)
        assert lib.ffno_layer_infer(ctypes.byref(d), None) == 0
        outs.append(np.array(be.get(out)).astype(np.float64))
    ref, _ = layer_fp64(x, S["w"][0], S["w"][1], S["W1"], S["b1"], S["W2"], S["b2"], K)
    for b in range(B):
        e = rel_l2(outs[0][b] - x[b], ref[b] - x[b])
        assert e < 1e-5, (b, e)
    if spread == 1.0:
        assert rel_l2(outs[0], outs[1]) < 1e-6
    # one flagged branch next to an unflagged one: refused; flagged lines longer than 64 samples: unsupported
    a, b_ = branch(0, True), branch(1, False)
    assert lib.ffno_spectral_x3_mix_pair(ctypes.byref(a), ctypes.byref(b_), C, 2, None) == -1


def test_self_range_refuses_long_lines(be):
    from fourierflow_amd._capi import BRANCH_SELF_RANGE
    lib = be.lib
    S = _setup(be, 1, 32, 128, 16, seed=9)
    a, b = S["branch"](0, S["mix"][0]), S["branch"](1, S["mix"][1])
    a.flags = b.flags = BRANCH_SELF_RANGE
    assert lib.ffno_spectral_x3_mix_pair(ctypes.byref(a), ctypes.byref(b), 64, 2, None) == -2


def test_infer_layer_argument_checks(be):
    from fourierflow_amd._capi import LayerInferDesc
    lib, p = be.lib, be.ptr
    C, H = 64, 256
    assert lib.ffno_layer_infer(None, None) == -1
    assert lib.ffno_layer_infer_supported(2, 16, 48, C, H, 8, 8) == 0        # N not a multiple of 32
    assert lib.ffno_layer_infer_supported(2, 16, 64, C, H, 17, 8) == 0        # > 16 modes
    assert lib.ffno_layer_infer_supported(2, 16, 64, 32, 128, 8, 8) == 0      # width 32
    assert lib.ffno_layer_infer_supported(2, 16, 64, C, H, 8, 10) == 0        # modes > M / 2 + 1
    assert lib.ffno_infer_mix_bytes(C, 17, 4) == 0
    S = _setup(be, 1, 8, 32, 4, seed=5)
    out = be.empty(S["x"].shape)
    # both branches on the same axis / a missing fragment table: refused
    a, b = S["branch"](0, S["mix"][0]), S["branch"](0, S["mix"][1])
    d = LayerInferDesc(a, b, 2, 0, p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), p(S["dx"]), p(out), C, H, None)
    assert lib.ffno_infer_ff(ctypes.byref(a), ctypes.byref(b), p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), None,
                             p(out), C, H, None, None) == -1
    b = S["branch"](1, S["mix"][1])
    b.dft_frags = None
    assert lib.ffno_infer_ff(ctypes.byref(a), ctypes.byref(b), p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), None,
                             p(out), C, H, None, None) == -1
    # bf16x3 packs / a residual in the first launch: refused
    a.planes_format = 0
    assert lib.ffno_spectral_x3_mix_pair(ctypes.byref(a), ctypes.byref(S["branch"](1, S["mix"][1])), C, 2, None) == -2


# ---- through the engine: forward(save_for_backward=False) = predict / validation / rollout ------------------------------------------
def _block(kw, seed, device):
    from fourierflow_amd.modules import FNOFactorized2DBlock
    blk = FNOFactorized2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_block_state_dict(kw, seed).items()}, strict=True)
    return blk.to(device)


def test_engine_forward_without_saving_takes_the_inference_layer(host_device):
    """A small block on both backends: the no-save forward runs ffno_layer_infer for every layer (one C call per layer), agrees with the
    oracle to 1e-5 and with the training-path forward of the same engine to fp32 rounding; a forward that saves for a backward pass
    keeps the training launches."""
    kw = dict(modes=4, width=64, input_dim=3, n_layers=3, share_weight=False, factor=4, ff_weight_norm=True, gain=0.5)
    seed, B, M, N = 9, 2, 8, 32
    blk = _block(kw, seed, host_device)
    eng = blk.engine()
    eng.x3_min_lines = 1
    eng.infer_min_lines = 1
    x_np, t_np = gu.make_block_io(kw, seed, B, M, N)
    x = torch.from_numpy(x_np).to(host_device)
    seen = []
    orig = eng._k
    eng._k = lambda name, fn, *a, _o=orig: (seen.append(name), _o(name, fn, *a))[1]
    with torch.no_grad():
        y_inf = blk(x)["forecast"].cpu().numpy()
    assert eng.infer_last and seen.count("layer_infer") == 3 and "layer_fwd" not in seen and "spectral_fused" not in seen
    assert eng.infer_self_ranged_last      # axis lengths 32 / 8: every line scaled from its own maximum, no range words in this pass
    # ... and with the tensor's range words instead (the round-6 first form; what axis lengths > 64 still run): same result to rounding
    eng.infer_self_range = False
    with torch.no_grad():
        y_rw = blk(x)["forecast"].cpu().numpy()
    assert eng.infer_last and not eng.infer_self_ranged_last and rel_l2(y_rw, y_inf) < 2e-6
    eng.infer_self_range = True
    seen.clear()
    eng.use_infer_layer = False
    with torch.no_grad():
        y_trn = blk(x)["forecast"].cpu().numpy()
    assert not eng.infer_last and "layer_infer" not in seen
    eng.use_infer_layer = True
    seen.clear()
    y_sav = blk(x)["forecast"].detach().cpu().numpy()
    assert not eng.infer_last and "layer_infer" not in seen      # a pass that saves for backward() keeps its four launches
    np.testing.assert_array_equal(y_sav, y_trn)
    ref_out, _, _ = ou.oracle_block_run(kw, seed, B, M, N, io=(x_np, t_np))
    ref = ref_out["forecast"].detach().numpy()
    print(f"[engine infer] vs oracle {rel_l2(y_inf, ref):.2e} (training-path forward {rel_l2(y_trn, ref):.2e}), vs training path {rel_l2(y_inf, y_trn):.2e}")
    assert rel_l2(y_inf, ref) < 1e-5 and rel_l2(y_inf, y_trn) < 2e-6
    # small launches stay on the latency kernels unless forced (the default threshold: more than 4 lines per CU and axis pair)
    eng.infer_min_lines = None
    with torch.no_grad():
        blk(x)
    assert not eng.infer_last


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["c64_4l_markov", "c64_24l_markov", "c64_3l_unshared"])
def test_inference_layer_vs_reference_golden(tag):
    """The reference's own forward outputs (tests/golden/block_*.npz, generated by the imported reference) through the inference
    layer: 64 x 64 / 4 layers, 32 x 32 / 24 layers (markov configuration), 16 x 32 / per-layer weights."""
    g = gu.load_golden("block_" + tag)
    kw = gu.golden_kwargs(g)
    B, M, N, seed = [int(v) for v in g["meta"]]
    blk = _block(kw, seed, "cuda:0")
    eng = blk.engine()
    eng.x3_min_lines = 1
    eng.infer_min_lines = 1
    x_np, _ = gu.make_block_io(kw, seed, B, M, N)
    blk.eval()
    with torch.no_grad():
        pred = blk(torch.from_numpy(x_np).cuda())["forecast"]
    assert eng.infer_last
    e = gu.compare_packed(g, "forecast", pred.cpu().numpy(), 1e-5)
    print(f"[inference layer vs golden {tag}] {e:.2e}")
    assert e < 1e-5


# ---- the whole stack in one persistent launch (ffno_infer_stack) -------------------------------------------------------------------
def _stack_setup(be, B, K, L, seed):
    """L layers over one [B, 64, 64, 64] tensor: shared Fourier packs (markov/24 shares them), per-layer feed-forward packs."""
    S = _setup(be, B, 64, 64, K, seed=seed)
    rs = np.random.RandomState(seed + 1)
    C, H = 64, 256
    layers = []
    for l in range(L):
        W1 = (rs.standard_normal((H, C)) / 8).astype(np.float32)
        W2 = (rs.standard_normal((C, H)) / 16).astype(np.float32)
        b1, b2 = (rs.standard_normal(H) * 0.1).astype(np.float32), (rs.standard_normal(C) * 0.1).astype(np.float32)
        packs, keep = pack_weights_h(be, W1, W2)
        layers.append(dict(W1=W1, W2=W2, b1=b1, b2=b2, packs=packs, keep=keep, db1=be.put(b1), db2=be.put(b2)))
    return S, layers


@pytest.mark.parametrize("B,K,L,mode", [(8, 4, 2, 1), (3, 4, 2, 1), (3, 4, 2, 5), (32, 16, 3, 1), (32, 16, 24, 0), (19, 16, 5, 0), (72, 8, 3, 0),
                                        (12, 16, 4, 0), (12, 16, 4, 4), (16, 8, 3, 0)])
def test_infer_stack_equals_the_layer_loop(be, B, K, L, mode):
    """ffno_infer_stack (the 8 workgroups of an image run both kernels of every layer as phases of one kernel) == the loop of
    ffno_layer_infer calls it replaces, BIT FOR BIT (same bodies, same order), and within 1e-5 of fp64 per layer update."""
    from fourierflow_amd._capi import BRANCH_SELF_RANGE, InferStackDesc, InferStackLayer, LayerInferDesc
    lib, p = be.lib, be.ptr
    C, H = 64, 256
    if be.kind == "emu" and B > 8:
        pytest.skip("emulator time budget (the GPU run covers the full-size stacks and the persistent launch)")
    sup = lib.ffno_infer_stack_supported(B, 64, 64, C, H, K, K, L)
    # (mode: | 1 one launch per phase, | 4 eight workgroups per image whatever the batch -- by default a batch <= 16 gets sixteen)
    if be.kind == "gpu" and not (mode & 1) and sup != 2:
        pytest.skip("the CUs of this device do not come as 8 XCDs of whole groups")
    assert sup >= 1
    S, layers = _stack_setup(be, B, K, L, seed=31 + B + K + L)
    x0 = S["x"]

    def branches(dx):
        out = []
        for i in range(2):
            br = S["branch"](i, S["mix"][i])
            br.in_ = p(dx)
            br.flags, br.in_amax = BRANCH_SELF_RANGE, None
            out.append(br)
        return out

    # reference: the loop of two-launch layers, in place, the last layer without residual into its own buffer
    dx = be.put(x0)
    last_ref = be.empty(x0.shape)
    a, b = branches(dx)
    for l, y in enumerate(layers):
        last = l == L - 1
        d = LayerInferDesc(a, b, 2, 0, p(y["packs"][0]), p(y["db1"]), p(y["packs"][1]), p(y["db2"]), None if last else p(dx),
                           p(last_ref) if last else p(dx), C, H, None)
        assert lib.ffno_layer_infer(ctypes.byref(d), None) == 0
    ref_last, ref_x = np.array(be.get(last_ref)).copy(), np.array(be.get(dx)).copy()
    # the stack
    dx2 = be.put(x0)
    last_out = be.empty(x0.shape)
    a2, b2 = branches(dx2)
    arr = (InferStackLayer * L)(*[InferStackLayer(a2.planes, b2.planes, p(y["packs"][0]), p(y["db1"]), p(y["packs"][1]), p(y["db2"]))
                                  for y in layers])
    sync = be.zeros(int(lib.ffno_infer_stack_sync_words(B)), np.uint32)
    sd = InferStackDesc(a2, b2, ctypes.cast(arr, ctypes.c_void_p), L, C, H, mode, p(last_out), p(sync))
    assert lib.ffno_infer_stack(ctypes.byref(sd), None) == 0
    words = np.array(be.get(sync))
    assert words[-1] == 0, f"error word {words[-1]} (tickets {words[:8]})"
    if be.kind == "gpu" and not (mode & 1):
        assert len(set(words[:8])) == 1 and words[0] % 8 == 0      # every XCD drew exactly its share of tickets (one workgroup per CU)
    if B * 8 >= 256:      # the loop's second kernel runs 8-row workgroups too: the same code on the same tiles, bit for bit
        np.testing.assert_array_equal(np.array(be.get(last_out)), ref_last)
        np.testing.assert_array_equal(np.array(be.get(dx2)), ref_x)
    else:                 # (a small batch: the loop picks 2-row workgroups to fill the chip -- other input scales per workgroup, fp32 rounding)
        assert rel_l2(be.get(last_out), ref_last) < 1e-6 and rel_l2(be.get(dx2), ref_x) < 1e-6
    # ... and the first layer against fp64 (the loop itself is covered above: test_infer_layer_vs_fp64_and_vs_the_training_layer)
    if L == 2:
        y0 = layers[0]
        r1, _ = layer_fp64(x0, S["w"][0], S["w"][1], y0["W1"], y0["b1"], y0["W2"], y0["b2"], K)
        assert rel_l2(ref_x.astype(np.float64) - x0, r1 - x0) < 1e-5


def test_infer_stack_phase_trace(be):
    """mode | 2 (a diagnostic, tools/trace_stack.py): the same result as mode 0, and behind the sync words the 8 workgroups of group 0
    leave ordered device-clock stamps for every phase of their first image (start <= marks <= body done <= barrier passed)."""
    from fourierflow_amd._capi import BRANCH_SELF_RANGE, InferStackDesc, InferStackLayer
    if be.kind != "gpu":
        pytest.skip("device clock")
    lib, p = be.lib, be.ptr
    B, K, L, C, H = 16, 8, 3, 64, 256
    if lib.ffno_infer_stack_supported(B, 64, 64, C, H, K, K, L) != 2:
        pytest.skip("no persistent launch on this device")
    S, layers = _stack_setup(be, B, K, L, seed=77)
    outs = []
    for mode in (0, 2):
        dx = be.put(S["x"])
        last = be.empty(S["x"].shape)
        brs = []
        for i in range(2):
            br = S["branch"](i, S["mix"][i])
            br.in_ = p(dx)
            br.flags, br.in_amax = BRANCH_SELF_RANGE, None
            brs.append(br)
        arr = (InferStackLayer * L)(*[InferStackLayer(brs[0].planes, brs[1].planes, p(y["packs"][0]), p(y["db1"]), p(y["packs"][1]), p(y["db2"]))
                                      for y in layers])
        ns, nt = int(lib.ffno_infer_stack_sync_words(B)), int(lib.ffno_infer_stack_trace_words(L))
        assert nt == 1 + 8 * 2 * L * 12 * 2
        sync = be.zeros(ns + nt, np.uint32)
        sd = InferStackDesc(brs[0], brs[1], ctypes.cast(arr, ctypes.c_void_p), L, C, H, mode, p(last), p(sync))
        assert lib.ffno_infer_stack(ctypes.byref(sd), None) == 0
        w = np.array(be.get(sync))
        assert w[ns - 1] == 0
        outs.append((np.array(be.get(last)).copy(), np.array(be.get(dx)).copy(), w))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert not outs[0][2][ns:].any()      # (mode 0 leaves the trace words alone)
    t = np.ascontiguousarray(outs[1][2][ns + 1:]).view(np.uint64).reshape(8, 2 * L, 12)
    assert (t[:, :, 0] > 0).all()
    for a, b in ((0, 3), (3, 4), (4, 5), (5, 1)):
        assert (t[:, :, a] <= t[:, :, b]).all(), (a, b)
    assert (t[:, :-1, 1] <= t[:, :-1, 2]).all()              # the barrier is passed after the body is done ...
    assert (t[:, :-1, 2] <= t[:, 1:, 0] + 1).all()           # ... and the next phase starts behind it
    assert (t[:, 2 * L - 1, 1] <= t[:, 2 * L - 1, 2]).all()      # (no barrier after the last phase: the stamp follows the body at once)


def test_infer_stack_argument_checks(be):
    lib = be.lib
    assert lib.ffno_infer_stack(None, None) == -1
    assert lib.ffno_infer_stack_supported(8, 64, 32, 64, 256, 4, 4, 2) == 0      # not 64 x 64
    assert lib.ffno_infer_stack_supported(0, 64, 64, 64, 256, 4, 4, 2) == 0      # no images
    assert lib.ffno_infer_stack_supported(3, 64, 64, 64, 256, 4, 4, 2) >= 1      # any batch (groups idle / groups walk several images)
    assert lib.ffno_infer_stack_supported(8, 64, 64, 64, 256, 4, 4, 33) == 0     # more than 32 layers
    assert lib.ffno_infer_stack_sync_words(32) == lib.ffno_infer_stack_sync_words(3) == 8 + 64 + 1


@pytest.mark.gpu
def test_engine_forward_takes_the_persistent_stack_at_the_bench_geometry():
    """markov/24 at batch 32 (64 x 64, 16 modes: the BASELINE geometry) through `forward(save_for_backward=False)`: ONE persistent
    launch for the 24 layers (engine.infer_stack_last), bit-identical to the same pass on the per-layer launches; batch 19 (the
    reference's own batch size: 19 of the 32 groups busy) and batch 40 (groups that walk two images) take it too and agree with
    the per-layer launches to fp32 rounding (the loop picks smaller row tiles for a batch that does not fill the chip)."""
    import torch
    from fourierflow_amd import _lib
    kw = dict(modes=16, width=64, input_dim=3, n_layers=24, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)
    blk = _block(kw, 5, "cuda:0")
    eng = blk.engine()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus != 256:
        pytest.skip("not a 256-CU device")
    x = torch.randn(32, 64, 64, 3, device="cuda:0")
    seen = []
    orig = eng._k
    eng._k = lambda name, fn, *a, _o=orig: (seen.append(name), _o(name, fn, *a))[1]
    with torch.no_grad():
        y_stack = blk(x)["forecast"].clone()
        assert eng.infer_last and eng.infer_self_ranged_last and eng.infer_stack_last
        assert "layer_infer" not in seen and "spectral_mix" not in seen
        y_again = blk(x)["forecast"].clone()       # (second call: the error word of the first is looked at without synchronising)
        eng.use_infer_stack = False
        seen.clear()
        y_loop = blk(x)["forecast"].clone()
        assert not eng.infer_stack_last and seen.count("layer_infer") == 24
        eng.use_infer_stack = True
        others = {}
        for Bo in (19, 40):
            xo = torch.randn(Bo, 64, 64, 3, device="cuda:0")
            if Bo == 40:      # (a last round with 8 of 32 groups busy: the engine's rule prefers the per-layer launches there)
                assert not eng._stack_pays(Bo)
                eng.infer_stack_any_batch = True
            ys = blk(xo)["forecast"].clone()
            assert eng.infer_last and eng.infer_stack_last, Bo
            assert int(eng._ws.stack_sync[-1].item()) == 0
            eng.use_infer_stack = False
            yl = blk(xo)["forecast"].clone()
            assert eng.infer_last and not eng.infer_stack_last
            eng.use_infer_stack = True
            others[Bo] = float((ys - yl).norm() / yl.norm())
    assert torch.equal(y_stack, y_loop) and torch.equal(y_again, y_loop)
    assert all(e < 1e-6 for e in others.values()), others
