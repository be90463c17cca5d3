"""Range safety of the fp16x2 default (VERDICT r02 weak #1 / ADVICE r02): the split-fp16 kernels used to take their operands
unscaled -- anything at or above 65504 became inf -- and the backward pass trusted ONE power of two taken from the loss
gradient.  Now every fp16x2 kernel scales what it splits from the RANGE WORD its producer recorded (include/ffno.h "Range
words"): per launch, on the device.  These tests drive the whole block through the engine -- on the CPU wave emulator
(-m "not gpu") and on the MI355X (-m gpu) -- in the regimes that used to overflow, against the oracle at the north-star bars
(forward 1e-5, gradients 5e-5 on the same ReLU active sets), and check that nothing is inf / nan."""
import numpy as np
import pytest
import torch

import golden_util as gu
import oracle_util as ou
from backend_util import host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc


def _build(kw, sd_np, device):
    from fourierflow_amd.modules import FNOFactorized2DBlock
    blk = FNOFactorized2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    return blk.to(device)


def _run_and_compare(label, kw, sd_np, x_np, t_np, device, fwd_tol=1e-5):
    blk = _build(kw, sd_np, device)
    eng = blk.engine()
    eng.use_x3, eng.x3_min_lines = True, 1          # the fp16x2 channel mix also on the small test grids
    assert eng.ff_split == "fp16x2" and eng.x3_mix_split == "fp16x2", "these tests are about the DEFAULT arithmetic"
    pred = blk(torch.from_numpy(x_np).to(device))["forecast"]
    loss = orc.lp_rel_loss(pred, torch.from_numpy(t_np).to(device))
    loss.backward()
    got = pred.detach().cpu().numpy()
    assert np.all(np.isfinite(got)), f"{label}: non-finite forward"
    B, M, N = x_np.shape[:3]
    masks = ou.engine_relu_masks(eng)
    ref_out, ref_loss, _ = ou.oracle_block_run(kw, 0, B, M, N, io=(x_np, t_np), sd_np=sd_np)
    err = rel_l2(got, ref_out["forecast"].detach().numpy())
    print(f"[{label}] forward rel-L2 vs oracle {err:.2e}")
    assert err < fwd_tol, (label, err)
    named = dict(blk.named_parameters())
    grads = {n: named[n].grad.cpu().numpy() for n in eng.param_names}
    for n, g in grads.items():
        assert np.all(np.isfinite(g)), f"{label}: non-finite gradient {n}"
    ou.check_grads_at_rounding_level(label, grads, lambda dt: ou.oracle_block_run(kw, 0, B, M, N, dtype=dt, relu_masks=masks,
                                                                                  io=(x_np, t_np), sd_np=sd_np)[2])
    return eng


KW = dict(modes=4, width=64, input_dim=3, n_layers=2, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)


@pytest.mark.parametrize("scale", [1e5, 1e6, 1e-7])
def test_block_with_unnormalised_inputs(host_device, scale):
    """Activations of 1e5 / 1e6 (should_normalize: false, physical units) and of 1e-7: the lifted features, the spectra (x sqrt(L))
    and the hidden layer all leave [2^-14, 65504]; the fp16x2 path must still meet the bars of the O(1) case."""
    seed, B, M, N = 3, 1, 8, 8
    sd_np = gu.make_block_state_dict(KW, seed)
    x_np, t_np = gu.make_block_io(KW, seed, B, M, N)
    x_np, t_np = (x_np * scale).astype(np.float32), (t_np * scale).astype(np.float32)
    eng = _run_and_compare(f"inputs x {scale:g} {host_device}", KW, sd_np, x_np, t_np, host_device)
    ws = eng._workspace(B, (M, N), True)
    words = ws.RW.cpu().numpy().view(np.float32).reshape(len(ws.rw_kinds), -1)
    assert words[ws.rw_kinds["x"], 0] > 0 and words[ws.rw_kinds["s"], 0] > 0, "the producers recorded their maxima"
    if scale >= 1e5:
        assert words[ws.rw_kinds["x"]].max() >= 65504, "the test must exercise activations beyond the half format"


def test_block_with_gradient_growth_through_the_layers(host_device):
    """A backward pass whose gradient grows > 1e4-fold from the head to the lift (large feed-forward gains and Fourier
    weights: every residual layer multiplies the gradient).  One scale for the whole pass -- what round 2 did -- overflows
    here; a scale per launch from the producer's range word cannot."""
    kw = dict(KW, n_layers=6)
    seed, B, M, N = 5, 1, 8, 8
    sd_np = gu.make_block_state_dict(kw, seed)
    for k in sd_np:
        if "fourier_weight." in k:          # (shared weights appear under the block AND under every layer)
            sd_np[k] = (sd_np[k] * 60.0).astype(np.float32)
        elif "backcast_ff" in k and k.endswith("weight_g"):
            sd_np[k] = (sd_np[k] * 5.0).astype(np.float32)
    x_np, t_np = gu.make_block_io(kw, seed, B, M, N)
    eng = _run_and_compare(f"gradient growth {host_device}", kw, sd_np, x_np, t_np, host_device)
    ws = eng._workspace(B, (M, N), True)
    words = ws.RW.cpu().numpy().view(np.float32).reshape(len(ws.rw_kinds), -1)
    g = words[ws.rw_kinds["g"]]
    growth = g[eng.L] / g[eng.L - 1]           # word L: the gradient handed to the lift; word L - 1: the head's
    print(f"gradient maxima per layer (head -> lift): {g[:eng.L][::-1]} -> {g[eng.L]}; growth {growth:.3g}")
    assert growth > 1e4, growth


def test_weights_beyond_the_half_range_are_reported(host_device):
    """The one operand the range words do not scale: a weight >= 65504 overflows its fp16 planes.  The engine folds max |W| of
    what it packs as fp16x2 on the device and looks at it one rebuild later (no synchronisation): the run stops with a message
    that names the any-range arithmetic; with that arithmetic the same weights are fine."""
    seed, B, M, N = 3, 1, 8, 8
    sd_np = gu.make_block_state_dict(KW, seed)
    key = next(k for k in sd_np if k.endswith("backcast_ff.layers.0.0.weight_g"))
    sd_np[key] = (sd_np[key] * 0 + 3.0e5).astype(np.float32)          # effective rows of norm 3e5
    x_np, _ = gu.make_block_io(KW, seed, B, M, N)
    x = torch.from_numpy(x_np).to(host_device)
    blk = _build(KW, sd_np, host_device)
    eng = blk.engine()
    eng.use_x3, eng.x3_min_lines, eng.weight_range_check_every = True, 1, 1
    with torch.no_grad():
        blk(x)                                                         # folds the maximum, asynchronously
        if host_device != "cpu":
            torch.cuda.synchronize()
        eng.weights_changed()                                          # (a training step would have changed them)
        with pytest.raises(FloatingPointError, match="bf16x3"):
            blk(x)
    blk = _build(KW, sd_np, host_device)
    eng = blk.engine()
    eng.use_x3, eng.x3_min_lines, eng.weight_range_check_every = True, 1, 1
    eng.ff_split = eng.x3_mix_split = "bf16x3"
    with torch.no_grad():
        out = blk(x)["forecast"]
        eng.weights_changed()
        out = blk(x)["forecast"]
    assert torch.isfinite(out).all()
