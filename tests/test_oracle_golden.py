"""Pin the CPU oracle against golden vectors produced by the imported reference
(tools/make_golden.py).  Everything here runs on CPU."""
import numpy as np
import pytest
import torch

import golden_util as gu
import oracle_util as ou
from oracle import ffno_oracle as orc

TOL = 2e-5  # fp32 vs fp32 with identical op order: observed ~1e-7; gradient sums re-associate


@pytest.mark.parametrize("tag", ["tiny_full", "tiny_lowpass", "c64_rect", "c64_k16", "c32_odd"])
def test_forward_fourier_matches_reference(tag):
    g = gu.load_golden("spectral_" + tag)
    B, M, N, C, K, seed = [int(v) for v in g["meta"]]
    mode = str(g["mode"])
    x, w0, w1, gy = gu.make_spectral_io(seed, B, M, N, C, K)
    xt = torch.tensor(x, requires_grad=True)
    w0t, w1t = torch.tensor(w0, requires_grad=True), torch.tensor(w1, requires_grad=True)
    y = orc.forward_fourier(xt, w0t, w1t, K, mode)
    y.backward(torch.tensor(gy))
    assert gu.compare_packed(g, "y", y.detach().numpy(), TOL) < TOL
    assert gu.compare_packed(g, "gx", xt.grad.numpy(), TOL) < TOL
    if mode == "full":
        assert gu.compare_packed(g, "gw0", w0t.grad.numpy(), TOL) < TOL
        assert gu.compare_packed(g, "gw1", w1t.grad.numpy(), TOL) < TOL


BLOCKS = ["c64_2l_shared", "c64_3l_unshared", "c64_4l_markov", "c64_24l_markov", "c64_fork",
          "c64_sharefork", "c64_sharefork_fork", "c64_lowpass", "c64_nofourier", "c32_nown",
          "c64_layernorm", "c64_ff3"]


@pytest.mark.parametrize("tag", BLOCKS)
def test_block_matches_reference(tag):
    g = gu.load_golden("block_" + tag)
    kw = gu.golden_kwargs(g)
    B, M, N, seed = [int(v) for v in g["meta"]]
    out, loss, grads = ou.oracle_block_run(kw, seed, B, M, N)
    assert gu.compare_packed(g, "forecast", out["forecast"].detach().numpy(), TOL) < TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, abs(float(g["loss"])))
    if "forecast_list" in gu.packed_names(g):
        fl = np.stack([f.detach().numpy() for f in out["forecast_list"]])
        assert gu.compare_packed(g, "forecast_list", fl, TOL) < TOL
    names = [n for n in gu.packed_names(g) if n.startswith("grad.")]
    assert names
    for n in names:
        got = grads[n[len("grad."):]]
        assert got is not None, n
        err = gu.compare_packed(g, n, got, TOL)
        assert err < 5e-5, (n, err)
    # every unique parameter of the reference is accounted for
    assert {n[len("grad."):] for n in names} == {k for k, v in grads.items() if v is not None}


def test_misc_lploss_wnlinear_cosine():
    g = gu.load_golden("misc")
    rel = orc.lp_rel_loss(torch.tensor(g["lp_a"]), torch.tensor(g["lp_b"])).item()
    assert abs(rel - float(g["lp_rel"])) < 1e-6
    sd = {k[len("wn_sd."):]: torch.tensor(g[k], requires_grad=True) for k in g.files if k.startswith("wn_sd.")}
    y = orc.linear_from_sd(sd, "", torch.tensor(g["wn_x"]))
    y.sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g["wn_y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sd["weight_g"].grad.numpy(), g["wn_gg"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["weight_v"].grad.numpy(), g["wn_gv"], rtol=1e-4, atol=1e-6)
    c = gu.load_golden("cosine")
    for s, f in zip(c["steps"], c["factor"]):
        assert abs(orc.cosine_warmup_factor(int(s), 500, 100000, 0.5) - float(f)) < 1e-12


def test_zongyi_baseline_config0_matches_reference():
    """BASELINE config 0 (torus_li/zongyi/4_layers, CPU plumbing case: 64x64, batch 2, modes 12, width 20)."""
    g = gu.load_golden("zongyi_4l")
    kw = gu.golden_kwargs(g)
    sd_np, x = gu.make_zongyi_state_dict(kw, int(g["seed"]))
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in sd_np.items()}
    out = orc.fno_zongyi_2d(sd, torch.tensor(x), modes=kw["modes1"], n_layers=kw["n_layers"])["forecast"]
    loss = (out ** 2).mean()
    loss.backward()
    assert tuple(out.shape) == (2, 64, 64, 1)
    assert gu.compare_packed(g, "forecast", out.detach().numpy(), TOL) < TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    for n in [k for k in gu.packed_names(g) if k.startswith("grad.")]:
        assert gu.compare_packed(g, n, sd[n[5:]].grad.numpy(), TOL) < 5e-5, n


def test_zongyi_markov_residual_flags_match_reference():
    """residual=True, conv_residual=False (torus_li/ablation/zongyi_markov_residual; grid_2d.py:74-77,126)."""
    g = gu.load_golden("zongyi_markov_residual")
    kw = gu.golden_kwargs(g)
    sd_np, x = gu.make_zongyi_state_dict(kw, int(g["seed"]), grid=int(g["grid"]))
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in sd_np.items()}
    out = orc.fno_zongyi_2d(sd, torch.tensor(x), modes=kw["modes1"], n_layers=kw["n_layers"], residual=kw["residual"],
                            conv_residual=kw["conv_residual"])["forecast"]
    loss = (out ** 2).mean()
    loss.backward()
    assert gu.compare_packed(g, "forecast", out.detach().numpy(), TOL) < TOL
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    for n in [k for k in gu.packed_names(g) if k.startswith("grad.")]:
        assert gu.compare_packed(g, n, sd[n[5:]].grad.numpy(), TOL) < 5e-5, n
