"""the training-step glue around the operator -- feature build with positional channels,
running normaliser, noise, inverse-normalise fused into the loss, epoch-0 accumulation, rollout."""
import ctypes

import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import be, host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc


def test_oracle_normalizer_matches_reference_golden():
    g = gu.load_golden("normalizer")
    nz = orc.NormalizerState(3, max_accumulations=2)
    for i in range(3):
        out = nz.forward(torch.tensor(g["xs"][i]), training=True)
        np.testing.assert_allclose(out.numpy(), g["outs"][i], rtol=2e-5, atol=2e-6)
    assert float(nz.count) == float(g["sd.count"]) and float(nz.n_accumulations) == float(g["sd.n_accumulations"])
    np.testing.assert_allclose(nz.sum.numpy(), g["sd.sum"], rtol=1e-5)
    np.testing.assert_allclose(nz.sum_squared.numpy(), g["sd.sum_squared"], rtol=1e-5)
    inv = nz.inverse(torch.tensor(g["outs"][2][..., 0:1]), 0)
    np.testing.assert_allclose(inv.numpy(), g["inv"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("B,M,N,Cx", [(2, 6, 5, 1), (3, 16, 16, 3)])
def test_markov_features_kernel(be, B, M, N, Cx):
    """ffno_markov_features vs the oracle: accumulate twice, then frozen statistics; noise; no-normalise path."""
    lib, p = be.lib, be.ptr
    D = Cx + 2
    rs = np.random.RandomState(B + M + Cx)
    nz = orc.NormalizerState(D, max_accumulations=2)
    state, derived, partial = be.zeros(2 * D + 2), be.zeros(2 * D), be.zeros(256 * 32)
    for it in range(3):
        x = (rs.standard_normal((B, M, N, Cx)) * 2 + 0.7).astype(np.float32)
        noise = rs.standard_normal((B, M, N, D)).astype(np.float32)
        acc = int(it < 2)
        dx, dn, out = be.put(x), be.put(noise), be.empty((B, M, N, D))
        assert lib.ffno_markov_features(p(dx), p(state), p(derived), p(dn), p(out), p(partial), B, M, N, Cx, 0.0, 1.0,
                                        0.01, 1e-8, acc, 1, None, None) == 0
        ref = orc.markov_features(torch.tensor(x), nz, torch.tensor(noise), 0.01, training=True)
        assert rel_l2(be.get(out), ref.numpy()) < 1e-5
    st = be.get(state)
    np.testing.assert_allclose(st[:D], nz.sum.numpy(), rtol=1e-5)
    np.testing.assert_allclose(st[D:2 * D], nz.sum_squared.numpy(), rtol=1e-5)
    assert st[2 * D] == float(nz.count) and st[2 * D + 1] == 2.0
    out = be.empty((B, M, N, D))
    assert lib.ffno_markov_features(p(dx), p(state), p(derived), None, p(out), p(partial), B, M, N, Cx, -1.0, 1.0,
                                    0.0, 1e-8, 0, 0, None, None) == 0
    ref = orc.markov_features(torch.tensor(x), None, None, 0.0, low=-1.0, high=1.0)
    assert rel_l2(be.get(out), ref.numpy()) < 1e-6


def test_lploss_with_inverse_normalisation(be):
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(2)
    B, n = 3, 300
    pred, tgt = rs.standard_normal((B, n)).astype(np.float32), rs.standard_normal((B, n)).astype(np.float32)
    aff = np.array([1.7, -0.4], np.float32)
    dp, dt, da = be.put(pred), be.put(tgt), be.put(aff)
    loss, gp, tmp = be.zeros(1), be.zeros((B, n)), be.zeros(2 * B)
    assert lib.ffno_lploss_fwd_bwd(p(dp), p(dt), p(loss), p(gp), p(tmp), B, n, 1.0, p(da), None) == 0
    pt = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    l = orc.lp_rel_loss(pt * 1.7 - 0.4, torch.tensor(tgt, dtype=torch.float64))
    l.backward()
    assert abs(be.get(loss)[0] - l.item()) < 1e-6
    assert rel_l2(be.get(gp), pt.grad.numpy()) < 1e-5


def test_markov_routine_matches_oracle_composition(host_device):
    """Epoch 0 accumulates only; then train steps = oracle(features -> block -> inverse -> rel-L2) + AdamW."""
    import oracle_util as ou
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    kw = dict(modes=4, width=64, input_dim=3, n_layers=2, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)
    seed, B, M, N = 71, 2, 16, 16
    sd_np = gu.make_block_state_dict(kw, seed)
    blk = FNOFactorized2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    blk = blk.to(host_device)
    exp = Grid2DMarkovExperiment(blk, n_steps=3, max_accumulations=2, noise_std=0.01,
                                 scheduler=dict(num_warmup_steps=1, num_training_steps=10)).to(host_device)
    rs = np.random.RandomState(seed)
    nz = orc.NormalizerState(3, max_accumulations=2)
    batches = [dict(x=(rs.standard_normal((B, M, N, 1)) * 3 + 1).astype(np.float32),
                    y=(rs.standard_normal((B, M, N, 1)) * 3 + 1).astype(np.float32),
                    noise=rs.standard_normal((B, M, N, 3)).astype(np.float32)) for _ in range(4)]
    dev = lambda a: torch.from_numpy(a).to(host_device)  # noqa: E731
    # epoch 0: statistics only, no optimisation
    for b in batches[:2]:
        assert exp.training_step(dict(x=dev(b["x"]), y=dev(b["y"])), epoch=0, noise=dev(b["noise"])) is None
        orc.markov_features(torch.tensor(b["x"]), nz, torch.tensor(b["noise"]), 0.01)
    assert exp.trainer().step_count == 0
    np.testing.assert_allclose(exp.normalizer.sum.cpu().numpy(), nz.sum.numpy(), rtol=1e-5)
    assert float(exp.normalizer.n_accumulations.item()) == 2.0
    # epoch 1: one train step, compared with autograd through the oracle + torch AdamW
    sd, uniq = ou.torch_state_dict(sd_np)
    opt = torch.optim.AdamW(list(uniq.values()), lr=2.5e-3, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: orc.cosine_warmup_factor(s, 1, 10, 0.5))
    for b in batches[2:]:
        loss = exp.training_step(dict(x=dev(b["x"]), y=dev(b["y"])), epoch=1, noise=dev(b["noise"]))
        feats = orc.markov_features(torch.tensor(b["x"]), nz, torch.tensor(b["noise"]), 0.01)   # frozen stats now
        opt.zero_grad()
        im = orc.ffno2d_block(sd, feats, modes=4, n_layers=2)["forecast"]
        ref_loss = orc.lp_rel_loss(nz.inverse(im, 0), torch.tensor(b["y"]))
        ref_loss.backward()
        opt.step()
        sch.step()
        assert abs(loss.item() - ref_loss.item()) < 2e-5
    named = dict(blk.named_parameters())
    errs = [rel_l2(named[n].detach().cpu().numpy(), p_.detach().numpy()) for n, p_ in uniq.items()]
    assert max(errs) < 5e-4, max(errs)
    # autoregressive rollout (validation path: no noise, frozen statistics)
    x0 = dev(batches[0]["x"])
    roll = exp.rollout(x0, 3)
    assert tuple(roll.shape) == (B, M, N, 3)
    with torch.no_grad():
        xr, ref = torch.tensor(batches[0]["x"]), []
        sdd = {k: v.detach() for k, v in sd.items()}
        for _ in range(3):
            f = orc.markov_features(xr, nz, None, 0.0, training=False)
            xr = nz.inverse(orc.ffno2d_block(sdd, f, modes=4, n_layers=2)["forecast"], 0)
            ref.append(xr)
    assert rel_l2(roll.cpu().numpy(), torch.cat(ref, -1).numpy()) < 1e-4


def test_checkpoint_roundtrip_and_resume(host_device, tmp_path):
    """Lightning-layout checkpoint (routines/base.py:79-102): weights + normaliser + flat AdamW state survive a
    save / load, training resumes bit-for-bit, and the reference loader's REMOVE_KEYS handling is honoured."""
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    kw = dict(modes=4, width=32, n_layers=2, input_dim=3, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)

    def make():
        torch.manual_seed(0)
        return Grid2DMarkovExperiment(FNOFactorized2DBlock(**kw), n_steps=2, noise_std=0.0,
                                      scheduler=dict(num_warmup_steps=2, num_training_steps=50)).to(host_device)

    g = torch.Generator().manual_seed(3)
    batches = [dict(x=torch.randn(2, 8, 8, 1, generator=g).to(host_device), y=torch.randn(2, 8, 8, 1, generator=g).to(host_device))
               for _ in range(5)]
    a = make()
    a.training_step(batches[0], epoch=0)
    for b in batches[1:3]:
        a.training_step(b, epoch=1)
    path = str(tmp_path / "last.ckpt")
    a.save_checkpoint(path, epoch=1)
    ck = torch.load(path, weights_only=False)
    assert {"state_dict", "epoch", "global_step", "optimizer_states", "lr_schedulers"} <= set(ck)
    assert ck["global_step"] == 2 and "normalizer.sum" in ck["state_dict"] and "conv.fourier_weight.0" in ck["state_dict"]
    la = [a.training_step(b, epoch=1).item() for b in batches[3:]]

    b_ = make()
    b_.resume_from_checkpoint(path)
    assert b_.trainer().step_count == 2
    lb = [b_.training_step(b, epoch=1).item() for b in batches[3:]]
    assert la == lb                                   # same kernels, same state: identical continuation
    for (k, va), (_, vb) in zip(a.state_dict().items(), b_.state_dict().items()):
        assert torch.equal(va.cpu(), vb.cpu()), k

    # a reference-style checkpoint carrying the jax-cfd buffers of the velocity path: dropped, loaded non-strict
    ck["state_dict"]["kx"] = torch.zeros(3)
    ck["optimizer_states"] = [{"state": {}, "param_groups": []}]
    torch.save(ck, path)
    c = make()
    c.resume_from_checkpoint(path)
    assert c.trainer().step_count == 2
    assert torch.equal(c.state_dict()["conv.in_proj.weight_v"].cpu(), ck["state_dict"]["conv.in_proj.weight_v"])


@pytest.mark.parametrize("wnorm", [True, False])
def test_resume_converts_a_real_torch_adamw_state_dict(host_device, tmp_path, wnorm):
    """a checkpoint written by the reference carries torch.optim.AdamW.state_dict(), which indexes parameters by
    their position in ``parameters()``.  (1) The mirror's linears register their parameters in the order torch itself produces
    for the reference's construction (nn.Linear, then weight_norm: linear.py:41-52) -- with and without weight-norm; (2) a real
    AdamW state over that order lands on the right slices of the flat moment buffers."""
    from fourierflow_amd.modules import FNOFactorized2DBlock, WNLinear
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    ref_lin = torch.nn.Linear(5, 7)
    if wnorm:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref_lin = torch.nn.utils.weight_norm(ref_lin)
    assert [n for n, _ in WNLinear(5, 7, wnorm=wnorm).named_parameters()] == [n for n, _ in ref_lin.named_parameters()]

    kw = dict(modes=4, width=32, n_layers=2, input_dim=3, share_weight=False, factor=4, ff_weight_norm=wnorm, gain=0.1)
    torch.manual_seed(0)
    exp = Grid2DMarkovExperiment(FNOFactorized2DBlock(**kw), n_steps=2, noise_std=0.0,
                                 scheduler=dict(num_warmup_steps=2, num_training_steps=50)).to(host_device)
    path = str(tmp_path / "ref.ckpt")
    exp.save_checkpoint(path, epoch=1)
    ck = torch.load(path, weights_only=False)
    params = [p for _, p in exp.named_parameters()]
    opt = torch.optim.AdamW([torch.nn.Parameter(p.detach().cpu().clone()) for p in params], lr=1e-3)
    for i, p in enumerate(opt.param_groups[0]["params"]):       # moments that encode the parameter index
        opt.state[p] = dict(step=torch.tensor(7.0), exp_avg=torch.full_like(p, float(i + 1)),
                            exp_avg_sq=torch.full_like(p, 100.0 + i))
    ck["optimizer_states"] = [opt.state_dict()]
    ck["lr_schedulers"] = [dict(last_epoch=7)]
    torch.save(ck, path)
    exp.resume_from_checkpoint(path)
    tr = exp.trainer()
    assert tr.opt_step == 7 and tr.step_count == 7
    eng = tr.engine
    for i, (name, p) in enumerate(exp.named_parameters()):
        short = name[len("conv."):]
        o, cnt = eng._offsets[short], p.numel()
        assert torch.all(tr.m[o:o + cnt] == float(i + 1)) and torch.all(tr.v[o:o + cnt] == 100.0 + i), name


def test_markov_routine_with_velocity_features(host_device):
    """`use_velocity: true` (torus_kochkov, torus_li/ablation/with_velocity; grid_2d_markov.py:130-144): the conv sees
    normalised (vorticity, u, v, x, y); first train step's loss equals the oracle's on the same statistics."""
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    import oracle_util as ou
    import golden_util as gu
    kw = dict(modes=4, width=32, n_layers=2, input_dim=5, share_weight=True, factor=4, ff_weight_norm=True, gain=0.5)
    sd_np = gu.make_block_state_dict(kw, 77)
    blk = FNOFactorized2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    exp = Grid2DMarkovExperiment(blk, n_steps=2, use_velocity=True, grid_size=[16], max_accumulations=1000,
                                 scheduler=dict(num_warmup_steps=2, num_training_steps=50)).to(host_device)
    assert {"kx_16", "ky_16", "lap_16"} <= set(exp.state_dict()) and exp.state_dict()["lap_16"].dtype == torch.complex64
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(2, 16, 16, 1, generator=g), torch.randn(2, 16, 16, 1, generator=g)
    batch = dict(x=x.to(host_device), y=y.to(host_device))
    exp.training_step(batch, epoch=0)
    loss = exp.training_step(batch, epoch=1).item()
    # oracle: features -> normalise with the accumulated statistics (2 accumulations of the same batch) -> block -> loss
    st = orc.NormalizerState(5, max_accumulations=1000)
    vel = orc.velocity_features(x)
    orc.markov_features(vel, st, None, 0.0, training=True)               # epoch 0: statistics only
    feats = orc.markov_features(vel, st, None, 0.0, training=True)       # train step: accumulate again, then normalise
    sd, _ = ou.torch_state_dict(sd_np, requires_grad=False)
    pred = orc.ffno2d_block(sd, feats, modes=4, n_layers=2)["forecast"]
    ref = orc.lp_rel_loss(st.inverse(pred, 0), y)
    assert abs(loss - ref.item()) < 2e-5
    roll = exp.rollout(batch["x"], 2)
    assert tuple(roll.shape) == (2, 16, 16, 2) and bool(torch.isfinite(roll).all())


@pytest.mark.parametrize("use_position,append_force,append_mu", [(True, True, False), (True, True, True), (False, False, True),
                                                                 (False, True, False)])
def test_markov_feature_variants(be, use_position, append_force, append_mu):
    """append_force / append_mu / use_position=False (grid_2d_markov.py:146-162; the torus_vis and ablation configs):
    channel order x | position | force | mu, normalised over all of them."""
    from fourierflow_amd._capi import MarkovExtra
    import ctypes
    lib, p = be.lib, be.ptr
    B, M, N, Cx = 2, 6, 5, 1
    D = Cx + 2 * use_position + append_force + append_mu
    rs = np.random.RandomState(D)
    x = rs.standard_normal((B, M, N, Cx)).astype(np.float32)
    f = rs.standard_normal((B, M, N)).astype(np.float32) if append_force else None
    mu = rs.uniform(1e-5, 1e-3, B).astype(np.float32) if append_mu else None
    dx, df, dmu, out = be.put(x), be.put(f), be.put(mu), be.empty((B, M, N, D))
    state, derived, partial = be.zeros(2 * D + 2), be.zeros(2 * D), be.zeros(256 * 32)
    extra = MarkovExtra(p(df), p(dmu), int(use_position), 0)
    assert lib.ffno_markov_features(p(dx), p(state), p(derived), None, p(out), p(partial), B, M, N, Cx, 0.0, 1.0, 0.0, 1e-8,
                                    1, 1, ctypes.byref(extra), None) == 0
    nz = orc.NormalizerState(D)
    ref = orc.markov_features(torch.tensor(x), nz, None, 0.0, use_position=use_position,
                              force=None if f is None else torch.tensor(f), mu=None if mu is None else torch.tensor(mu))
    assert tuple(ref.shape) == (B, M, N, D)
    assert rel_l2(be.get(out), ref.numpy()) < 1e-5


def test_markov_routine_with_force_and_viscosity(host_device):
    """torus_vis_force style routine: conv.input_dim = 1 + 2 + 1 + 1; loss of the first step vs the oracle composition."""
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    import oracle_util as ou
    kw = dict(modes=4, width=32, n_layers=2, input_dim=5, share_weight=True, factor=4, ff_weight_norm=True, gain=0.5)
    sd_np = gu.make_block_state_dict(kw, 78)
    blk = FNOFactorized2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    exp = Grid2DMarkovExperiment(blk, n_steps=2, append_force=True, append_mu=True, max_accumulations=1000,
                                 scheduler=dict(num_warmup_steps=2, num_training_steps=50)).to(host_device)
    g = torch.Generator().manual_seed(6)
    x, y = torch.randn(2, 8, 8, 1, generator=g), torch.randn(2, 8, 8, 1, generator=g)
    f, mu = torch.randn(2, 8, 8, generator=g), torch.rand(2, generator=g) * 1e-3
    batch = dict(x=x.to(host_device), y=y.to(host_device), f=f.to(host_device), mu=mu.to(host_device))
    exp.training_step(batch, epoch=0)
    loss = exp.training_step(batch, epoch=1).item()
    st = orc.NormalizerState(5, max_accumulations=1000)
    orc.markov_features(x, st, None, 0.0, training=True, force=f, mu=mu)
    feats = orc.markov_features(x, st, None, 0.0, training=True, force=f, mu=mu)
    sd, _ = ou.torch_state_dict(sd_np, requires_grad=False)
    pred = orc.ffno2d_block(sd, feats, modes=4, n_layers=2)["forecast"]
    ref = orc.lp_rel_loss(st.inverse(pred, 0), y)
    assert abs(loss - ref.item()) < 2e-5


def test_markov_routine_with_shuffled_grid(host_device):
    """torus_li/ablation/shuffle_xy_grid (`shuffle_grid: true`, grid_2d_markov.py:75-80,177-183): fixed random row / column
    permutations around the model; one train step and a rollout against the oracle with the same permutations."""
    import oracle_util as ou
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    kw = dict(modes=3, width=32, input_dim=3, n_layers=2, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)
    seed, B, G = 81, 2, 8
    sd_np = gu.make_block_state_dict(kw, seed)
    blk = FNOFactorized2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    torch.manual_seed(3)
    exp = Grid2DMarkovExperiment(blk, n_steps=2, should_normalize=False, shuffle_grid=True, grid_size=[G],
                                 scheduler=dict(num_warmup_steps=1, num_training_steps=10)).to(host_device)
    assert "_x_idx" not in exp.state_dict()                                # plain attributes in the reference
    xi, yi = exp._x_idx.cpu(), exp._y_idx.cpu()
    xv, yv = torch.argsort(xi), torch.argsort(yi)
    assert sorted(xi.tolist()) == list(range(G)) and xi.tolist() != list(range(G))
    rs = np.random.RandomState(seed)
    x, y = (rs.standard_normal((B, G, G, 1)).astype(np.float32) for _ in range(2))
    sd, uniq = ou.torch_state_dict(sd_np)

    def model(feats):                                                      # :177-183
        im = orc.ffno2d_block(sd, feats[:, xi][:, :, yi], modes=3, n_layers=2)["forecast"]
        return im[:, :, yv][:, xv]

    ref_loss = orc.lp_rel_loss(model(orc.markov_features(torch.tensor(x), None, None, 0.0)), torch.tensor(y))
    ref_loss.backward()
    dev = lambda a: torch.from_numpy(a).to(host_device)  # noqa: E731
    loss = exp.training_step(dict(x=dev(x), y=dev(y)), epoch=0)
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    eng = exp.trainer().engine
    for n, p_ in uniq.items():
        assert rel_l2(eng.grad_view(n).cpu().numpy(), p_.grad.numpy()) < 3e-3, n
    with torch.no_grad():                                                  # weights moved by one AdamW step: compare rollouts
        sd2 = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
        xr, ref = torch.tensor(x), []
        for _ in range(2):
            f = orc.markov_features(xr, None, None, 0.0)
            xr = orc.ffno2d_block(sd2, f[:, xi][:, :, yi], modes=3, n_layers=2)["forecast"][:, :, yv][:, xv]
            ref.append(xr)
    roll = exp.rollout(dev(x), 2)
    assert rel_l2(roll.cpu().numpy(), torch.cat(ref, -1).numpy()) < 1e-4


def test_fourier_encode_and_lploss_mirrors(host_device):
    """modules.position.fourier_encode (position.py:6-31) against its closed form, and modules.loss.LpLoss (loss.py:4-46,
    the relative-L2 form the routines use) on the HIP loss kernel, value and gradient."""
    import math
    from fourierflow_amd.modules import fourier_encode
    from fourierflow_amd.modules.loss import LpLoss
    pos = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 5), torch.linspace(-1, 1, 7), indexing="ij"), dim=-1)
    enc = fourier_encode(pos, 32, 8, base=2)
    assert tuple(enc.shape) == (5, 7, 2, 17)
    scales = 2.0 ** torch.linspace(0, math.log2(16), 8)
    ref = torch.cat([(pos[..., None] * scales * math.pi).sin(), (pos[..., None] * scales * math.pi).cos(), pos[..., None]], -1)
    np.testing.assert_allclose(enc.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    g = gu.load_golden("position")                                   # the reference's own outputs
    np.testing.assert_allclose(fourier_encode(torch.tensor(g["pos"]), 32, 8, base=2).numpy(), g["enc"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(fourier_encode(torch.tensor(g["pos"]), 20, 4, base=3).numpy(), g["enc_b3"], rtol=1e-6, atol=1e-6)
    rs = np.random.RandomState(4)
    a = torch.tensor(rs.standard_normal((3, 6, 6, 1)).astype(np.float32), requires_grad=True)
    b = torch.tensor(rs.standard_normal((3, 6, 6, 1)).astype(np.float32))
    want = orc.lp_rel_loss(a, b)
    want.backward()
    ad = a.detach().to(host_device).requires_grad_(True)
    got = LpLoss(size_average=True)(ad, b.to(host_device))
    got.backward()
    assert abs(got.item() - want.item()) < 1e-6
    assert rel_l2(ad.grad.cpu().numpy(), a.grad.numpy()) < 1e-5
    with pytest.raises(NotImplementedError):
        LpLoss(p=1)
