"""Grid2DRolloutExperiment + FNOZongyi2DBlock (BASELINE config 0: experiments/torus_li/zongyi/4_layers) on the HIP path:
one optimisation step through the n_steps autoregressive rollout against the oracle's restatement of
routines/grid_2d_rollout.py:75-150 + torch.optim.AdamW + StepLR, and the validation metrics."""
import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc

KW = dict(modes1=3, modes2=3, width=20, input_dim=12, n_layers=2)
B, G, T = 2, 12, 3


def _setup(device, **routine_kw):
    from fourierflow_amd.modules import FNOZongyi2DBlock
    from fourierflow_amd.routines import Grid2DRolloutExperiment
    sd_np, _ = gu.make_zongyi_state_dict(KW, 7)
    conv = FNOZongyi2DBlock(**KW)
    conv.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    routine = Grid2DRolloutExperiment(conv, n_steps=T, optimizer=dict(lr=2.5e-3, weight_decay=1e-4),
                                      scheduler=dict(step_size=2, gamma=0.5), **routine_kw).to(device)
    rs = np.random.RandomState(3)
    data = (rs.standard_normal((B, G, G, 10 + T)) * 0.5).astype(np.float32)
    return routine, sd_np, data


def _oracle_step(sd_np, data, lr, teacher_forcing=False):
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in sd_np.items()}
    xx = torch.cat([torch.tensor(data[..., :10]), orc.rollout_positions(B, G, G)], dim=-1)
    yy = torch.tensor(data[..., 10:])
    conv = lambda z: orc.fno_zongyi_2d(sd, z, modes=KW["modes1"], n_layers=KW["n_layers"])["forecast"]   # noqa: E731
    out = orc.rollout_learning_step(conv, xx, yy, T, teacher_forcing=teacher_forcing)
    opt = torch.optim.AdamW(list(sd.values()), lr=lr, weight_decay=1e-4)
    out[0].backward()
    grads = {k: v.grad.clone() for k, v in sd.items()}
    opt.step()
    return out, grads, {k: v.detach() for k, v in sd.items()}


# ---- the loop of routines/grid_2d_rollout.py:38-50,75-150 pinned by a run of the reference's own method bodies --------------------
# (tools/make_golden_rollout.py lifts `forward` / `_learning_step` from the reference file and runs them over the reference's
#  real FNOZongyi2DBlock and LpLoss; the fixture holds inputs / outputs / parameter gradients only)
ROLLOUT_CASES = ["pos", "nopos", "tf_train", "tf_eval"]


def _golden_case(tag):
    g = gu.load_golden("rollout_step")
    ap, tf, training, wseed, _ = [int(v) for v in g[f"{tag}.flags"]]
    sd_np, _ = gu.make_zongyi_state_dict(KW, wseed)
    return g, bool(ap), bool(tf), bool(training), float(g[f"{tag}.step_size"]), sd_np, g[f"{tag}.data"]


@pytest.mark.parametrize("tag", ROLLOUT_CASES)
def test_oracle_rollout_matches_reference_golden(tag):
    g, ap, tf, training, step_size, sd_np, data = _golden_case(tag)
    assert [int(v) for v in g["meta"]] == [B, G, T]
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in sd_np.items()}
    xx = torch.cat([torch.tensor(data[..., :10]), orc.rollout_positions(B, G, G)], dim=-1)
    conv = lambda z: orc.fno_zongyi_2d(sd, z, modes=KW["modes1"], n_layers=KW["n_layers"])["forecast"]   # noqa: E731
    loss, loss_full, pred, step_losses, p, time_until = orc.rollout_learning_step(
        conv, xx, torch.tensor(data[..., 10:]), T, append_pos=ap, teacher_forcing=tf, training=training, step_size=step_size)
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}.loss"])) < 2e-6 and abs(loss_full.item() - float(g[f"{tag}.loss_full"])) < 2e-6
    assert rel_l2(pred.detach().numpy(), g[f"{tag}.pred"]) < 2e-6
    np.testing.assert_allclose([l.item() for l in step_losses], g[f"{tag}.step_losses"], atol=2e-6)
    np.testing.assert_allclose(p.detach().numpy(), g[f"{tag}.p"], atol=2e-6)
    assert float(time_until) == float(g[f"{tag}.time_until"])
    for k, v in sd.items():
        assert rel_l2(v.grad.numpy(), g[f"{tag}.grad.{k}"]) < 2e-5, k


@pytest.mark.parametrize("tag", ROLLOUT_CASES)
def test_rollout_hip_path_matches_reference_golden(host_device, tag):
    """The HIP routine (Grid2DRolloutExperiment.forward over the FNOZongyi2DBlock engine) against the reference run itself."""
    from fourierflow_amd.modules import FNOZongyi2DBlock
    from fourierflow_amd.routines import Grid2DRolloutExperiment
    g, ap, tf, training, step_size, sd_np, data = _golden_case(tag)
    conv = FNOZongyi2DBlock(**KW)
    conv.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    routine = Grid2DRolloutExperiment(conv, n_steps=T, append_pos=ap, teacher_forcing=tf, step_size=step_size).to(host_device)
    routine.train(training)
    eng = routine.trainer().engine
    conv.fused_grad_accumulation = True          # what training_step sets: the n_steps passes add their gradients up in the engine
    conv.max_live_passes = max(conv.max_live_passes, T)
    eng.zero_grad()
    loss, loss_full, pred, step_losses, p, time_until = routine.forward({'data': torch.from_numpy(data.copy()).to(host_device)})
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}.loss"])) < 1e-5 and abs(loss_full.item() - float(g[f"{tag}.loss_full"])) < 1e-5
    assert rel_l2(pred.detach().cpu().numpy(), g[f"{tag}.pred"]) < 1e-5
    np.testing.assert_allclose([l.item() for l in step_losses], g[f"{tag}.step_losses"], atol=1e-5)
    np.testing.assert_allclose(p.cpu().numpy(), g[f"{tag}.p"], atol=1e-5)
    assert float(time_until) == float(g[f"{tag}.time_until"])
    for n in eng.param_names:
        assert rel_l2(eng.grad_view(n).cpu().numpy(), g[f"{tag}.grad.{n}"]) < 2e-4, n


@pytest.mark.parametrize("teacher_forcing", [False, True])
def test_rollout_training_step_matches_oracle(host_device, teacher_forcing):
    routine, sd_np, data = _setup(host_device, teacher_forcing=teacher_forcing)
    (loss_ref, full_ref, *_), grads_ref, new_ref = _oracle_step(sd_np, data, 2.5e-3, teacher_forcing)
    batch = {'data': torch.from_numpy(data).to(host_device)}
    xx = torch.cat([batch['data'][..., :10], routine._positions(B, G, G, host_device)], dim=-1)
    loss, loss_full = routine.training_step({'x': xx, 'y': batch['data'][..., 10:]})
    assert abs(loss.item() - loss_ref.item()) < 1e-5 and abs(loss_full.item() - full_ref.item()) < 1e-5
    eng = routine.trainer().engine
    for n in eng.param_names:
        assert rel_l2(eng.grad_view(n).cpu().numpy(), grads_ref[n].numpy()) < 2e-4, n
    new = dict(routine.conv.named_parameters())
    for n in eng.param_names:     # AdamW's first step moves every entry by ~lr: compare the moved parameters
        assert rel_l2(new[n].detach().cpu().numpy(), new_ref[n].numpy()) < 1e-5, n
    assert all(p.grad is None for p in routine.conv.parameters())       # gradients live in the engine's flat buffer


def test_rollout_step_lr_and_validation_metrics(host_device):
    routine, sd_np, data = _setup(host_device)
    tr = routine.trainer()
    lrs = []
    for epoch in range(5):
        lrs.append(tr.current_lr())
        routine.on_train_epoch_end()
    assert lrs == [2.5e-3, 2.5e-3, 1.25e-3, 1.25e-3, 6.25e-4]          # StepLR(step_size=2, gamma=0.5), per epoch
    out = routine.forward({'data': torch.from_numpy(data).to(host_device)})      # :38-50 entry point (builds positions)
    sd = {k: torch.tensor(v) for k, v in sd_np.items()}
    xx = torch.cat([torch.tensor(data[..., :10]), orc.rollout_positions(B, G, G)], dim=-1)
    conv = lambda z: orc.fno_zongyi_2d(sd, z, modes=KW["modes1"], n_layers=KW["n_layers"])["forecast"]   # noqa: E731
    ref = orc.rollout_learning_step(conv, xx, torch.tensor(data[..., 10:]), T)
    assert abs(out[0].item() - ref[0].item()) < 1e-5 and abs(out[1].item() - ref[1].item()) < 1e-5
    assert rel_l2(out[2].detach().cpu().numpy(), ref[2].numpy()) < 1e-5
    np.testing.assert_allclose(out[4].cpu().numpy(), ref[4].numpy(), atol=1e-5)
    assert float(out[5]) == float(ref[5])
    val = routine.validation_step({'x': xx.to(host_device), 'y': torch.from_numpy(data[..., 10:]).to(host_device)})
    assert abs(val['valid_loss'].item() - ref[1].item()) < 1e-5
    test = routine.test_step({'x': xx.to(host_device), 'y': torch.from_numpy(data[..., 10:]).to(host_device)})
    assert test['test_losses'].shape == (T,) and test['test_correlations'].shape == (T,)


def test_markov_routine_runs_the_zongyi_ablation(host_device):
    """torus_li/ablation/zongyi_markov_residual: Grid2DMarkovExperiment + FNOZongyi2DBlock(residual, no conv residual) +
    StepLR, no normaliser: two training steps move the loss and follow the per-epoch schedule."""
    from fourierflow_amd.modules import FNOZongyi2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    torch.manual_seed(0)
    conv = FNOZongyi2DBlock(modes1=3, modes2=3, width=20, n_layers=2, input_dim=3, residual=True, conv_residual=False)
    routine = Grid2DMarkovExperiment(conv, n_steps=2, should_normalize=False, optimizer=dict(lr=2.5e-3, weight_decay=1e-4),
                                     scheduler=dict(step_size=100, gamma=0.5)).to(host_device)
    x = torch.randn(2, 12, 12, 1, device=host_device)
    y = torch.randn(2, 12, 12, 1, device=host_device)
    l0 = routine.training_step({'x': x, 'y': y}, epoch=0).item()
    for _ in range(2):
        l1 = routine.training_step({'x': x, 'y': y}, epoch=0).item()
    assert l1 < l0
    assert routine.trainer().current_lr() == 2.5e-3
    routine.training_step({'x': x, 'y': y}, epoch=100)
    assert routine.trainer().current_lr() == 1.25e-3
