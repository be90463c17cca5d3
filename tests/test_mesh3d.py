"""FNOFactorizedMesh3D (SURVEY 8 row a8 / BASELINE config 5): oracle vs the reference's golden vectors, and
the HIP path (3 spectral axes as views, +8 zero padding / crop as index maps, multi-output head) vs both."""
import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc

TAGS = ["c32_small", "c64_shared"]


def oracle_run(kw, seed, B, S, dtype=torch.float32, relu_masks=None):
    import oracle_util as ou
    sd, uniq = ou.torch_state_dict(gu.make_mesh3d_state_dict(kw, seed), dtype)
    x_np, t_np = gu.make_mesh3d_io(kw, seed, B, S)
    out = orc.ffno_mesh3d(sd, torch.tensor(x_np, dtype=dtype), modes=(kw["modes_x"], kw["modes_y"], kw["modes_z"]),
                          n_layers=kw["n_layers"], relu_masks=relu_masks)
    loss = ((out - torch.tensor(t_np, dtype=dtype)) ** 2).mean()
    loss.backward()
    return out, loss, {k: p.grad.detach().numpy() for k, p in uniq.items()}


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_mesh3d_matches_reference_golden(tag):
    g = gu.load_golden("mesh3d_" + tag)
    kw = gu.golden_kwargs(g)
    meta = [int(v) for v in g["meta"]]
    B, S, seed = meta[0], tuple(meta[1:4]), meta[4]
    out, loss, grads = oracle_run(kw, seed, B, S)
    assert gu.compare_packed(g, "out", out.detach().numpy(), 2e-5) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    for n in [k for k in gu.packed_names(g) if k.startswith("grad.")]:
        assert gu.compare_packed(g, n, grads[n[5:]], 2e-5) < 5e-5, n


@pytest.mark.parametrize("tag", TAGS)
def test_mesh3d_hip_path_matches_reference_golden(tag, host_device):
    from fourierflow_amd.modules import FNOFactorizedMesh3D
    if tag == "c64_shared" and str(host_device) == "cpu":
        pytest.skip("the width-64 golden (2856 padded pixels) runs on the GPU; the emulator covers c32_small")
    g = gu.load_golden("mesh3d_" + tag)
    kw = gu.golden_kwargs(g)
    meta = [int(v) for v in g["meta"]]
    B, S, seed = meta[0], tuple(meta[1:4]), meta[4]
    blk = FNOFactorizedMesh3D(**kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in gu.make_mesh3d_state_dict(kw, seed).items()}
    assert set(blk.state_dict().keys()) == set(sd.keys())
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(host_device)
    x_np, t_np = gu.make_mesh3d_io(kw, seed, B, S)
    out = blk(torch.from_numpy(x_np).to(host_device))
    assert tuple(out.shape) == (B, *S, kw["output_dim"])
    assert gu.compare_packed(g, "out", out.detach().cpu().numpy(), 1e-5) < 1e-5
    loss = ((out - torch.from_numpy(t_np).to(host_device)) ** 2).mean()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    loss.backward()
    named = dict(blk.named_parameters())
    errs = {n: gu.compare_packed(g, n, named[n[5:]].grad.cpu().numpy(), 1e-5)
            for n in gu.packed_names(g) if n.startswith("grad.")}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 3e-3, (worst, errs[worst])       # ReLU bit-flip discontinuity, see tests/test_block.py
    assert float(np.median(list(errs.values()))) < 3e-4, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    # without the discontinuity: the oracle on the HIP path's ReLU active sets, every gradient at rounding level
    import oracle_util as ou
    eng = blk.engine()
    masks = ou.engine_relu_masks(eng)
    print(f"[mesh3d {tag} {host_device}] worst gradient vs reference golden {errs[worst]:.2e} ({worst})")
    ou.check_grads_at_rounding_level(f"mesh3d {tag} {host_device}", {n: named[n].grad.cpu().numpy() for n in eng.param_names},
                                     lambda dt: oracle_run(kw, seed, B, S, dtype=dt, relu_masks=masks)[2])


@pytest.mark.gpu
def test_mesh3d_baseline_config5_shape_on_gpu():
    """BASELINE config 5 (64x64x64 -> 72^3 padded, modes 8, width 32): forward vs the CPU oracle, one layer."""
    from fourierflow_amd.modules import FNOFactorizedMesh3D
    import oracle_util as ou
    kw = dict(modes_x=8, modes_y=8, modes_z=8, width=32, input_dim=4, output_dim=4, n_layers=1, share_weight=False,
              factor=4, ff_weight_norm=True, n_ff_layers=2, layer_norm=False)
    seed, B, S = 3, 1, (64, 64, 64)
    blk = FNOFactorizedMesh3D(**kw)
    sd_np = gu.make_mesh3d_state_dict(kw, seed)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    blk = blk.cuda()
    x_np, _ = gu.make_mesh3d_io(kw, seed, B, S)
    with torch.no_grad():
        out = blk(torch.from_numpy(x_np).cuda())
        sd, _ = ou.torch_state_dict(sd_np, torch.float32, requires_grad=False)
        ref = orc.ffno_mesh3d(sd, torch.from_numpy(x_np), modes=(8, 8, 8), n_layers=1)
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.gpu
def test_mesh3d_baseline_config5_backward_two_layers_on_gpu():
    """BASELINE config 5 shape WITH the backward pass: 64^3 -> 72^3 padded, modes 8, width 32, two layers, every
    parameter gradient against the oracle on the same ReLU active sets (<= 5e-5), forward <= 1e-5."""
    from fourierflow_amd.modules import FNOFactorizedMesh3D
    import oracle_util as ou
    kw = dict(modes_x=8, modes_y=8, modes_z=8, width=32, input_dim=4, output_dim=4, n_layers=2, share_weight=False,
              factor=4, ff_weight_norm=True, n_ff_layers=2, layer_norm=False)
    seed, B, S = 5, 1, (64, 64, 64)
    blk = FNOFactorizedMesh3D(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_mesh3d_state_dict(kw, seed).items()})
    blk = blk.cuda()
    x_np, t_np = gu.make_mesh3d_io(kw, seed, B, S)
    out = blk(torch.from_numpy(x_np).cuda())
    loss = ((out - torch.from_numpy(t_np).cuda()) ** 2).mean()
    loss.backward()
    eng = blk.engine()
    masks = ou.engine_relu_masks(eng)
    ref_out, ref_loss, _ = oracle_run(kw, seed, B, S, relu_masks=masks)
    e_fwd = rel_l2(out.detach().cpu().numpy(), ref_out.detach().numpy())
    named = dict(blk.named_parameters())
    print(f"[mesh3d 64^3 two layers] forward {e_fwd:.2e}")
    assert e_fwd < 1e-5
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * max(1.0, ref_loss.item())
    ou.check_grads_at_rounding_level("mesh3d 64^3 two layers", {n: named[n].grad.cpu().numpy() for n in eng.param_names},
                                     lambda dt: oracle_run(kw, seed, B, S, dtype=dt, relu_masks=masks)[2])


def test_structured_mesh_routine_train_step(host_device):
    """StructuredMeshExperiment (structured_mesh.py:21-31): model(x) -> LpLoss.rel -> manual optimisation; the loss
    of the first step equals the oracle's, and training reduces it."""
    import oracle_util as ou
    from fourierflow_amd.modules import FNOFactorizedMesh3D
    from fourierflow_amd.routines import StructuredMeshExperiment
    kw = dict(modes_x=3, modes_y=2, modes_z=2, width=32, input_dim=4, output_dim=2, n_layers=2, share_weight=False,
              factor=4, ff_weight_norm=True, n_ff_layers=2, layer_norm=False)
    seed, B, S = 9, 1, (4, 3, 2)
    sd_np = gu.make_mesh3d_state_dict(kw, seed)
    blk = FNOFactorizedMesh3D(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    exp = StructuredMeshExperiment(blk.to(host_device), scheduler=dict(num_warmup_steps=1, num_training_steps=20))
    x_np, t_np = gu.make_mesh3d_io(kw, seed, B, S)
    batch = dict(x=torch.from_numpy(x_np).to(host_device), y=torch.from_numpy(t_np).to(host_device))
    sd, _ = ou.torch_state_dict(sd_np, requires_grad=False)
    ref = orc.lp_rel_loss(orc.ffno_mesh3d(sd, torch.from_numpy(x_np), modes=(3, 2, 2), n_layers=2), torch.from_numpy(t_np))
    l0 = exp.training_step(batch).item()
    assert abs(l0 - ref.item()) < 1e-5
    l1 = exp.training_step(batch).item()
    l1 = exp.training_step(batch).item()
    assert l1 < l0
    assert abs(exp.validation_step(batch).item() - l1) < 0.2
