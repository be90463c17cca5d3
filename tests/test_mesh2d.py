"""FNOFactorizedMesh2D (the airfoil / pipe / elasticity operator): oracle vs the reference's golden
vectors, and the HIP path (weight 0 on the FIRST axis, independent modes per axis, +8 zero padding / crop as index maps)
vs both."""
import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc

TAGS = ["c32_small", "c64_shared"]


def oracle_run(kw, seed, B, S, dtype=torch.float32):
    import oracle_util as ou
    sd, uniq = ou.torch_state_dict(gu.make_mesh2d_state_dict(kw, seed), dtype)
    x_np, t_np = gu.make_mesh2d_io(kw, seed, B, S)
    out = orc.ffno_mesh2d(sd, torch.tensor(x_np, dtype=dtype), modes=(kw["modes_x"], kw["modes_y"]), n_layers=kw["n_layers"])
    loss = ((out - torch.tensor(t_np, dtype=dtype)) ** 2).mean()
    loss.backward()
    return out, loss, {k: p.grad.detach().numpy() for k, p in uniq.items()}


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_mesh2d_matches_reference_golden(tag):
    g = gu.load_golden("mesh2d_" + tag)
    kw = gu.golden_kwargs(g)
    meta = [int(v) for v in g["meta"]]
    B, S, seed = meta[0], tuple(meta[1:3]), meta[3]
    out, loss, grads = oracle_run(kw, seed, B, S)
    assert gu.compare_packed(g, "out", out.detach().numpy(), 2e-5) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    for n in [k for k in gu.packed_names(g) if k.startswith("grad.")]:
        assert gu.compare_packed(g, n, grads[n[5:]], 2e-5) < 5e-5, n


@pytest.mark.parametrize("tag", TAGS)
def test_mesh2d_hip_path_matches_reference_golden(tag, host_device):
    from fourierflow_amd.modules import FNOFactorizedMesh2D
    g = gu.load_golden("mesh2d_" + tag)
    kw = gu.golden_kwargs(g)
    meta = [int(v) for v in g["meta"]]
    B, S, seed = meta[0], tuple(meta[1:3]), meta[3]
    blk = FNOFactorizedMesh2D(**kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in gu.make_mesh2d_state_dict(kw, seed).items()}
    assert set(blk.state_dict().keys()) == set(sd.keys())
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(host_device)
    x_np, t_np = gu.make_mesh2d_io(kw, seed, B, S)
    out = blk(torch.from_numpy(x_np).to(host_device))
    assert tuple(out.shape) == (B, *S, 1)
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


@pytest.mark.gpu
def test_mesh2d_airfoil_shape_on_gpu():
    """experiments/airfoil/ffno/*/config.yaml: 221 x 51 mesh (padded 229 x 59), modes 32 / 16, width 64: forward vs the oracle."""
    from fourierflow_amd.modules import FNOFactorizedMesh2D
    import oracle_util as ou
    kw = dict(modes_x=32, modes_y=16, width=64, input_dim=4, n_layers=2, share_weight=False, factor=4, ff_weight_norm=True,
              n_ff_layers=2, layer_norm=False)
    seed, B, S = 5, 2, (221, 51)
    blk = FNOFactorizedMesh2D(**kw)
    sd_np = gu.make_mesh2d_state_dict(kw, seed)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    blk = blk.cuda()
    x_np, _ = gu.make_mesh2d_io(kw, seed, B, S)
    with torch.no_grad():
        out = blk(torch.from_numpy(x_np).cuda())
        sd, _ = ou.torch_state_dict(sd_np, torch.float32, requires_grad=False)
        ref = orc.ffno_mesh2d(sd, torch.from_numpy(x_np), modes=(32, 16), n_layers=2)
    assert rel_l2(out.cpu().numpy(), ref.numpy()) < 1e-5


def test_mesh2d_structured_mesh_routine(host_device):
    """StructuredMeshExperiment drives the 2-D mesh operator exactly like the 3-D one (structured_mesh.py:21-31)."""
    import oracle_util as ou
    from fourierflow_amd.modules import FNOFactorizedMesh2D
    from fourierflow_amd.routines import StructuredMeshExperiment
    kw = dict(modes_x=3, modes_y=2, width=32, input_dim=4, n_layers=2, share_weight=False, factor=4, ff_weight_norm=True,
              n_ff_layers=2, layer_norm=False)
    seed, B, S = 11, 2, (6, 5)
    sd_np = gu.make_mesh2d_state_dict(kw, seed)
    blk = FNOFactorizedMesh2D(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    exp = StructuredMeshExperiment(blk.to(host_device), scheduler=dict(num_warmup_steps=1, num_training_steps=20))
    x_np, t_np = gu.make_mesh2d_io(kw, seed, B, S)
    batch = dict(x=torch.from_numpy(x_np).to(host_device), y=torch.from_numpy(t_np).to(host_device))
    sd, _ = ou.torch_state_dict(sd_np, requires_grad=False)
    ref = orc.lp_rel_loss(orc.ffno_mesh2d(sd, torch.from_numpy(x_np), modes=(3, 2), n_layers=2), torch.from_numpy(t_np))
    l0 = exp.training_step(batch).item()
    assert abs(l0 - ref.item()) < 1e-5
    for _ in range(4):
        l1 = exp.training_step(batch).item()
    assert l1 < l0


@pytest.mark.parametrize("x3", [True, False], ids=["x3", "fp32"])
def test_mesh2d_mixed_fused_and_staged_axes(host_device, x3):
    """modes_x = 20, modes_y = 5.  On the fp32-MFMA kernels 20 modes exceed the fused tile (K <= 16 at width 64) while 5 fit:
    the engine picks the kernel per axis (staged + fused).  On the split kernels both axes are fused -- the 20-mode axis on the
    4-line tile of spectral_x3k, the 5-mode axis on the 16- / 8-line tile -- as two launches: a many-mode axis does not share a
    paired launch with a <= 16-mode one (it would drag the short axis onto 4-line tiles: the airfoil mesh ran 91 instead of 104
    steps/s that way).  Forward and gradients vs the oracle's autograd."""
    import oracle_util as ou
    from fourierflow_amd.modules import FNOFactorizedMesh2D
    kw = dict(modes_x=20, modes_y=5, width=64, input_dim=4, n_layers=2, share_weight=False, factor=4, ff_weight_norm=True,
              n_ff_layers=2, layer_norm=False)
    seed, B, S = 21, 1, (32, 8)
    sd_np = gu.make_mesh2d_state_dict(kw, seed)
    blk = FNOFactorizedMesh2D(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    blk = blk.to(host_device)
    blk.engine().use_x3 = x3
    x_np, t_np = gu.make_mesh2d_io(kw, seed, B, S)
    out = blk(torch.from_numpy(x_np).to(host_device))
    eng = blk.engine()
    assert eng._can_fuse(eng._ws.views) == ([True, True] if x3 else [False, True])
    assert eng._saved_x3 == (([True, True], False) if x3 else ([False, False], False))
    loss = ((out - torch.from_numpy(t_np).to(host_device)) ** 2).mean()
    loss.backward()
    ref_out, ref_loss, ref_grads = oracle_run(kw, seed, B, S, torch.float64)
    assert rel_l2(out.detach().cpu().numpy(), ref_out.detach().numpy()) < 1e-5
    named = dict(blk.named_parameters())
    errs = {n: rel_l2(named[n].grad.cpu().numpy(), g) for n, g in ref_grads.items()}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 3e-3, (worst, errs[worst])
    assert float(np.median(list(errs.values()))) < 3e-4
