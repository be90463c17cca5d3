"""FNOMesh2D / FNOMesh3D, the geo-FNO baselines of the airfoil / pipe / plasticity experiments (reference
zongyi_fno/mesh_2d.py:14-106, mesh_3d.py:7-113): the oracle against the reference's golden vectors, the HIP path against both
(complex64 state_dict, unequal mode counts, non-square meshes, widths 32 and 64, four 3-D corner blocks), and the
StructuredMeshExperiment step with torch.optim.Adam + StepLR."""
import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc

TAGS = ["c32", "c64", "3d_c32"]


def _case(tag):
    g = gu.load_golden("geofno_" + tag)
    kw = gu.golden_kwargs(g)
    B, X, Y, seed, Z = [int(v) for v in g["meta"]]
    sd_np = gu.make_geofno_state_dict(kw, seed)
    x, t = gu.make_geofno_io(seed, B, X, Y, Z)
    return g, kw, sd_np, x, t


def _oracle(kw, sd, x):
    if "modes3" in kw:
        return orc.fno_mesh3d(sd, x, modes1=kw["modes1"], modes2=kw["modes2"], modes3=kw["modes3"], n_layers=kw["n_layers"])
    return orc.fno_mesh2d(sd, x, modes1=kw["modes1"], modes2=kw["modes2"], n_layers=kw["n_layers"])


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_geofno_matches_reference_golden(tag):
    g, kw, sd_np, x, t = _case(tag)
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in sd_np.items()}
    out = _oracle(kw, sd, torch.tensor(x))
    loss = ((out - torch.tensor(t)) ** 2).mean()
    loss.backward()
    assert gu.compare_packed(g, "out", out.detach().numpy(), 2e-5) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    for n in [k for k in gu.packed_names(g) if k.startswith("grad.")]:
        gr = sd[n[5:]].grad
        gr = torch.view_as_real(gr) if gr.is_complex() else gr
        assert gu.compare_packed(g, n, gr.numpy(), 2e-5) < 5e-5, n


@pytest.mark.parametrize("tag", TAGS)
def test_geofno_hip_path_matches_reference_golden(tag, host_device):
    from fourierflow_amd.modules import FNOMesh2D, FNOMesh3D
    if host_device == "cpu" and tag == "c64":
        pytest.skip("width-64 golden: GPU only (the wave emulator runs the width-32 goldens)")
    g, kw, sd_np, x, t = _case(tag)
    blk = (FNOMesh3D if "modes3" in kw else FNOMesh2D)(**kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    assert list(blk.state_dict().keys()) == list(sd.keys())
    assert all(blk.state_dict()[k].dtype == sd[k].dtype and blk.state_dict()[k].shape == sd[k].shape for k in sd)   # complex64
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(host_device)
    out = blk(torch.from_numpy(x).to(host_device))
    assert gu.compare_packed(g, "out", out.detach().cpu().numpy(), 1e-5) < 1e-5
    loss = ((out - torch.from_numpy(t).to(host_device)) ** 2).mean()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    loss.backward()
    named = dict(blk.named_parameters())
    errs = {n: gu.compare_packed(g, n, named[n[5:]].grad.cpu().numpy(), 1e-5)
            for n in gu.packed_names(g) if n.startswith("grad.")}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 5e-5, (worst, errs[worst])          # GELU is smooth: no ReLU bit-flip caveat here


def test_geofno_structured_mesh_routine_adam_steplr(host_device):
    """experiments/pipe/geo-fno/*: StructuredMeshExperiment + FNOMesh2D + torch.optim.Adam(lr 1e-3, wd 1e-4) +
    StepLR(100, 0.5), loss_scale 20: two optimisation steps against torch autograd through the oracle."""
    from fourierflow_amd.modules import FNOMesh2D
    from fourierflow_amd.routines import StructuredMeshExperiment
    g, kw, sd_np, x, t = _case("c32")
    blk = FNOMesh2D(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    routine = StructuredMeshExperiment(blk, loss_scale=20, optimizer=dict(lr=1e-3, weight_decay=1e-4), optimizer_type="adam",
                                       scheduler=dict(step_size=1, gamma=0.5)).to(host_device)
    sd = {k: torch.tensor(v, requires_grad=True) for k, v in sd_np.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    xb, tb = torch.from_numpy(x).to(host_device), torch.from_numpy(t).to(host_device)
    for step in range(2):
        opt.zero_grad()
        out = _oracle(kw, sd, torch.tensor(x))
        lref = orc.lp_rel_loss(out, torch.tensor(t))
        (lref * 20).backward()
        opt.step()
        sch.step()
        loss = routine.training_step(dict(x=xb, y=tb))
        routine.on_train_epoch_end()
        assert abs(loss.item() - lref.item()) < 1e-5
    named = dict(blk.named_parameters())
    for n, ref in sd.items():
        r = torch.view_as_real(ref.detach()) if ref.is_complex() else ref.detach()
        assert rel_l2(named[n].detach().cpu().numpy(), r.numpy()) < 1e-5, n
    assert abs(routine.trainer().current_lr() - 1e-3 * 0.25) < 1e-12
