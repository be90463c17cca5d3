"""FNOZongyi2DBlock on the HIP path (BASELINE config 0; reference zongyi_fno/grid_2d.py:16-129):
the block against the reference's golden vectors (forward, loss, every parameter gradient) and against the oracle on
ragged shapes, including the input gradient and several live passes (what Grid2DRolloutExperiment's rollout needs)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc


def _block(kw, sd_np, device):
    from fourierflow_amd.modules import FNOZongyi2DBlock
    blk = FNOZongyi2DBlock(**kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    assert list(blk.state_dict().keys()) == list(sd.keys())       # reference registration order (grid_2d.py:106-122)
    blk.load_state_dict(sd, strict=True)
    return blk.to(device)


@pytest.mark.parametrize("name", ["zongyi_4l", "zongyi_markov_residual"])
def test_zongyi_hip_path_matches_reference_golden(host_device, name):
    if host_device == "cpu" and name == "zongyi_4l":
        pytest.skip("64 x 64 config-0 shape: GPU only (the wave emulator covers the 16 x 16 golden and the ragged shapes)")
    g = gu.load_golden(name)
    kw = gu.golden_kwargs(g)
    grid = int(g["grid"]) if "grid" in g.files else 64
    sd_np, x = gu.make_zongyi_state_dict(kw, int(g["seed"]), grid=grid)
    blk = _block(kw, sd_np, host_device)
    out = blk(torch.from_numpy(x).to(host_device))["forecast"]
    assert tuple(out.shape) == (2, grid, grid, 1)
    assert gu.compare_packed(g, "forecast", out.detach().cpu().numpy(), 1e-5) < 1e-5     # BASELINE: <= 1e-5 rel-L2 fp32
    loss = (out ** 2).mean()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    loss.backward()
    named = dict(blk.named_parameters())
    errs = {n: gu.compare_packed(g, n, named[n[5:]].grad.cpu().numpy(), 1e-5)
            for n in gu.packed_names(g) if n.startswith("grad.")}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 1e-4, (worst, errs[worst])


def _small_case(width, input_dim, K, B, M, N, L, seed, **flags):
    kw = dict(modes1=K, modes2=K, width=width, input_dim=input_dim, n_layers=L, **flags)
    sd_np, _ = gu.make_zongyi_state_dict(kw, seed)
    rs = np.random.RandomState(seed + 1)
    x = rs.standard_normal((B, M, N, input_dim)).astype(np.float32)
    t = rs.standard_normal((B, M, N, 1)).astype(np.float32)
    return kw, sd_np, x, t


@pytest.mark.parametrize("width,input_dim,K,B,M,N,L,residual,conv_residual",
                         [(20, 12, 3, 2, 12, 12, 2, False, True), (7, 5, 2, 3, 9, 9, 1, True, True),
                          (32, 3, 4, 1, 16, 16, 3, False, False), (20, 3, 3, 2, 10, 10, 2, True, False)])
def test_zongyi_small_shapes_match_oracle_with_input_gradient(host_device, width, input_dim, K, B, M, N, L, residual,
                                                              conv_residual):
    kw, sd_np, x, t = _small_case(width, input_dim, K, B, M, N, L, seed=5, residual=residual, conv_residual=conv_residual)
    sd = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd_np.items()}
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = orc.fno_zongyi_2d(sd, xr, modes=K, n_layers=L, residual=residual, conv_residual=conv_residual)["forecast"]
    lref = orc.lp_rel_loss(ref, torch.tensor(t, dtype=torch.float64))
    lref.backward()

    blk = _block(kw, sd_np, host_device)
    xd = torch.from_numpy(x).to(host_device).requires_grad_(True)
    out = blk(xd)["forecast"]
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    loss = orc.lp_rel_loss(out, torch.from_numpy(t).to(host_device))
    loss.backward()
    assert rel_l2(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-4
    for n, p in blk.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), sd[n].grad.numpy()) < 1e-4, n


def test_zongyi_two_live_passes_chain_like_a_rollout(host_device):
    """y1 = f(x), y2 = f(cat(x[..., 1:], y1)): the gradient flows through both passes (grid_2d_rollout.py:104-134)."""
    kw, sd_np, x, t = _small_case(20, 4, 3, 2, 12, 12, 2, seed=9)
    sd = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd_np.items()}

    def rollout(f, x0):
        y1 = f(x0)
        y2 = f(torch.cat([x0[..., 1:], y1], dim=-1))
        return (y1 ** 2).mean() + (y2 ** 3).mean()

    lref = rollout(lambda z: orc.fno_zongyi_2d(sd, z, modes=3, n_layers=2)["forecast"], torch.tensor(x, dtype=torch.float64))
    lref.backward()
    blk = _block(kw, sd_np, host_device)
    loss = rollout(lambda z: blk(z)["forecast"], torch.from_numpy(x).to(host_device))
    assert abs(loss.item() - lref.item()) < 1e-5 * max(1.0, abs(lref.item()))
    loss.backward()
    for n, p in blk.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), sd[n].grad.numpy()) < 1e-4, n
    # more live passes than slots is an error, not silent corruption
    blk.max_live_passes = 1
    blk._engine = None
    xs = torch.from_numpy(x).to(host_device)
    a = blk(xs)["forecast"]
    b = blk(xs)["forecast"]
    with pytest.raises(RuntimeError, match="max_live_passes"):
        (a.sum() + b.sum()).backward()


def test_zongyi_rejects_unsupported_arguments(host_device):
    from fourierflow_amd.modules import FNOZongyi2DBlock
    with pytest.raises(NotImplementedError):
        FNOZongyi2DBlock(modes1=4, modes2=4, width=48).engine()
    with pytest.raises(NotImplementedError):
        FNOZongyi2DBlock(modes1=4, modes2=6, width=20)
    blk = FNOZongyi2DBlock(modes1=12, modes2=12, width=20, input_dim=3).to(host_device)
    with pytest.raises(ValueError, match="modes"):
        blk(torch.zeros(1, 16, 16, 3, device=host_device))
    with pytest.raises(ValueError, match="square"):      # the reference's irfft2(s=(N, M)) breaks on M != N (grid_2d.py:68)
        blk(torch.zeros(1, 32, 48, 3, device=host_device))
    with pytest.raises(Exception):
        blk(torch.zeros(1, 16, 16, 3, dtype=torch.float64, device=host_device))
