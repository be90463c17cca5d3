"""CNOFactorized2DBlock / CNOFactorizedMesh2D / CNOFactorizedMesh3D (reference factorized_cno/*.py + modules/dct.py): the
F-FNO operators with an orthonormal DCT-II per axis and real per-mode weights.  Oracle (explicit DCT matrices) against the
reference's golden vectors; HIP path (DCT through the truncated real-DFT kernels) against the same goldens."""
import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import host_device, rel_l2  # noqa: F401
from oracle import ffno_oracle as orc

TAGS = ["2d_shared", "mesh2d_c64", "mesh3d_c32"]


def _case(tag):
    g = gu.load_golden("cno_" + tag)
    kw, kind = gu.golden_kwargs(g), str(g["kind"])
    B, seed, *S = [int(v) for v in g["meta"]]
    sd_np, x, t = gu.make_cno_case(kind, kw, seed, B, tuple(S))
    return g, kind, kw, sd_np, x, t


def _oracle(kind, kw, sd, x):
    if kind == "2d":
        return orc.ffno2d_block(sd, x, modes=kw["modes"], n_layers=kw["n_layers"])["forecast"]
    if kind == "mesh2d":
        return orc.ffno_mesh2d(sd, x, modes=(kw["modes_x"], kw["modes_y"]), n_layers=kw["n_layers"])
    return orc.ffno_mesh3d(sd, x, modes=(kw["modes_x"], kw["modes_y"], kw["modes_z"]), n_layers=kw["n_layers"])


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_cno_matches_reference_golden(tag):
    import oracle_util as ou
    g, kind, kw, sd_np, x, t = _case(tag)
    sd, uniq = ou.torch_state_dict(sd_np)
    out = _oracle(kind, kw, sd, torch.tensor(x))
    loss = ((out - torch.tensor(t)) ** 2).mean()
    loss.backward()
    assert gu.compare_packed(g, "out", out.detach().numpy(), 2e-5) < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    for n in [k for k in gu.packed_names(g) if k.startswith("grad.")]:
        assert gu.compare_packed(g, n, uniq[n[5:]].grad.numpy(), 2e-5) < 1e-4, n


@pytest.mark.parametrize("tag", TAGS)
def test_cno_hip_path_matches_reference_golden(tag, host_device):
    from fourierflow_amd import modules
    g, kind, kw, sd_np, x, t = _case(tag)
    if host_device == "cpu" and tag == "mesh2d_c64":
        pytest.skip("width-64 golden: GPU only")
    cls = {"2d": modules.CNOFactorized2DBlock, "mesh2d": modules.CNOFactorizedMesh2D, "mesh3d": modules.CNOFactorizedMesh3D}[kind]
    blk = cls(**kw)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    assert set(blk.state_dict().keys()) == set(sd.keys())
    assert all(tuple(blk.state_dict()[k].shape) == tuple(sd[k].shape) for k in sd)        # real [in, out, modes] weights
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(host_device)
    out = blk(torch.from_numpy(x).to(host_device))
    out = out["forecast"] if isinstance(out, dict) else out
    assert gu.compare_packed(g, "out", out.detach().cpu().numpy(), 1e-5) < 1e-5
    loss = ((out - torch.from_numpy(t).to(host_device)) ** 2).mean()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * max(1.0, float(g["loss"]))
    loss.backward()
    named = dict(blk.named_parameters())
    errs = {n: gu.compare_packed(g, n, named[n[5:]].grad.cpu().numpy(), 1e-5)
            for n in gu.packed_names(g) if n.startswith("grad.")}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 3e-3, (worst, errs[worst])       # ReLU bit-flip discontinuity, see tests/test_block.py
    assert float(np.median(list(errs.values()))) < 3e-4
