"""Full-size parity at the reference's REAL mesh shapes (VERDICT r04 "missing" #4), under ``-m gpu``:

  * plasticity  -- experiments/plasticity/ffno/12_layers/config.yaml:12-14,24-35: FNOFactorizedMesh3D on [2, 101, 31, 20] meshes,
    padded to 109 x 39 x 28 (109 is PRIME; mesh_3d.py:165), modes (32, 12, 8), width 64, 12 layers, output_dim 4 -- the only
    shipped 3-D shape with a 17..64-mode axis at width 64 (the many-mode fused kernel on a strided 3-D view);
  * airfoil     -- experiments/airfoil/ffno/24_layers/config.yaml:16,24-34: FNOFactorizedMesh2D on [10, 221, 51] meshes (229 x 59
    padded), modes (32, 16), width 64, 24 layers.

Forward <= 1e-5 relative L2, loss <= 1e-5 and EVERY parameter gradient against the committed sketch of the fp64 oracle run on the same
seeded inputs (tests/fullsize_util.py: 16 random projections + 256 sampled entries per tensor, generated in the build container).
Reference: fourierflow/modules/factorized_fno/mesh_3d.py:120,154-177, mesh_2d.py:107-175.
"""
import numpy as np
import pytest
import torch

import fullsize_util as fu
import golden_util as gu
from oracle import ffno_oracle as orc

# gradients against the SKETCH of the fp64 oracle run (the oracle's own ReLU decisions; see tests/test_bench_geometry.py)
SKETCH_GRAD_TOL = 2e-4


def _check(label, tag, blk, out, loss, grad_tol=SKETCH_GRAD_TOL):
    """Forward <= 1e-5, loss <= 1e-5, every parameter gradient inside SKETCH_GRAD_TOL of the committed sketch of the fp64 oracle run
    on the same seeded inputs (tests/fullsize_util.py, tools/make_fullsize_fixtures.py; VERDICT r05 #7: until round 5 these tests ran
    the oracle live -- 48 + 58 s of the GPU run -- with the HIP path's ReLU active sets injected; that form stays at markov/24)."""
    eng = blk.engine()
    named = dict(blk.named_parameters())
    fu.check(label, tag, out.detach().cpu().numpy(), loss.item(), {n: named[n].grad.cpu().numpy() for n in eng.param_names},
             grad_tol=grad_tol)


@pytest.mark.gpu
def test_plasticity_real_shape_forward_backward_vs_oracle():
    from fourierflow_amd.modules import FNOFactorizedMesh3D
    kw = dict(modes_x=32, modes_y=12, modes_z=8, width=64, input_dim=4, output_dim=4, n_layers=12, share_weight=False, factor=4,
              ff_weight_norm=True, n_ff_layers=2, layer_norm=False)
    seed, B, S = 101, 2, (101, 31, 20)
    sd_np = gu.make_mesh3d_state_dict(kw, seed)
    blk = FNOFactorizedMesh3D(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    blk = blk.cuda()
    x_np, t_np = gu.make_mesh3d_io(kw, seed, B, S)
    out = blk(torch.from_numpy(x_np).cuda())
    loss = orc.lp_rel_loss(out, torch.from_numpy(t_np).cuda())
    loss.backward()
    eng = blk.engine()
    assert [v.L for v in eng._ws.views] == [109, 39, 28] and [v.K for v in eng._ws.views] == [32, 12, 8]
    assert all(eng._saved_x3[0]), eng._saved_x3      # every axis on the fused split kernels (x: the many-mode kernel on a prime length)
    _check("plasticity 109x39x28 12L", "plasticity", blk, out, loss)


@pytest.mark.gpu
def test_airfoil_real_shape_forward_backward_vs_oracle():
    from fourierflow_amd.modules import FNOFactorizedMesh2D
    kw = dict(modes_x=32, modes_y=16, width=64, input_dim=4, n_layers=24, share_weight=False, factor=4, ff_weight_norm=True,
              n_ff_layers=2, layer_norm=False)
    seed, B, S = 221, 10, (221, 51)
    sd_np = gu.make_mesh2d_state_dict(kw, seed)
    blk = FNOFactorizedMesh2D(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    blk = blk.cuda()
    x_np, t_np = gu.make_mesh2d_io(kw, seed, B, S)
    out = blk(torch.from_numpy(x_np).cuda())
    loss = orc.lp_rel_loss(out, torch.from_numpy(t_np).cuda())
    loss.backward()
    eng = blk.engine()
    assert [v.L for v in eng._ws.views] == [229, 59] and all(eng._saved_x3[0]), eng._saved_x3
    # (observed on MI355X, round 6: worst 2.2e-4 at a weight-norm gain whose own fp32-oracle noise is 2.3e-4)
    _check("airfoil 229x59 24L", "airfoil", blk, out, loss)
