"""Full-size parity at the reference's REAL mesh shapes (VERDICT r04 "missing" #4), under ``-m gpu``:

  * plasticity  -- experiments/plasticity/ffno/12_layers/config.yaml:12-14,24-35: FNOFactorizedMesh3D on [2, 101, 31, 20] meshes,
    padded to 109 x 39 x 28 (109 is PRIME; mesh_3d.py:165), modes (32, 12, 8), width 64, 12 layers, output_dim 4 -- the only
    shipped 3-D shape with a 17..64-mode axis at width 64 (the many-mode fused kernel on a strided 3-D view);
  * airfoil     -- experiments/airfoil/ffno/24_layers/config.yaml:16,24-34: FNOFactorizedMesh2D on [10, 221, 51] meshes (229 x 59
    padded), modes (32, 16), width 64, 24 layers.

Forward <= 1e-5 relative L2 and EVERY parameter gradient at rounding level (5e-5) against the oracle evaluated on the HIP path's
ReLU active sets; the number of hidden units on which the oracle's own ReLU decisions differ is bounded and printed.
Reference: fourierflow/modules/factorized_fno/mesh_3d.py:120,154-177, mesh_2d.py:107-175.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
import oracle_util as ou
from backend_util import rel_l2
from oracle import ffno_oracle as orc

FLIP_BOUND = 1e-5      # fraction of hidden units within an ulp of zero (observed: printed by the tests)


def _flip_fraction(masks, run_plain):
    """Hidden units on which the oracle's OWN ReLU decisions (unmasked fp32 forward) differ from the HIP path's active sets."""
    orc.RELU_TRACE = trace = []
    try:
        with torch.no_grad():
            run_plain()
    finally:
        orc.RELU_TRACE = None
    flips = total = layer = 0
    for prefix, i, active in trace:
        if "backcast_ff" not in prefix or i != 0:
            continue
        m = masks[("backcast", layer)][0].reshape(-1)
        layer += 1
        flips += int((active.reshape(-1) != m).sum())
        total += m.numel()
    assert layer == len(masks)
    return flips, total


def _check(label, blk, out, loss, x_np, t_np, sd_np, oracle_fn, modes, n_layers):
    eng = blk.engine()
    masks = ou.engine_relu_masks(eng)

    def oracle(dtype=torch.float32, use_masks=True):
        sd, uniq = ou.torch_state_dict(sd_np, dtype)
        o = oracle_fn(sd, torch.tensor(x_np, dtype=dtype), modes=modes, n_layers=n_layers, relu_masks=masks if use_masks else None)
        l = orc.lp_rel_loss(o, torch.tensor(t_np, dtype=dtype))
        l.backward()
        return o, l, {k: (p.grad.detach().numpy() if p.grad is not None else None) for k, p in uniq.items()}

    ref_out, ref_loss, ref_grads = oracle()
    e_fwd = rel_l2(out.detach().cpu().numpy(), ref_out.detach().numpy())
    print(f"[{label}] forward rel-L2 {e_fwd:.2e}, |loss diff| {abs(loss.item() - ref_loss.item()):.2e}")
    assert e_fwd < 1e-5
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    named = dict(blk.named_parameters())
    first = {torch.float32: ref_grads}
    ou.check_grads_at_rounding_level(label, {n: named[n].grad.cpu().numpy() for n in eng.param_names},
                                     lambda dt: first.get(dt) or oracle(dt)[2])
    sd0, _ = ou.torch_state_dict(sd_np, torch.float32, requires_grad=False)
    flips, total = _flip_fraction(masks, lambda: oracle_fn(sd0, torch.from_numpy(x_np), modes=modes, n_layers=n_layers))
    print(f"[{label}] ReLU decisions that differ from the oracle's own: {flips} of {total} hidden units ({flips / total:.1e})")
    assert flips <= FLIP_BOUND * total, (flips, total)


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
    _check("plasticity 109x39x28 12L", blk, out, loss, x_np, t_np, sd_np, orc.ffno_mesh3d, (32, 12, 8), 12)


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
    _check("airfoil 229x59 24L", blk, out, loss, x_np, t_np, sd_np, orc.ffno_mesh2d, (32, 16), 24)
