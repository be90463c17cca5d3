"""Generate tests/golden/fullsize_*.npz: sketches (tests/fullsize_util.py) of the CPU oracle's fp64 forward / loss / parameter
gradients at the full-size shapes whose `-m gpu` parity tests used to run the oracle live on the GPU box.

    python tools/make_fullsize_fixtures.py [tag ...]

Runs in the build container (no GPU, nothing of fourierflow_amd's product path is imported: oracle + seeded inputs only).
The oracle evaluates its OWN ReLU decisions here (no active sets of a HIP run exist at generation time); at these sizes the
handful of hidden units within an ulp of zero (< 1e-7 of them, counted by the live markov/24 test) move a gradient by far less
than the tolerance the tests hold."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import fullsize_util as fu      # noqa: E402
import golden_util as gu        # noqa: E402
import oracle_util as ou        # noqa: E402
from oracle import ffno_oracle as orc      # noqa: E402

MARKOV24 = dict(modes=16, width=64, input_dim=3, n_layers=24, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)
KOCHKOV256 = dict(width=64, input_dim=5, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)
MESH3D_CFG5 = dict(modes_x=8, modes_y=8, modes_z=8, width=32, input_dim=4, output_dim=1, n_layers=12, share_weight=False, factor=4,
                   ff_weight_norm=True, n_ff_layers=2, layer_norm=False)
PLASTICITY = dict(modes_x=32, modes_y=12, modes_z=8, width=64, input_dim=4, output_dim=4, n_layers=12, share_weight=False, factor=4,
                  ff_weight_norm=True, n_ff_layers=2, layer_norm=False)
AIRFOIL = dict(modes_x=32, modes_y=16, width=64, input_dim=4, n_layers=24, share_weight=False, factor=4, ff_weight_norm=True,
               n_ff_layers=2, layer_norm=False)
DT = torch.float64      # (switched to float32 for the second run: the oracle's own rounding noise, see main())


def block2d(kw, seed, B, M, N):
    out, loss, grads = ou.oracle_block_run(kw, seed, B, M, N, dtype=DT)
    return out["forecast"].detach().numpy(), float(loss.item()), grads


def mesh(fn, make_sd, make_io, kw, seed, B, S, modes):
    sd_np = make_sd(kw, seed)
    x_np, t_np = make_io(kw, seed, B, S)
    sd, uniq = ou.torch_state_dict(sd_np, DT)
    o = fn(sd, torch.tensor(x_np, dtype=DT), modes=modes, n_layers=kw["n_layers"])
    l = orc.lp_rel_loss(o, torch.tensor(t_np, dtype=DT))
    l.backward()
    return o.detach().numpy(), float(l.item()), {k: (p.grad.detach().numpy() if p.grad is not None else None) for k, p in uniq.items()}


CASES = {
    # tag: (description, thunk)  -- seeds / shapes are the ones of the tests that read the fixture
    "markov24_b32": ("torus_li/markov/24_layers, batch 32, 64 x 64 (seed 2024)", lambda: block2d(MARKOV24, 2024, 32, 64, 64)),
    "markov24_b19": ("torus_li/markov/24_layers, batch 19, 64 x 64 (seed 2024)", lambda: block2d(MARKOV24, 2024, 19, 64, 64)),
    "kochkov256_12l_k32": ("256 x 256, 12 layers, 32 modes, batch 2 (seed 288)",
                           lambda: block2d(dict(KOCHKOV256, modes=32, n_layers=12), 288, 2, 256, 256)),
    "kochkov256_24l_k64": ("256 x 256, 24 layers, 64 modes, batch 2 (seed 320)",
                           lambda: block2d(dict(KOCHKOV256, modes=64, n_layers=24), 320, 2, 256, 256)),
    "mesh3d_cfg5": ("FNOFactorizedMesh3D 64^3 -> 72^3, modes 8, width 32, 12 layers, batch 1 (seed 55)",
                    lambda: mesh(orc.ffno_mesh3d, gu.make_mesh3d_state_dict, gu.make_mesh3d_io, MESH3D_CFG5, 55, 1, (64, 64, 64), (8, 8, 8))),
    "plasticity": ("FNOFactorizedMesh3D [2, 101, 31, 20] -> 109 x 39 x 28, modes (32, 12, 8), width 64, 12 layers (seed 101)",
                   lambda: mesh(orc.ffno_mesh3d, gu.make_mesh3d_state_dict, gu.make_mesh3d_io, PLASTICITY, 101, 2, (101, 31, 20), (32, 12, 8))),
    "airfoil": ("FNOFactorizedMesh2D [10, 221, 51] -> 229 x 59, modes (32, 16), width 64, 24 layers (seed 221)",
                lambda: mesh(orc.ffno_mesh2d, gu.make_mesh2d_state_dict, gu.make_mesh2d_io, AIRFOIL, 221, 10, (221, 51), (32, 16))),
}


def main():
    global DT
    tags = sys.argv[1:] or list(CASES)
    torch.set_num_threads(os.cpu_count() or 8)
    for tag in tags:
        desc, thunk = CASES[tag]
        t0 = time.time()
        DT = torch.float64
        out, loss, grads = thunk()
        # the same run in fp32: |oracle(fp32) - oracle(fp64)| per tensor is the reference op sequence's OWN rounding noise (ReLU
        # decisions within an ulp of zero included) -- the scale a gradient that is a sum with heavy cancellation is held to
        # (tests/oracle_util.py::check_grads_at_rounding_level does the same with live runs)
        DT = torch.float32
        out32, loss32, grads32 = thunk()
        d = {"desc": np.array(desc), "loss": np.float64(loss), "dtype": np.array("float64"), "loss_fp32": np.float64(loss32)}
        d.update(fu.make_sketch(tag, "out", out))
        d["out::noise"] = np.float64(np.linalg.norm(out32.astype(np.float64) - out) / max(np.linalg.norm(out), 1e-30))
        for n, gr in grads.items():
            if gr is not None:
                d.update(fu.make_sketch(tag, "grad/" + n, gr))
                d[f"grad/{n}::noise"] = np.float64(np.linalg.norm(grads32[n].astype(np.float64) - gr) / max(np.linalg.norm(gr), 1e-30))
        path = os.path.join(fu.GOLDEN, f"fullsize_{tag}.npz")
        np.savez_compressed(path, **d)
        worst = max((float(d[k]) for k in d if k.endswith("::noise") and k.startswith("grad/")), default=0.0)
        print(f"[fullsize] {tag}: {len(grads)} gradients, loss {loss:.6f}, forward noise {float(d['out::noise']):.1e}, worst gradient noise "
              f"{worst:.1e}, {os.path.getsize(path) / 1024:.0f} KB, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
