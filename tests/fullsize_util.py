"""Sketches of full-size oracle results: the full-depth / full-batch parity tests of the secondary shapes used to run
the CPU oracle on the GPU box (forward + backward of 12-24 layer models: 30-65 s each, 5 min of the `-m gpu` run).  The oracle's
answer only depends on the seeded inputs, so it is computed ONCE in the build container (`tools/make_fullsize_fixtures.py`, fp64)
and committed as a sketch per tensor (`tests/golden/fullsize_*.npz`):

    n, ||T||_2, 256 sampled entries, and 16 Rademacher projections <T, s_k>  (s_k in {-1, +1}^n from a seeded bit stream)

For a candidate T' the projections give an unbiased estimate of the FULL-tensor error: E <T' - T, s_k>^2 = ||T' - T||^2, so
``sketch_rel_err`` = sqrt(mean_k <T' - T, s_k>^2) / ||T|| estimates the relative L2 error of the whole tensor (chi-square with 16
degrees of freedom: within a factor 0.6..1.5 of the true value with 99 % probability) -- every element is covered, nothing but 2 KB
per tensor is stored.  The sampled entries are an exact second check on 256 positions.
One live-oracle full-size test stays (tests/test_bench_geometry.py, markov/24 at batch 32)."""
import hashlib
import os

import numpy as np

NPROJ, NSAMP = 16, 256
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _seed(tag: str, name: str):
    return int.from_bytes(hashlib.sha256(f"{tag}/{name}".encode()).digest()[:8], "little")


def _signs(tag, name, n):
    """[NPROJ, n] int8 of +-1 from a seeded PCG64 byte stream (stable across numpy versions)."""
    rng = np.random.Generator(np.random.PCG64(_seed(tag, name)))
    bits = np.unpackbits(np.frombuffer(rng.bytes(NPROJ * ((n + 7) // 8)), dtype=np.uint8).reshape(NPROJ, -1), axis=1)[:, :n]
    return bits.astype(np.int8) * 2 - 1


def _project(tag, name, flat):
    s = _signs(tag, name, flat.size)
    return np.array([np.dot(flat, s[k].astype(np.float64)) for k in range(NPROJ)], dtype=np.float64)


def make_sketch(tag: str, name: str, t) -> dict:
    flat = np.asarray(t, dtype=np.float64).reshape(-1)
    n = flat.size
    rng = np.random.Generator(np.random.PCG64(_seed(tag, name) ^ 0x5EED))
    idx = np.sort(rng.choice(n, size=min(NSAMP, n), replace=False)).astype(np.int64)
    return {f"{name}::n": np.int64(n), f"{name}::norm": np.float64(np.linalg.norm(flat)), f"{name}::idx": idx,
            f"{name}::val": flat[idx], f"{name}::proj": _project(tag, name, flat)}


def sketch_rel_err(g, tag: str, name: str, t):
    """(estimated relative L2 error of the whole tensor, relative L2 error on the sampled entries) of ``t`` against the sketch."""
    flat = np.asarray(t, dtype=np.float64).reshape(-1)
    assert flat.size == int(g[f"{name}::n"]), (name, flat.size, int(g[f"{name}::n"]))
    norm = max(float(g[f"{name}::norm"]), 1e-30)
    d = _project(tag, name, flat) - g[f"{name}::proj"]
    est = float(np.sqrt(np.mean(d * d)) / norm)
    idx, val = g[f"{name}::idx"], g[f"{name}::val"]
    samp = float(np.linalg.norm(flat[idx] - val) / max(np.linalg.norm(val), 1e-30))
    return est, samp


def load(tag: str):
    return np.load(os.path.join(GOLDEN, f"fullsize_{tag}.npz"))


def names(g):
    return sorted({k.split("::")[0] for k in g.files if "::" in k})


def check(label, tag, out, loss, grads, fwd_tol=1e-5, loss_tol=1e-5, grad_tol=None, grad_med_tol=None, fwd_min=None, noise_factor=8.0):
    """Forward, loss and every parameter gradient against the committed sketch of the fp64 oracle run.  A gradient passes inside
    max(grad_tol, noise_factor x the oracle's OWN fp32-vs-fp64 difference for that tensor) -- the second term is what
    tests/oracle_util.py::check_grads_at_rounding_level grants gradients that are sums with heavy cancellation (weight-norm gains, head
    biases), measured at generation time.  Returns the observed errors (printed: the margins are on record)."""
    g = load(tag)
    e_fwd, s_fwd = sketch_rel_err(g, tag, "out", out)
    e_loss = abs(float(loss) - float(g["loss"]))
    errs, over = {}, {}
    for n in grads:
        errs[n] = sketch_rel_err(g, tag, "grad/" + n, grads[n])[0]
        if grad_tol is not None:
            band = max(grad_tol, noise_factor * float(g[f"grad/{n}::noise"]))
            if errs[n] >= band:
                over[n] = (errs[n], band)
    worst = max(errs, key=errs.get)
    med = float(np.median(list(errs.values())))
    nlim = sum(1 for n in grads if grad_tol is not None and errs[n] >= grad_tol)
    print(f"[{label}] vs fp64-oracle sketch: forward rel-L2 {e_fwd:.2e} (sampled entries {s_fwd:.2e}), |loss diff| {e_loss:.2e}, "
          f"gradients: median {med:.2e}, worst {errs[worst]:.2e} ({worst}; the oracle's own fp32 noise there "
          f"{float(g['grad/' + worst + '::noise']):.1e}); {nlim} above {grad_tol} (held to {noise_factor} x their noise)")
    assert e_fwd < fwd_tol and s_fwd < 4 * fwd_tol, (e_fwd, s_fwd)
    if fwd_min is not None:
        assert e_fwd > fwd_min
    assert e_loss < loss_tol, e_loss
    assert not over, over
    if grad_med_tol is not None:
        assert med < grad_med_tol, med
    return e_fwd, med, errs[worst]
