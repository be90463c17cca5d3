"""tests/fullsize_util.py (sketches of full-size oracle runs: VERDICT r05 #7): the estimator is what it claims to be, and the
committed fixtures are complete."""
import numpy as np
import pytest

import fullsize_util as fu

TAGS = ["markov24_b32", "markov24_b19", "kochkov256_12l_k32", "kochkov256_24l_k64", "mesh3d_cfg5", "plasticity", "airfoil"]


def test_sketch_estimates_the_full_tensor_error():
    rs = np.random.RandomState(0)
    t = rs.standard_normal(300_000)
    g = fu.make_sketch("unit", "t", t)
    assert fu.sketch_rel_err(g, "unit", "t", t) == (0.0, 0.0)
    for rel in (1e-7, 1e-5, 1e-3):
        noisy = t + rel * np.linalg.norm(t) / np.sqrt(t.size) * rs.standard_normal(t.size)
        est, samp = fu.sketch_rel_err(g, "unit", "t", noisy)
        assert 0.5 * rel < est < 2.0 * rel and 0.5 * rel < samp < 2.0 * rel, (rel, est, samp)
    # an error confined to a few entries (which the 256 samples will not hit) is still seen by the projections
    spiky = t.copy()
    spiky[12345] += 1e-2 * np.linalg.norm(t)
    est, _ = fu.sketch_rel_err(g, "unit", "t", spiky)
    assert 0.5e-2 < est < 2e-2
    # the sign vectors are a pure function of (tag, name, n)
    np.testing.assert_array_equal(fu._signs("unit", "t", 1000), fu._signs("unit", "t", 1000))
    assert not np.array_equal(fu._signs("unit", "t", 1000), fu._signs("unit", "u", 1000))


@pytest.mark.parametrize("tag", TAGS)
def test_committed_fullsize_fixture_is_complete(tag):
    g = fu.load(tag)
    names = fu.names(g)
    assert "out" in names and 0.5 < float(g["loss"]) < 2.0 and str(g["dtype"]) == "float64"
    grads = [n for n in names if n.startswith("grad/")]
    assert len(grads) >= 20
    for n in names:
        assert int(g[f"{n}::n"]) > 0 and np.isfinite(g[f"{n}::proj"]).all() and g[f"{n}::proj"].shape == (fu.NPROJ,)
        assert np.isfinite(float(g[f"{n}::norm"]))
