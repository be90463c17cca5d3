"""General pointwise-linear kernels (fourierflow_amd/csrc/glin.hip: ffno_glin_*, ffno_dropout*) through the C ABI vs fp64 numpy
-- the feed-forward shapes the fused chain kernels do not take (n_ff_layers != 2, dropout > 0, in_dropout > 0; reference
fourierflow/modules/feedforward.py:6-24, grid_2d.py:113,158).  CPU wave emulator (-m "not gpu") and MI355X (-m gpu)."""
import numpy as np
import pytest

from backend_util import be, rel_l2  # noqa: F401

TOL = 1e-5


def keep_mask(be, shape, p, seed):
    m = be.zeros(int(np.prod(shape)), np.uint8)
    assert be.lib.ffno_dropout_mask(be.ptr(m), m.size if be.kind == "emu" else m.numel(), p, seed, None) == 0
    return np.asarray(be.get(m)).reshape(shape).astype(np.float64)


@pytest.mark.parametrize("P,Cin,Cout", [(70, 64, 128), (130, 128, 128), (65, 128, 64), (33, 32, 96), (5000, 256, 256), (257, 256, 64)])
@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_glin_fwd_bwd(be, P, Cin, Cout, p_drop):
    if be.kind == "emu" and P > 1000:
        pytest.skip("large case runs on the GPU only")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(P + Cin + Cout)
    x = rs.standard_normal((P, Cin)).astype(np.float32)
    W = (rs.standard_normal((Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
    b = (rs.standard_normal(Cout) * 0.1).astype(np.float32)
    resid = rs.standard_normal((P, Cout)).astype(np.float32)
    g = rs.standard_normal((P, Cout)).astype(np.float32)
    seed = 1234 + P
    dx_, dW_, db_ = be.put(x), be.put(W), be.put(b)
    keep = keep_mask(be, (P, Cout), p_drop, seed) if p_drop else np.ones((P, Cout))
    if p_drop:
        assert abs(keep.mean() - (1 - p_drop)) < 4 * np.sqrt(p_drop * (1 - p_drop) / keep.size) + 1e-3      # a fair coin of bias p
    sc = 1.0 / (1.0 - p_drop)
    pre = x.astype(np.float64) @ W.astype(np.float64).T + b
    # hidden layer: Linear -> Dropout -> ReLU
    y = be.empty((P, Cout))
    assert lib.ffno_glin_fwd(p(dx_), p(dW_), p(db_), None, p(y), P, Cin, Cout, 1, p_drop, seed, None) == 0
    ref_y = np.maximum(pre, 0) * keep * sc
    assert rel_l2(be.get(y), ref_y) < TOL
    # last layer: Linear -> Dropout, + residual
    out = be.empty((P, Cout))
    assert lib.ffno_glin_fwd(p(dx_), p(dW_), p(db_), p(be.put(resid)), p(out), P, Cin, Cout, 0, p_drop, seed, None) == 0
    assert rel_l2(be.get(out), pre * keep * sc + resid) < TOL
    # backward of the hidden layer (gate = [y > 0]: contains the dropped units) and of the last layer (gate = keep)
    dg = be.put(g)
    for relu in (1, 0):
        dpre = g.astype(np.float64) * sc * ((ref_y > 0) if relu else keep)
        dx = be.put(np.ones((P, Cin), np.float32))
        yb = y if relu else None
        assert lib.ffno_glin_bwd_data(p(dg), p(yb), p(dW_), p(dx), P, Cin, Cout, p_drop, seed, 1, None) == 0      # accumulate onto ones
        assert rel_l2(be.get(dx), dpre @ W.astype(np.float64) + 1.0) < TOL
        part = be.zeros(lib.ffno_glin_wgrad_partial_floats(P, Cin, Cout))
        gW, gb = be.zeros((Cout, Cin)), be.zeros(Cout)
        for acc in (0, 1):
            assert lib.ffno_glin_bwd_weights(p(dg), p(yb), p(dx_), p(part), p(gW), p(gb), P, Cin, Cout, p_drop, seed, acc, None) == 0
        assert rel_l2(be.get(gW), 2 * dpre.T @ x.astype(np.float64)) < TOL
        assert rel_l2(be.get(gb), 2 * dpre.sum(0)) < TOL


def test_dropout_in_place_and_argument_checks(be):
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(0)
    x = rs.standard_normal(5000).astype(np.float32)
    dx = be.put(x)
    assert lib.ffno_dropout(p(dx), x.size, 0.25, 77, None) == 0
    keep = keep_mask(be, (5000,), 0.25, 77)
    np.testing.assert_allclose(be.get(dx), x * keep / 0.75, rtol=1e-6)
    assert not np.array_equal(keep, keep_mask(be, (5000,), 0.25, 78))           # another seed, another mask
    assert lib.ffno_dropout(p(dx), x.size, 0.0, 1, None) == 0                     # p = 0: untouched
    assert lib.ffno_dropout(p(dx), x.size, 1.0, 1, None) == -1 and lib.ffno_dropout(None, 4, 0.1, 1, None) == -1
    assert lib.ffno_glin_supported(256, 256) == 1 and lib.ffno_glin_supported(257, 64) == 0
    assert lib.ffno_glin_fwd(p(dx), p(dx), None, None, p(dx), 4, 300, 64, 0, 0.0, 0, None) == -2
