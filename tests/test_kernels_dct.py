"""DCT branch of the CNOFactorized* operators (reference factorized_cno/grid_2d.py:51-96 + modules/dct.py) on the truncated
real-DFT kernels: forward against scipy's orthonormal DCT-II, the kept coefficients, residual / accumulate epilogue, the
adjoint configuration, and the real weight-gradient reduction."""
import numpy as np
import pytest
from scipy.fft import dct, idct

from backend_util import be, rel_l2  # noqa: F401

TOL = 1e-5


@pytest.mark.parametrize("B,M,N,C,K", [(2, 10, 12, 32, 5), (1, 9, 16, 64, 16), (2, 33, 7, 32, 7), (1, 40, 20, 64, 20)])
@pytest.mark.parametrize("axis", [0, 1])
def test_dct_branch(be, B, M, N, C, K, axis):
    L = N if axis == 0 else M
    if K > L:
        pytest.skip("modes exceed axis")
    lib, p = be.lib, be.ptr
    rs = np.random.RandomState(B + M + N + K + axis)
    x = rs.standard_normal((B, M, N, C)).astype(np.float32)
    w = (rs.standard_normal((C, C, K)) / 8).astype(np.float32)
    R = B * M if axis == 0 else B * N
    hx, hw, tw2 = be.put(x), be.put(w), be.twiddle(2 * L)
    wp, wpt = be.zeros((K, 2, C, C)), be.zeros((K, 2, C, C))
    assert lib.ffno_fw_pack_real(p(hw), p(wp), p(wpt), C, K, None) == 0
    np.testing.assert_array_equal(be.get(wp)[:, 0], w.transpose(2, 0, 1))
    assert not be.get(wp)[:, 1].any()
    out, spec, mix = be.empty(x.shape), be.empty((K, R, 2, C)), be.empty((K, R, 2, C))
    assert lib.ffno_dct_branch(p(hx), p(out), None, p(spec), p(mix), p(wp), p(tw2), B, M, N, C, K, axis, 0, 0, None) == 0
    ax = 2 if axis == 0 else 1
    X = dct(x.astype(np.float64), type=2, norm="ortho", axis=ax)
    Xk = np.take(X, range(K), axis=ax)
    Y = np.einsum("bmki,iok->bmko" if axis == 0 else "bkni,iok->bkno", Xk, w.astype(np.float64))
    pad = np.zeros_like(X)
    idx = [slice(None)] * 4
    idx[ax] = slice(0, K)
    pad[tuple(idx)] = Y
    ref = idct(pad, type=2, norm="ortho", axis=ax)
    assert rel_l2(be.get(out), ref) < TOL
    # kept coefficients: [k][line][re/im][c], imaginary parts exactly zero
    sp = be.get(spec)
    lines = Xk.transpose(2, 0, 1, 3).reshape(K, R, C) if axis == 0 else Xk.transpose(1, 0, 2, 3).reshape(K, R, C)
    assert rel_l2(sp[:, :, 0], lines) < TOL and not sp[:, :, 1].any()
    # adjoint configuration (transposed weights) with residual + accumulate
    resid = rs.standard_normal(x.shape).astype(np.float32)
    base = rs.standard_normal(x.shape).astype(np.float32)
    hres, acc = be.put(resid), be.put(base)
    assert lib.ffno_dct_branch(p(hx), p(acc), p(hres), p(spec), p(mix), p(wpt), p(tw2), B, M, N, C, K, axis, 1, 1, None) == 0
    Yt = np.einsum("bmko,iok->bmki" if axis == 0 else "bkno,iok->bkni", Xk, w.astype(np.float64))
    pad[tuple(idx)] = Yt
    assert rel_l2(be.get(acc), idct(pad, type=2, norm="ortho", axis=ax) + resid + base) < TOL
    # no weights: plain low-pass, `spec` still holds the coefficients afterwards
    assert lib.ffno_dct_branch(p(hx), p(out), None, p(spec), p(mix), None, p(tw2), B, M, N, C, K, axis, 0, 0, None) == 0
    pad[tuple(idx)] = Xk
    assert rel_l2(be.get(out), idct(pad, type=2, norm="ortho", axis=ax)) < TOL
    assert rel_l2(be.get(spec)[:, :, 0], lines) < TOL


def test_real_weight_gradient_reduce(be):
    lib, p = be.lib, be.ptr
    C, K, nsplit = 32, 3, 2
    rs = np.random.RandomState(1)
    part = rs.standard_normal((nsplit, K, 2, C, C)).astype(np.float32)
    g0 = rs.standard_normal((C, C, K)).astype(np.float32)
    hp, hg = be.put(part), be.put(g0)
    assert lib.ffno_fw_grad_reduce_real(p(hp), p(hg), C, K, nsplit, 1, None) == 0
    assert rel_l2(be.get(hg), g0 + part.sum(0)[:, 0].transpose(1, 2, 0)) < 1e-6
    assert lib.ffno_dct_branch(None, p(hg), None, p(hg), p(hg), None, p(hg), 1, 4, 4, 32, 2, 0, 0, 0, None) == -1
    assert lib.ffno_dct_branch(p(hp), p(hg), None, p(hg), p(hp), None, p(hg), 1, 4, 4, 32, 5, 0, 0, 0, None) == -3
