"""The level-1 operator entry point (include/ffno.h: ffno_spectral2d_fwd / _bwd through fourierflow_amd.ops.spectral_conv2d) at the
headline shape [32, 64, 64, 64], 16 modes, against the engine's paired fused launch on the same operands.
    python tools/time_spectral2d.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from backend_util import Backend  # noqa: E402
from test_infer_layer import _setup  # noqa: E402
from fourierflow_amd import _lib, ops  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    B, M, N, C, K = 32, 64, 64, 64, 16
    be = Backend("gpu")
    lib, p = be.lib, be.ptr
    S = _setup(be, B, M, N, K, seed=1)
    x = S["dx"]
    w0, w1 = torch.from_numpy(S["w"][0]).cuda(), torch.from_numpy(S["w"][1]).cuda()
    assert lib.ffno_spectral2d_path(B, M, N, C, K) == 1
    with torch.no_grad():
        y = ops.spectral_conv2d(x, w0, w1, K)
    s_img, t_img = be.empty(S["x"].shape), be.empty(S["x"].shape)
    sword = be.zeros(1, np.uint32)
    a2, b2 = S["branch"](0, s_img, p(sword)), S["branch"](1, t_img, p(sword))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.ffno_spectral_x3_pair(ctypes.byref(a2), ctypes.byref(b2), C, 0, 1, 0, 2, st) == 0
    torch.cuda.synchronize()
    ref = s_img + t_img
    print("level-1 operator vs paired launch (sum of its two outputs) rel-L2:", float((y - ref).norm() / ref.norm()))
    with torch.no_grad():
        t_op = timeit(lambda: ops.spectral_conv2d(x, w0, w1, K))
    # the C entry point itself (no autograd.Function, no workspace lookup): what a non-Python host pays
    wsd = ops._spectral2d_workspace(x, w0, w1, K)
    twn_, twm_ = ops._twiddle(N, x.device), ops._twiddle(M, x.device)
    yd = torch.empty_like(x)
    P_ = ctypes.c_void_p
    t_c = timeit(lambda: lib.ffno_spectral2d_fwd(P_(x.data_ptr()), P_(w0.data_ptr()), P_(w1.data_ptr()), P_(yd.data_ptr()), P_(wsd.data_ptr()),
                                                 P_(twn_.data_ptr()), P_(twm_.data_ptr()), B, M, N, C, K, 0, st))
    lib.ffno_spectral2d_weights_version(P_(wsd.data_ptr()), 0)
    t_c_repack = timeit(lambda: lib.ffno_spectral2d_fwd(P_(x.data_ptr()), P_(w0.data_ptr()), P_(w1.data_ptr()), P_(yd.data_ptr()), P_(wsd.data_ptr()),
                                                        P_(twn_.data_ptr()), P_(twm_.data_ptr()), B, M, N, C, K, 0, st))
    t_pair = timeit(lambda: lib.ffno_spectral_x3_pair(ctypes.byref(a2), ctypes.byref(b2), C, 0, 1, 0, 2, st))
    # ... and what the same call cost before round 6: the stage sequence (forced by going through the C entry points directly)
    ws = torch.empty(int(lib.ffno_spectral2d_ws_floats(B, M, N, C, K)), dtype=torch.float32, device="cuda")
    twn, twm = S["keep"][2]
    sa, sb = ws[:K * B * M * 2 * C], ws[K * B * M * 2 * C:2 * K * B * M * 2 * C]
    wp = torch.empty(2 * K * C * C, device="cuda")
    wpt = torch.empty_like(wp)
    out = torch.empty_like(x)
    P = ctypes.c_void_p

    def stages():
        for axis, (w, tw) in enumerate(((w0, twn), (w1, twm))):
            lib.ffno_dft_fwd(P(x.data_ptr()), P(sa.data_ptr()), p(tw), B, M, N, C, K, axis, 0, st)
            lib.ffno_fw_pack(P(w.data_ptr()), P(wp.data_ptr()), P(wpt.data_ptr()), C, K, st)
            lib.ffno_mode_mix(P(sa.data_ptr()), P(wp.data_ptr()), P(sb.data_ptr()), B * (M if axis == 0 else N), C, K, 0, st)
            lib.ffno_dft_inv(P(sb.data_ptr()), P(out.data_ptr()), None, p(tw), B, M, N, C, K, axis, 1, int(axis == 1), st)
    t_st = timeit(stages, 50)
    print(f"ops.spectral_conv2d (ffno_spectral2d_fwd: amax + one fused launch per axis)   {t_op:8.2f} us")
    print(f"ffno_spectral2d_fwd from C (declared weight version: packs reused)             {t_c:8.2f} us")
    print(f"ffno_spectral2d_fwd from C (no version declared: re-packs on every call)       {t_c_repack:8.2f} us")
    print(f"engine's paired fused launch (two branch images, no sum)                     {t_pair:8.2f} us   ratio {t_op / t_pair:.2f}")
    print(f"round-5 level-1 path: dft_fwd -> fw_pack -> mode_mix -> dft_inv per axis       {t_st:8.2f} us")


if __name__ == "__main__":
    main()
