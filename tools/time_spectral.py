#!/usr/bin/env python3
"""Micro-timing of the fused spectral branch kernel at the markov/24 shape (both axes, forward and adjoint settings)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fourierflow_amd import _lib  # noqa: E402
lib = _lib.get_lib()
B, M, N, C, K = 32, 64, 64, 64, 16
dev = torch.device("cuda:0")
x = torch.randn(B, M, N, C, device=dev)
out = torch.zeros(B, M, N, C, device=dev)
R = B * M
spec = torch.zeros(K * R * 2 * C, device=dev)
planes = torch.randn(2 * K * C * C, device=dev) * 0.05
host = np.zeros(2 * N, np.float32)
lib.ffno_twiddle_fill_host(host.ctypes.data_as(ctypes.c_void_p), N)
tw = torch.from_numpy(host).to(dev)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def timeit(name, fn, n=30):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:40s} {1e3 * a.elapsed_time(b) / n:8.1f} us", flush=True)


for axis in (0, 1):
    for acc in (0, 1):
        timeit(f"fused axis={axis} fwd save acc={acc}", lambda: lib.ffno_spectral_fused(p(x), p(out), None, p(spec), p(planes), p(tw), B, M, N, C, K, axis, 0, 1, 0, acc, None, None))
timeit("fused axis=0 inference (no save)", lambda: lib.ffno_spectral_fused(p(x), p(out), None, None, p(planes), p(tw), B, M, N, C, K, 0, 0, 1, 0, 0, None, None))
