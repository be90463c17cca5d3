"""Two ways to run the SAME kernel sources from the tests through the SAME C ABI:

  "emu": CPU wave-emulator build (tests/emu) on numpy arrays            -> runs with -m "not gpu"
  "gpu": the shipped gfx950 library on torch.cuda tensors of an MI355X   -> runs with -m gpu

``Backend.put`` uploads a numpy array, ``ptr`` gives the raw pointer for the ctypes call and ``get``
downloads the result, so one test body serves both.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

_EMU = None

BACKENDS = [pytest.param("emu", marks=pytest.mark.emu), pytest.param("gpu", marks=pytest.mark.gpu)]


def emu_lib():
    global _EMU
    if _EMU is None:
        import build_emu
        from fourierflow_amd import _capi
        lib = ctypes.CDLL(build_emu.build())
        _capi.bind(lib)
        assert lib.ffno_build_target() == b"emu"
        _EMU = lib
    return _EMU


class Backend:
    def __init__(self, kind):
        self.kind = kind
        if kind == "emu":
            self.lib = emu_lib()
            self.device = "cpu"
        else:
            import torch
            from fourierflow_amd import _lib
            assert torch.cuda.is_available(), "gpu backend needs an MI355X"
            self.lib = _lib.get_lib()
            self.device = "cuda:0"

    def put(self, a):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        if self.kind == "emu":
            return a.copy()
        import torch
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        t = torch.from_numpy(a.copy()).to(self.device)
        # every device operand lives as long as the backend object (one test): a call like  f(p(be.put(a)), p(be.put(b)))  hands
        # raw pointers to an asynchronous launch, and a temporary freed between the two puts would give its block to the second
        self.__dict__.setdefault("_keep", []).append(t)
        return t

    def empty(self, shape, dtype=np.float32):
        a = np.empty(shape, dtype)
        a.fill(np.nan if np.issubdtype(dtype, np.floating) else 0)
        return self.put(a)

    def zeros(self, shape, dtype=np.float32):
        return self.put(np.zeros(shape, dtype))

    def ptr(self, h):
        if h is None:
            return None
        if self.kind == "emu":
            return h.ctypes.data_as(ctypes.c_void_p)
        return ctypes.c_void_p(h.data_ptr())

    def get(self, h):
        if self.kind == "emu":
            return h
        import torch
        torch.cuda.synchronize()
        return h.cpu().numpy()

    def twiddle(self, L):
        tw = np.zeros(2 * L, np.float32)
        assert self.lib.ffno_twiddle_fill_host(tw.ctypes.data_as(ctypes.c_void_p), L) == 0
        return self.put(tw)


@pytest.fixture(params=BACKENDS)
def be(request):
    return Backend(request.param)


@pytest.fixture(params=BACKENDS)
def host_device(request):
    """For tests of the Python host (engine / modules): yields the torch device to use."""
    from fourierflow_amd import _lib
    if request.param == "emu":
        _lib._install_test_backend(emu_lib())
        yield "cpu"
        _lib._install_test_backend(None)
    else:
        import torch
        assert torch.cuda.is_available()
        yield "cuda:0"


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
