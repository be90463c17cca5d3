"""Load the CPU-emulated build of the kernels (tests only) and wrap numpy arrays as pointers."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

_LIB = None


def emu_lib():
    global _LIB
    if _LIB is None:
        import build_emu
        from fourierflow_amd import _capi
        path = build_emu.build()
        lib = ctypes.CDLL(path)
        _capi.bind(lib)
        assert lib.ffno_build_target() == b"emu"
        _LIB = lib
    return _LIB


def ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def twiddle(L):
    tw = np.zeros(2 * L, np.float32)
    assert emu_lib().ffno_twiddle_fill_host(ptr(tw), L) == 0
    return tw


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
