"""The gfx950 shared library builds (hipcc cross-compiles without a GPU), loads, and exports every
symbol that include/ffno.h declares; the ctypes table covers the header exactly. No compute calls."""
import ctypes
import os
import re

import torch  # noqa: F401  (its bundled HIP runtime must be the one our library binds to)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "ffno.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(ffno_\w+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    from fourierflow_amd import _capi, build
    path = build.build(verbose=False)
    lib = ctypes.CDLL(path)
    bound = set(_capi.bind(lib))           # AttributeError here == missing export
    declared = header_functions()
    assert declared == bound, (declared - bound, bound - declared)
    assert lib.ffno_build_target() == b"gfx950"
    m = re.search(r"#define\s+FFNO_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "ffno.h")).read())
    assert lib.ffno_abi_version() == int(m.group(1)) == _capi.ABI_VERSION


def test_loader_refuses_a_library_of_another_abi_generation():
    import pytest
    from fourierflow_amd import _capi, _lib

    class Old:
        @staticmethod
        def ffno_abi_version():
            return _capi.ABI_VERSION - 1

    with pytest.raises(_lib.FFNOLibraryError, match="ABI generation"):
        _lib.check_abi(Old, "old.so")


def test_wn_desc_layout_matches_header():
    from fourierflow_amd._capi import WnDesc
    assert ctypes.sizeof(WnDesc) == 6 * 8 + 2 * 4
    assert [f[0] for f in WnDesc._fields_] == ["g", "v", "w", "dw", "dg", "dv", "rows", "cols"]


def test_host_twiddle_table():
    import numpy as np
    from fourierflow_amd import _capi, build
    lib = ctypes.CDLL(build.build(verbose=False))
    _capi.bind(lib)
    for L in (64, 12, 109):
        tw = np.zeros(2 * L, np.float32)
        assert lib.ffno_twiddle_fill_host(tw.ctypes.data_as(ctypes.c_void_p), L) == 0
        j = np.arange(L)
        np.testing.assert_allclose(tw[:L], np.cos(2 * np.pi * j / L) / np.sqrt(L), atol=1e-7)
        np.testing.assert_allclose(tw[L:], np.sin(2 * np.pi * j / L) / np.sqrt(L), atol=1e-7)
        assert tw[L] == 0.0
    assert lib.ffno_twiddle_fill_host(None, 8) == -1
