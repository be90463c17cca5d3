"""Loader of the gfx950 C-ABI library (fourierflow_amd/lib/libffno_hip.so).

There is NO CPU fallback: if the HIP library cannot be loaded every operator raises.  (The tests can
install the CPU wave-emulator build of the *same kernel sources* through ``_install_test_backend``.  That
hook refuses to work unless the process was started with ``FFNO_ALLOW_TEST_BACKEND=1`` -- tests/conftest.py
sets it, nothing in the package does -- so no stray call can re-route the operators of a product run.)
"""
from __future__ import annotations

import ctypes
import os
import threading

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libffno_hip.so")

_lock = threading.Lock()
_lib = None
_test_backend = None


class FFNOLibraryError(RuntimeError):
    pass


def get_lib() -> ctypes.CDLL:
    """The bound C-ABI library; builds it with hipcc on first use if the .so is absent."""
    global _lib
    if _test_backend is not None:
        return _test_backend
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            from . import build as _build
            state = _build.staleness()
            if state == "unverifiable":
                # a prebuilt library without the sources it came from (installed deployment): nothing to compare it with --
                # load it, and say so
                import warnings
                warnings.warn(f"{LIB_PATH}: the kernel sources / include/ffno.h are not readable here, so the library cannot be "
                              f"checked against them; loading it as it is", RuntimeWarning)
            elif state == "stale":
                # absent, or built from different kernel sources / ABI header than the ones in this tree (a stale .so would
                # silently run old kernels behind new host code): rebuild (file-locked: all ranks of a node may get here at
                # once; the first one builds, the others find a fresh library), or refuse to load
                try:
                    _build.build(verbose=False)
                except Exception as e:  # noqa: BLE001
                    what = "is missing" if not os.path.exists(LIB_PATH) else "was built from other sources than this tree's"
                    raise FFNOLibraryError(
                        f"libffno_hip.so {what} and could not be rebuilt ({e}). Run "
                        f"`python -m fourierflow_amd.build` (needs hipcc, ROCm >= 7.0). "
                        f"There is no CPU fallback for the F-FNO operators.") from e
            # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as the
            # system one).  It MUST be loaded first so this library binds to the runtime that owns
            # torch's streams and allocations; two runtimes in one process = hipErrorNoDevice.
            import torch  # noqa: F401
            try:
                lib = ctypes.CDLL(LIB_PATH)
            except OSError as e:
                raise FFNOLibraryError(f"cannot load {LIB_PATH}: {e}") from e
            _capi.bind(lib)
            if lib.ffno_build_target() != b"gfx950":
                raise FFNOLibraryError(f"{LIB_PATH} is not a gfx950 build")
            check_abi(lib, LIB_PATH)
            _lib = lib
    return _lib


def check_abi(lib, path) -> None:
    """The struct layouts and argument lists of _capi.py belong to ONE generation of include/ffno.h: a library of another
    generation (a prebuilt .so loaded "unverifiable", an older build on the path) would read shifted arguments."""
    got = int(lib.ffno_abi_version())
    if got != _capi.ABI_VERSION:
        raise FFNOLibraryError(f"{path} speaks ABI generation {got}, this host code expects {_capi.ABI_VERSION} "
                               f"(include/ffno.h FFNO_ABI_VERSION): rebuild with `python -m fourierflow_amd.build --force`")


def is_test_backend() -> bool:
    return _test_backend is not None


def _install_test_backend(lib):
    """tests/ only: route the host code to the CPU wave-emulator build of the kernel sources."""
    global _test_backend
    if lib is not None and os.environ.get("FFNO_ALLOW_TEST_BACKEND") != "1":
        raise FFNOLibraryError("the emulator test backend can only be installed in a process started with "
                               "FFNO_ALLOW_TEST_BACKEND=1 (tests/conftest.py); product runs use libffno_hip.so only")
    if lib is not None:
        check_abi(lib, "the emulator test backend")
    _test_backend = lib


def current_stream(device) -> int:
    """hipStream_t of torch's current stream on ``device`` (0 for the emulator backend)."""
    if _test_backend is not None:
        return 0
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_device_tensor(t, what: str):
    import torch
    if _test_backend is not None:
        if t.is_cuda:
            raise FFNOLibraryError("emulator backend takes CPU tensors")
    elif not t.is_cuda:
        raise FFNOLibraryError(
            f"{what}: expected a tensor on an MI355X (cuda) device, got {t.device}. "
            f"The fourierflow_amd operators are HIP-only; there is no CPU path.")
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")
