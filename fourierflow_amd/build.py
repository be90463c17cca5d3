"""Build the gfx950 C-ABI library (hipcc) in-tree: fourierflow_amd/lib/libffno_hip.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
.so is git-ignored but travels to the GPU box with the working tree.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libffno_hip.so")
SOURCES = ["spectral.hip", "spectral_x3.hip", "ff.hip", "ffx.hip", "pointwise.hip", "velocity.hip", "spectral2d.hip", "plin.hip", "layer.hip", "layernorm.hip", "glin.hip", "infer.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 F-FNO kernels)")


def source_files():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stamp() -> str:
    h = hashlib.sha256()
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    for p in sorted(source_files() + headers + [os.path.join(ROOT, "include", "ffno.h")]):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def staleness() -> str:
    """"fresh": the library on disk was built from exactly this tree's sources; "stale": from other sources (stamp mismatch / no
    stamp / no library); "unverifiable": the sources or the ABI header are not readable here (an installed deployment that
    ships the prebuilt .so without csrc/ or include/) -- nothing to compare the stamp with."""
    stamp_file = LIB + ".stamp"
    if not os.path.exists(LIB):
        return "stale"
    try:
        want = _stamp()
    except OSError:
        return "unverifiable"
    try:
        with open(stamp_file) as f:
            return "fresh" if f.read() == want else "stale"
    except OSError:
        return "stale"


def is_stale() -> bool:
    """True when the library on disk is known to come from other sources than the ones in this tree."""
    return staleness() == "stale"


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    """Compile + link under an exclusive file lock (the ranks of a data-parallel job may all find the library stale at once),
    into temporary names that are renamed into place: a concurrent reader never sees a half-written .so."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = LIB + ".stamp"
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            stamp = _stamp()
            # (checked AFTER the lock was taken: another process may have finished the same build while we waited)
            if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
                return LIB
            tag = f".tmp{os.getpid()}"
            # objects are rebuilt only when their own source, a header or the flags changed (per-object stamps), and the
            # translation units compile side by side: a kernel edit costs one file's compile time, not eleven
            from concurrent.futures import ThreadPoolExecutor
            hdr = hashlib.sha256()
            for p in sorted([os.path.join(CSRC, x) for x in os.listdir(CSRC) if x.endswith(".h")] + [os.path.join(ROOT, "include", "ffno.h")]):
                with open(p, "rb") as f:
                    hdr.update(f.read())
            hdr.update(repr(tuple(extra_flags)).encode())

            def compile_one(src):
                obj = os.path.join(LIBDIR, os.path.basename(src) + ".o")
                with open(src, "rb") as f:
                    want = hashlib.sha256(hdr.digest() + f.read()).hexdigest()
                try:
                    if not force and os.path.exists(obj) and open(obj + ".stamp").read() == want:
                        return obj
                except OSError:
                    pass
                cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src,
                       "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", obj, "-Wno-unused-result", *extra_flags]
                if verbose:
                    print("[ffno build]", " ".join(cmd), flush=True)
                if os.path.exists(obj + ".stamp"):
                    os.remove(obj + ".stamp")
                subprocess.check_call(cmd)
                with open(obj + ".stamp", "w") as f:
                    f.write(want)
                return obj

            with ThreadPoolExecutor(max_workers=max(1, min(len(source_files()), (os.cpu_count() or 4)))) as ex:
                objs = list(ex.map(compile_one, source_files()))
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + tag, *objs]
            if verbose:
                print("[ffno build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            if os.path.exists(stamp_file):
                os.remove(stamp_file)          # never a new library beside an old stamp (or the reverse)
            os.replace(LIB + tag, LIB)
            with open(stamp_file + tag, "w") as f:
                f.write(stamp)
            os.replace(stamp_file + tag, stamp_file)
            return LIB
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
