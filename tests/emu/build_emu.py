"""Build the CPU wave-emulator flavour of the kernels (TEST INFRASTRUCTURE).

Compiles the *same* sources as the gfx950 library with -DFFNO_EMU against tests/emu/hip_emu.h and
writes tests/emu/_build/libffno_emu.so.  Only tests load it; the package never does.
"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "fourierflow_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libffno_emu.so")
SOURCES = ["spectral.hip", "spectral_x3.hip", "ff.hip", "ffx.hip", "pointwise.hip", "velocity.hip", "spectral2d.hip", "plin.hip", "layer.hip", "layernorm.hip", "glin.hip", "infer.hip"]


def _cxx():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("clang++ (ext_vector_type support) not found for the emulator build")


def build(verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    h = hashlib.sha256()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    for p in sorted(srcs + headers + [os.path.join(ROOT, "include", "ffno.h"), os.path.abspath(__file__)]):      # (+ the flags)
        h.update(open(p, "rb").read())
    stamp = h.hexdigest()
    sf = LIB + ".stamp"
    if os.path.exists(LIB) and os.path.exists(sf) and open(sf).read() == stamp:
        return LIB
    # pytest-xdist workers all get here at once after a source change: one builds (exclusive file lock, temporary names renamed
    # into place), the others wait and find a fresh library -- nobody ever loads a half-linked .so
    import fcntl
    with open(os.path.join(OUT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(LIB) and os.path.exists(sf) and open(sf).read() == stamp:
                return LIB
            tag = f".tmp{os.getpid()}"
            objs = []
            for src in srcs:
                obj = os.path.join(OUT, os.path.basename(src) + ".o")
                cmd = [_cxx(), "-x", "c++", "-std=c++17", "-O2", "-mf16c", "-fPIC", "-I", HERE, "-I", CSRC,
                       "-I", os.path.join(ROOT, "include"), "-Wno-unknown-pragmas", "-Wno-unknown-attributes", "-Wno-psabi", "-c", src,
                       "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                objs.append(obj)
            subprocess.check_call([_cxx(), "-shared", "-fPIC", "-o", LIB + tag, *objs])
            if os.path.exists(sf):
                os.remove(sf)
            os.replace(LIB + tag, LIB)
            with open(sf + tag, "w") as f:
                f.write(stamp)
            os.replace(sf + tag, sf)
            return LIB
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(verbose=True))
