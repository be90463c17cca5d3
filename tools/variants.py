#!/usr/bin/env python3
"""Build VARIANTS of the library for A/B timing on the GPU box (cross-compiled here, shipped with the snapshot under
tools/scratch/variants/, git-ignored): only the named translation units are recompiled with the extra flags, the rest of the objects
are the tree's.   python tools/variants.py NAME ffx.hip -DFOO=1 ...   ->  tools/scratch/variants/libffno_NAME.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourierflow_amd import build as B  # noqa: E402

name, rest = sys.argv[1], sys.argv[2:]
units = [a for a in rest if a.endswith(".hip")]
flags = [a for a in rest if not a.endswith(".hip")]
B.build(verbose=False)
out = os.path.join(ROOT, "tools", "scratch", "variants")
os.makedirs(out, exist_ok=True)
objs = []
for src in B.source_files():
    base = os.path.basename(src)
    if base in units:
        obj = os.path.join(out, f"{name}_{base}.o")
        subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-I", B.CSRC,
                               "-I", os.path.join(ROOT, "include"), "-o", obj, "-Wno-unused-result", *flags])
    else:
        obj = os.path.join(B.LIBDIR, base + ".o")
    objs.append(obj)
lib = os.path.join(out, f"libffno_{name}.so")
subprocess.check_call([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib, *objs])
print(lib)
