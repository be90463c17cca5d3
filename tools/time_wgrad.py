#!/usr/bin/env python3
"""Time the all-layers feed-forward weight-gradient launch (ffno_ffh_bwd_weights_partial_multi) at the headline shape -- 24 layers,
P = 131072 pixels, C = 64, H = 256, 32 slices per layer, s as two addends -- for one or more builds of the library, and compare
each build's slices with the first one's.   python tools/time_wgrad.py [lib.so ...]   (default: the tree's library)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourierflow_amd import _capi  # noqa: E402
from fourierflow_amd._capi import FfWgDesc, FxPackDesc  # noqa: E402

libs = sys.argv[1:] or [os.path.join(ROOT, "fourierflow_amd", "lib", "libffno_hip.so")]
L, P, C, H, NS = 24, int(os.environ.get("WG_P", 131072)), 64, 256, int(os.environ.get("WG_NS", 32))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
W1 = (torch.randn(H, C, generator=g) / 8).to(dev)
W2 = (torch.randn(C, H, generator=g) / 16).to(dev)
b1 = (torch.randn(H, generator=g) * 0.1).to(dev)
sa = [torch.randn(P, C, generator=g).to(dev) * 0.5 for _ in range(2)]
sb = [torch.randn(P, C, generator=g).to(dev) * 0.5 for _ in range(2)]
gg = [torch.randn(P, C, generator=g).to(dev) * 1e-3 for _ in range(2)]
if os.environ.get("WG_ZERO") == "1":      # all-zero activations and gradients: the same instruction stream on operands that toggle nothing
    for t in sa + sb + gg:
        t.zero_()
    sa[0][0, 0] = 1.0
    gg[0][0, 0] = 1.0      # (non-zero range words)
ref = None
for path in libs:
    lib = ctypes.CDLL(path)
    for fn_name in ("ffno_ffh_pack_bytes", "ffno_ffh_pack", "ffno_amax", "ffno_ff_wgrad_partial_floats", "ffno_ffh_bwd_weights_partial_multi"):
        res, args = _capi.SIGNATURES[fn_name]      # (older builds of the library lack newer entry points: bind what is used)
        getattr(lib, fn_name).restype, getattr(lib, fn_name).argtypes = res, args
    nb = lib.ffno_ffh_pack_bytes(C, H)
    bufs = [torch.zeros(nb // 4, dtype=torch.int32, device=dev) for _ in range(4)]
    spec = [(W1, C, 1, 1), (W2, 1, H, 2), (W2, 1, H, 1), (W1, C, 1, 2)]
    descs = (FxPackDesc * 4)(*[FxPackDesc(p(a), p(b), sh, sc, ty, 0) for (a, sh, sc, ty), b in zip(spec, bufs)])
    table = torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()).to(dev)
    assert lib.ffno_ffh_pack(p(table), 4, C, H, None) == 0
    words = torch.zeros(2, dtype=torch.int32, device=dev)
    for t in sa + sb:
        assert lib.ffno_amax(p(t), t.numel(), ctypes.c_void_p(words.data_ptr()), None) == 0
    for t in gg:
        assert lib.ffno_amax(p(t), t.numel(), ctypes.c_void_p(words.data_ptr() + 4), None) == 0
    nfl = int(lib.ffno_ff_wgrad_partial_floats(C, H, NS))
    parts = torch.zeros(L, nfl, device=dev)
    jobs = [FfWgDesc(p(sa[l & 1]), p(gg[l & 1]), p(bufs[0]), p(b1), p(bufs[2]), ctypes.c_void_p(parts[l].data_ptr()),
                     ctypes.c_void_p(words.data_ptr()), ctypes.c_void_p(words.data_ptr() + 4), p(sb[l & 1]), None) for l in range(L)]
    tab = (FfWgDesc * L)(*jobs)
    fn = lambda: lib.ffno_ffh_bwd_weights_partial_multi(tab, L, P, C, H, NS, 0, 1, None)  # noqa: E731
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = 1e3 * a.elapsed_time(b) / n
    out = parts[:2].cpu().numpy().copy()
    if ref is None:
        ref, note = out, "(reference)"
    else:
        same = np.array_equal(out, ref)
        note = "bit-identical to the first" if same else "rel-L2 vs first %.2e" % (np.linalg.norm(out - ref) / np.linalg.norm(ref))
    print(f"{os.path.basename(path):40s} {us:9.1f} us   {note}", flush=True)
