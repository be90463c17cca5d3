#!/usr/bin/env python3
"""Time the feed-forward chain launches (ffno_ffh_fwd2 / ffno_ffh_bwd_data2: two addends, no stored sum, residual, sign words) at the
headline shape for one or more builds of the library.   python tools/time_chain.py [lib.so ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourierflow_amd import _capi  # noqa: E402
from fourierflow_amd._capi import FfOpts, FxPackDesc  # noqa: E402

libs = sys.argv[1:] or [os.path.join(ROOT, "fourierflow_amd", "lib", "libffno_hip.so")]
P, C, H = int(os.environ.get("FF_P", 131072)), 64, 256
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
W1 = (torch.randn(H, C, generator=g) / 8).to(dev)
W2 = (torch.randn(C, H, generator=g) / 16).to(dev)
b1 = (torch.randn(H, generator=g) * 0.1).to(dev)
b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
NSET = 6      # rotate over buffer sets (0.8 GB): HBM / Infinity Cache, not L2
sets = [[torch.randn(P, C, generator=g).to(dev) * 0.5 for _ in range(4)] for _ in range(NSET)]
outs = [torch.empty(P, C, device=dev) for _ in range(NSET)]
ref = None
for path in libs:
    lib = ctypes.CDLL(path)
    for fn_name in ("ffno_ffh_pack_bytes", "ffno_ffh_pack", "ffno_amax", "ffno_ff_mask_words", "ffno_ffh_fwd2", "ffno_ffh_bwd_data2"):
        res, args = _capi.SIGNATURES[fn_name]
        getattr(lib, fn_name).restype, getattr(lib, fn_name).argtypes = res, args
    nb = lib.ffno_ffh_pack_bytes(C, H)
    bufs = [torch.zeros(nb // 4, dtype=torch.int32, device=dev) for _ in range(4)]
    spec = [(W1, C, 1, 1), (W2, 1, H, 2), (W2, 1, H, 1), (W1, C, 1, 2)]
    descs = (FxPackDesc * 4)(*[FxPackDesc(p(a), p(b), sh, sc, ty, 0) for (a, sh, sc, ty), b in zip(spec, bufs)])
    table = torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()).to(dev)
    assert lib.ffno_ffh_pack(p(table), 4, C, H, None) == 0
    words = torch.zeros(4, dtype=torch.int32, device=dev)
    for t in sets[0][:2]:
        assert lib.ffno_amax(p(t), t.numel(), ctypes.c_void_p(words.data_ptr()), None) == 0
    mask = torch.zeros(int(lib.ffno_ff_mask_words(P, H)), dtype=torch.int32, device=dev)
    of = FfOpts(ctypes.c_void_p(words.data_ptr()), ctypes.c_void_p(words.data_ptr() + 4), 0, 0, 0)
    ob = FfOpts(ctypes.c_void_p(words.data_ptr()), ctypes.c_void_p(words.data_ptr() + 8), 0, 0, 0)

    def fwd(k):
        a, b, r, _ = sets[k % NSET]
        return lib.ffno_ffh_fwd2(p(a), p(b), None, p(r), p(bufs[0]), p(b1), p(bufs[1]), p(b2), p(outs[k % NSET]), p(mask), P, C, H, ctypes.byref(of), None)

    def bwd(k):
        a, b, _, _ = sets[k % NSET]
        return lib.ffno_ffh_bwd_data2(p(a), p(b), p(sets[k % NSET][3]), p(mask), p(bufs[2]), p(bufs[3]), p(outs[k % NSET]), P, C, H, ctypes.byref(ob), None)

    res = []
    for name, fn in (("fwd2", fwd), ("bwd_data2", bwd)):
        for k in range(NSET):
            assert fn(k) == 0
        torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 60
        a_.record()
        for k in range(n):
            fn(k)
        b_.record()
        torch.cuda.synchronize()
        res.append((name, 1e3 * a_.elapsed_time(b_) / n))
    fwd(0)
    out = outs[0].cpu().numpy().copy()
    if ref is None:
        ref, note = out, "(reference)"
    else:
        note = "bit-identical to the first" if np.array_equal(out, ref) else "DIFFERENT from the first"
    print(f"{os.path.basename(path):32s} " + "  ".join(f"{n_} {us:6.1f} us" for n_, us in res) + f"   {note}", flush=True)
