#!/usr/bin/env python3
"""Micro-timing of the feed-forward kernels (fp32-MFMA ff.hip vs bf16x3 ffx.hip) at the markov/24 shape."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fourierflow_amd import _lib  # noqa: E402
from fourierflow_amd._capi import FxPackDesc  # noqa: E402

lib = _lib.get_lib()
P, C, H = int(os.environ.get("FF_P", 131072)), 64, 256
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
s = torch.randn(P, C, generator=g).to(dev)
resid = torch.randn(P, C, generator=g).to(dev)
db = torch.randn(P, C, generator=g).to(dev)
W1 = (torch.randn(H, C, generator=g) / 8).to(dev)
W2 = (torch.randn(C, H, generator=g) / 16).to(dev)
b1 = (torch.randn(H, generator=g) * 0.1).to(dev)
b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
W1t, W2t = W1.t().contiguous(), W2.t().contiguous()
out, ds = torch.empty(P, C, device=dev), torch.empty(P, C, device=dev)
h, dh = torch.empty(P, H, device=dev), torch.empty(P, H, device=dev)
mask = torch.zeros(lib.ffno_ff_mask_words(P, H), dtype=torch.int32, device=dev)
nsplit = 256
partial = torch.zeros(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit), device=dev)
gW1, gW2, gb1, gb2 = torch.zeros(H, C, device=dev), torch.zeros(C, H, device=dev), torch.zeros(H, device=dev), torch.zeros(C, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
nb = lib.ffno_ffx_pack_bytes(C, H)
bufs = [torch.zeros(nb // 4, dtype=torch.int32, device=dev) for _ in range(4)]
spec = [(W1, C, 1, 1), (W2, 1, H, 2), (W2, 1, H, 1), (W1, C, 1, 2)]
descs = (FxPackDesc * 4)(*[FxPackDesc(p(a), p(b), sh, sc, ty, 0) for (a, sh, sc, ty), b in zip(spec, bufs)])
table = torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()).to(dev)


def timeit(name, fn, n=20):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:32s} {1e3 * a.elapsed_time(b) / n:8.1f} us", flush=True)


st = None
timeit("ffx_pack (4 matrices)", lambda: lib.ffno_ffx_pack(p(table), 4, C, H, st))
timeit("ff_fwd  fp32 (h+mask)", lambda: lib.ffno_ff_fwd(p(s), p(resid), p(W1), p(b1), p(W2), p(b2), p(out), p(h), p(mask), P, C, H, st))
ref = out.clone()
timeit("ff_fwd  fp32 (inference)", lambda: lib.ffno_ff_fwd(p(s), p(resid), p(W1), p(b1), p(W2), p(b2), p(out), None, None, P, C, H, st))
timeit("ff_bwd_data fp32", lambda: lib.ffno_ff_bwd_data(p(db), p(mask), p(W1t), p(W2t), p(dh), p(ds), P, C, H, st))
ref_ds = ds.clone()
timeit("ff_bwd_weights_partial fp32", lambda: lib.ffno_ff_bwd_weights_partial(p(s), p(db), p(h), p(dh), p(partial), P, C, H, nsplit, st))
lib.ffno_ff_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, nsplit, 0, st)
ref_g = [t.clone() for t in (gW1, gW2, gb1, gb2)]
timeit("ffx_fwd  (mask)", lambda: lib.ffno_ffx_fwd(p(s), p(resid), p(bufs[0]), p(b1), p(bufs[1]), p(b2), p(out), p(mask), P, C, H, st))
print("   fwd  rel diff vs fp32 kernel", float((out - ref).norm() / ref.norm()))
timeit("ffx_fwd  (inference)", lambda: lib.ffno_ffx_fwd(p(s), p(resid), p(bufs[0]), p(b1), p(bufs[1]), p(b2), p(out), None, P, C, H, st))
timeit("ffx_bwd_data", lambda: lib.ffno_ffx_bwd_data(p(db), p(mask), p(bufs[2]), p(bufs[3]), p(ds), P, C, H, st))
print("   ds   rel diff vs fp32 kernel", float((ds - ref_ds).norm() / ref_ds.norm()))
timeit("ffx_bwd_weights_partial", lambda: lib.ffno_ffx_bwd_weights_partial(p(s), p(db), p(bufs[0]), p(b1), p(bufs[2]), p(partial), P, C, H, nsplit, st))
timeit("ffx_bwd_weights_reduce", lambda: lib.ffno_ffx_bwd_weights_reduce(p(partial), p(gW1), p(gW2), p(gb1), p(gb2), C, H, nsplit, 0, st))
for n, a, b in zip(("dW1", "dW2", "db1", "db2"), (gW1, gW2, gb1, gb2), ref_g):
    print(f"   {n}  rel diff vs fp32 kernels", float((a - b).norm() / b.norm()))
