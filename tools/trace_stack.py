"""Where the time of one image group goes inside the persistent inference stack (ffno_infer_stack, mode | 2): the 8 workgroups of group
0 stamp the device's 100 MHz clock at the start of every phase, when their body is done and when the group barrier lets them through.
    python tools/trace_stack.py [batch] [layers]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from backend_util import Backend  # noqa: E402
from test_infer_layer import _stack_setup  # noqa: E402
from fourierflow_amd._capi import BRANCH_SELF_RANGE, InferStackDesc, InferStackLayer  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    K, C, H = 16, 64, 256
    be = Backend("gpu")
    lib, p = be.lib, be.ptr
    S, layers = _stack_setup(be, B, K, L, seed=3)
    x0 = (S["x"] * 0.05).astype(np.float32)
    dx = be.put(x0)
    last = be.empty(x0.shape)
    brs = []
    for i in range(2):
        br = S["branch"](i, S["mix"][i])
        br.in_ = p(dx)
        br.flags, br.in_amax = BRANCH_SELF_RANGE, None
        brs.append(br)
    a, b = brs
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    arr = (InferStackLayer * L)(*[InferStackLayer(a.planes, b.planes, p(y["packs"][0]), p(y["db1"]), p(y["packs"][1]), p(y["db2"]))
                                  for y in layers])
    ns = int(lib.ffno_infer_stack_sync_words(B))
    nt = int(lib.ffno_infer_stack_trace_words(L))
    sync = be.zeros(ns + nt, np.uint32)
    sd = InferStackDesc(a, b, ctypes.cast(arr, ctypes.c_void_p), L, C, H, 2, p(last), p(sync))
    for _ in range(5):
        assert lib.ffno_infer_stack(ctypes.byref(sd), st) == 0
    torch.cuda.synchronize()
    w = np.array(be.get(sync))
    assert w[ns - 1] == 0, f"error word {w[ns - 1]}"
    t = w[ns + 1:].view(np.uint64).reshape(8, 2 * L, 12).astype(np.float64) * 0.01      # us
    t0 = t[:, 0, 0].min()
    t -= t0
    body = t[:, :, 1] - t[:, :, 0]
    wait = t[:, :, 2] - t[:, :, 1]
    print(f"# batch {B}, {L} layers, group 0 (image 0): per phase, over the 8 members -- body us (min / mean / max), barrier us (wait of the "
          f"last arriver = the barrier's own cost; mean = + imbalance)")
    print("# member roles in K1: 0-3 row tiles (16 lines each), 4-7 column tiles; in K2: 8-row tiles 0-7")
    k1, k2, w1, w2, w1min, w2min = [], [], [], [], [], []
    for ph in range(2 * L):
        bb, ww = body[:, ph], wait[:, ph]
        if ph < 6 or ph >= 2 * L - 2:
            print(f"phase {ph:2d} ({'K1' if ph % 2 == 0 else 'K2'} of layer {ph // 2:2d})  body {bb.min():6.2f} / {bb.mean():6.2f} / {bb.max():6.2f}   "
                  f"barrier {ww.min():6.2f} / {ww.mean():6.2f} / {ww.max():6.2f}   start {t[:, ph, 0].min():8.2f} .. {t[:, ph, 0].max():8.2f}")
        if 2 <= ph < 2 * L - 2:
            (k1 if ph % 2 == 0 else k2).append(bb)
            (w1 if ph % 2 == 0 else w2).append(ww.mean())
            (w1min if ph % 2 == 0 else w2min).append(ww.min())
    k1, k2 = np.array(k1), np.array(k2)
    print(f"layers 1..{L - 2}: K1 body mean {k1.mean():.2f} (row tiles {k1[:, :4].mean():.2f}, column tiles {k1[:, 4:].mean():.2f}; slowest member {k1.max(axis=1).mean():.2f})   "
          f"barrier after K1: last arriver {np.mean(w1min):.2f}, mean {np.mean(w1):.2f}")
    print(f"               K2 body mean {k2.mean():.2f} (slowest member {k2.max(axis=1).mean():.2f})   barrier after K2: last arriver {np.mean(w2min):.2f}, mean {np.mean(w2):.2f}")
    # marks inside the bodies (wave 0 of every member; after the body's first / second / third workgroup barrier)
    sub = t[:, 2:2 * L - 2, :]
    for name, par, labels in (("K1", 0, ("first line requested + twiddles staged", "phase 1 (loads + forward DFT of two lines per wave)",
                                          "phase 2 (channel mix, weight stream)", "phase 3' (mixed spectra leave as fragments)")),
                              ("K2", 1, ("phase A (column branch -> LDS) + row lines requested", "phase A' (row branch + tile assembly, weights requested)",
                                          "weights into LDS", "phase B (feed-forward + store)"))):
        ph = sub[:, par::2, :]
        segs = (ph[:, :, 3] - ph[:, :, 0], ph[:, :, 4] - ph[:, :, 3], ph[:, :, 5] - ph[:, :, 4], ph[:, :, 1] - ph[:, :, 5])
        print(f"{name}: " + "   ".join(f"{lab} {seg.mean():.2f}" for lab, seg in zip(labels, segs)))
    ph = sub[:, 0::2, :]
    if ph[:, :, 6].max() > 0:
        names = ("DFT fragments built", "line 0 arrived + its maximum", "line 0 products + tile rows", "line 1 arrived + its maximum", "line 1 products + tile rows", "barrier")
        idx = (3, 6, 7, 8, 9, 10, 4)
        print("K1 phase 1, wave 0: " + "   ".join(f"{n} {(ph[:, :, b] - ph[:, :, a]).mean():.2f}" for n, a, b in zip(names, idx[:-1], idx[1:])))
    span = t[:, 2 * L - 1, 1].max() - t[:, 0, 0].min()
    print(f"whole stack of image 0: {span:.1f} us = {span / L:.2f} us per layer")


if __name__ == "__main__":
    main()
