"""Stand-alone timing of the inference layer's two launches (csrc/infer.hip) against the training layer's forward launches on the
same operands: markov/24 geometry (batch 32, 64 x 64, 16 modes) by default.  200 back-to-back launches between two HIP events.
    python tools/time_infer.py [B M N K]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from backend_util import Backend  # noqa: E402
from test_infer_layer import _setup, layer_fp64  # noqa: E402
from fourierflow_amd._capi import LayerFwdDesc, LayerInferDesc  # noqa: E402


def main():
    B, M, N, K = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (32, 64, 64, 16)
    be = Backend("gpu")
    lib, p = be.lib, be.ptr
    C, H = 64, 256
    S = _setup(be, B, M, N, K, seed=1)
    out = be.empty(S["x"].shape)
    oword = be.zeros(1, np.uint32)
    a, b = S["branch"](0, S["mix"][0]), S["branch"](1, S["mix"][1])
    d = LayerInferDesc(a, b, 2, 0, p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), p(S["dx"]), p(out), C, H, p(oword))
    s_img, t_img, out2 = be.empty(S["x"].shape), be.empty(S["x"].shape), be.empty(S["x"].shape)
    sword = be.zeros(1, np.uint32)
    P = B * M * N
    a2, b2 = S["branch"](0, s_img, p(sword)), S["branch"](1, t_img, p(sword))
    d2 = LayerFwdDesc(a2, b2, 1, 2, p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), None, p(S["dx"]), p(out2), None, P, C, H,
                      1, 0, 0, 0, 0, None)
    assert lib.ffno_layer_infer(ctypes.byref(d), None) == 0
    assert lib.ffno_layer_fwd(ctypes.byref(d2), None) == 0
    torch.cuda.synchronize()
    g1, g2 = be.get(out), be.get(out2)
    print("infer vs training layer rel-L2:", float(np.linalg.norm(g1 - g2) / np.linalg.norm(g2)))
    if B * M * N <= 32 * 64 * 64 and os.environ.get("TIME_INFER_FP64"):
        ref, _ = layer_fp64(S["x"], S["w"][0], S["w"][1], S["W1"], S["b1"], S["W2"], S["b2"], K)
        print("infer vs fp64 update rel-L2:", float(np.linalg.norm(g1 - S["x"] - (ref - S["x"])) / np.linalg.norm(ref - S["x"])))

    def timeit(fn, n=200):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n

    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {
        "K1 mix_pair": timeit(lambda: lib.ffno_spectral_x3_mix_pair(ctypes.byref(a), ctypes.byref(b), C, 2, st)),
        "K2 infer_ff": timeit(lambda: lib.ffno_infer_ff(ctypes.byref(a), ctypes.byref(b), p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]),
                                                        p(S["db2"]), p(S["dx"]), p(out), C, H, p(oword), st)),
        "layer_infer (K1+K2)": timeit(lambda: lib.ffno_layer_infer(ctypes.byref(d), st)),
        "x3_pair (training fwd, nothing saved)": timeit(lambda: lib.ffno_spectral_x3_pair(ctypes.byref(a2), ctypes.byref(b2), C, 0, 1, 0, 2, st)),
        "layer_fwd (pair + chain, nothing saved)": timeit(lambda: lib.ffno_layer_fwd(ctypes.byref(d2), st)),
    }
    # FFNO_BRANCH_SELF_RANGE (axis lengths <= 64): every line scaled from its own maximum, no range word read or recorded
    if M <= 64 and N <= 64:
        from fourierflow_amd._capi import BRANCH_SELF_RANGE
        a3, b3 = S["branch"](0, S["mix"][0]), S["branch"](1, S["mix"][1])
        a3.flags = b3.flags = BRANCH_SELF_RANGE
        a3.in_amax = b3.in_amax = None
        d3 = LayerInferDesc(a3, b3, 2, 0, p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), p(S["dx"]), p(out), C, H, None)
        assert lib.ffno_layer_infer(ctypes.byref(d3), None) == 0
        torch.cuda.synchronize()
        g3 = be.get(out)
        print("self-ranged vs range-word inference layer rel-L2:", float(np.linalg.norm(g3 - g1) / np.linalg.norm(g1)))
        for rep in range(2):      # (twice, interleaved with the range-word form: same box, same minute)
            res[f"K1 mix_pair, self-ranged lines [{rep}]"] = timeit(lambda: lib.ffno_spectral_x3_mix_pair(ctypes.byref(a3), ctypes.byref(b3), C, 2, st))
            res[f"K1 mix_pair, range word [{rep}]"] = timeit(lambda: lib.ffno_spectral_x3_mix_pair(ctypes.byref(a), ctypes.byref(b), C, 2, st))
            res[f"K2 infer_ff, no out word [{rep}]"] = timeit(lambda: lib.ffno_infer_ff(
                ctypes.byref(a3), ctypes.byref(b3), p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), p(S["dx"]), p(out), C, H, None, st))
            res[f"K2 infer_ff, out word [{rep}]"] = timeit(lambda: lib.ffno_infer_ff(
                ctypes.byref(a), ctypes.byref(b), p(S["packs"][0]), p(S["db1"]), p(S["packs"][1]), p(S["db2"]), p(S["dx"]), p(out), C, H, p(oword), st))
            res[f"layer_infer self-ranged [{rep}]"] = timeit(lambda: lib.ffno_layer_infer(ctypes.byref(d3), st))
            res[f"layer_infer range words [{rep}]"] = timeit(lambda: lib.ffno_layer_infer(ctypes.byref(d), st))
    for k, v in res.items():
        print(f"{k:45s} {v:8.2f} us")


if __name__ == "__main__":
    main()
