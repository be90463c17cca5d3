"""The whole layer stack of a forward-only pass: ffno_infer_stack (one persistent launch, the 8 workgroups of an image run both kernels
of every layer as phases) against the loop of ffno_layer_infer calls, markov/24 geometry (batch 32, 64 x 64, 16 modes, 24 layers).
    python tools/time_stack.py [layers]"""
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
from fourierflow_amd._capi import BRANCH_SELF_RANGE, InferStackDesc, InferStackLayer, LayerInferDesc  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    B, K, C, H = 32, 16, 64, 256
    be = Backend("gpu")
    lib, p = be.lib, be.ptr
    S, layers = _stack_setup(be, B, K, L, seed=3)
    x0 = (S["x"] * 0.05).astype(np.float32)      # (keeps 24 random layers finite)
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
    descs = []
    for l, y in enumerate(layers):
        lst = l == L - 1
        descs.append(LayerInferDesc(a, b, 2, 0, p(y["packs"][0]), p(y["db1"]), p(y["packs"][1]), p(y["db2"]), None if lst else p(dx),
                                    p(last) if lst else p(dx), C, H, None))
    arr = (InferStackLayer * L)(*[InferStackLayer(a.planes, b.planes, p(y["packs"][0]), p(y["db1"]), p(y["packs"][1]), p(y["db2"]))
                                  for y in layers])
    sync = be.zeros(int(lib.ffno_infer_stack_sync_words(B)), np.uint32)
    sd0 = InferStackDesc(a, b, ctypes.cast(arr, ctypes.c_void_p), L, C, H, 0, p(last), p(sync))
    sd1 = InferStackDesc(a, b, ctypes.cast(arr, ctypes.c_void_p), L, C, H, 1, p(last), p(sync))

    def loop():
        for d in descs:
            assert lib.ffno_layer_infer(ctypes.byref(d), st) == 0

    def reset():
        dx.copy_(torch.from_numpy(x0)) if hasattr(dx, "copy_") else None

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n

    for rep in range(3):
        t_loop = timeit(loop)
        t_p = timeit(lambda: lib.ffno_infer_stack(ctypes.byref(sd0), st))
        t_1 = timeit(lambda: lib.ffno_infer_stack(ctypes.byref(sd1), st))
        words = np.array(be.get(sync))
        print(f"[{rep}] {L} layers: loop of ffno_layer_infer {t_loop:8.1f} us ({t_loop / L:.2f} / layer)   persistent {t_p:8.1f} us "
              f"({t_p / L:.2f} / layer)   one launch per phase {t_1:8.1f} us   error word {words[8 + B]}")


if __name__ == "__main__":
    main()
