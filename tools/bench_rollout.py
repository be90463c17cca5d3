#!/usr/bin/env python3
"""Inference latency of the Markov rollout (the reference's `inference_time`, routines/grid_2d_markov.py:263-326):
ms per autoregressive model step at small batch (host loop; a hipGraph capture of the same launches was measured and is
no faster -- 1.70 vs 1.64 ms/step at batch 1 at the time (0.87 ms/step today with the paired spectral launch): the step is bound by ~170 dependent kernel boundaries on the device, not by
the host enqueue -- see DESIGN.md "Negative results")."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--n-steps", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--velocity", action="store_true")
    ap.add_argument("--staged", action="store_true", help="three (paired) stage kernels instead of the fused spectral kernel")
    args = ap.parse_args()
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.routines import Grid2DMarkovExperiment
    torch.manual_seed(0)
    blk = FNOFactorized2DBlock(modes=16, width=64, n_layers=args.layers, input_dim=5 if args.velocity else 3, share_weight=True,
                               factor=4, ff_weight_norm=True, gain=0.1)
    exp = Grid2DMarkovExperiment(blk, n_steps=args.n_steps, use_velocity=args.velocity, grid_size=[args.grid]).cuda()
    G, B = args.grid, args.batch
    blk.engine().use_fused = not args.staged
    exp.training_step(dict(x=torch.randn(4, G, G, 1).cuda(), y=torch.randn(4, G, G, 1).cuda()), epoch=0)
    x0 = torch.randn(B, G, G, 1).cuda()

    def timeit(fn):
        fn(x0, args.n_steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn(x0, args.n_steps)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / (args.reps * args.n_steps)

    eager = timeit(exp.rollout)
    print(json.dumps({"metric": "ms per autoregressive step (markov/%d, %dx%d, batch %d%s)" % (args.layers, G, G, B, ", velocity features" if args.velocity else ""),
                      "value": round(eager, 3), "unit": "ms", "higher_is_better": False, "n_steps": args.n_steps}))


if __name__ == "__main__":
    main()
