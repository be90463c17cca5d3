#!/usr/bin/env python3
"""Experiment (round 5): does the step gain from running the batch as G independent sub-batches on G streams?  The layer stack
has no cross-sample dependence, so G trainers with batch 32 / G each, every one on its own stream, keep 256 / G-workgroup
launches of different phases in flight at once (one launch's ramp / drain under another's body).  Prints samples/s for
G = 1, 2, 4 (train steps on replicated models: the weight-gradient reduction across groups is NOT included -- an upper bound)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fourierflow_amd.modules import FNOFactorized2DBlock  # noqa: E402
from fourierflow_amd.trainer import FFNOTrainer  # noqa: E402

kw = dict(modes=16, width=64, input_dim=3, n_layers=24, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1)
dev = torch.device("cuda:0")
B = 32
for G in (1, 2, 4, 1, 2, 4):
    torch.manual_seed(0)
    trs, xs, ys, streams = [], [], [], []
    for g in range(G):
        blk = FNOFactorized2DBlock(**kw).to(dev)
        trs.append(FFNOTrainer(blk, lr=2.5e-3, weight_decay=1e-4, num_warmup_steps=500, num_training_steps=100000))
        xs.append(torch.randn(B // G, 64, 64, 3, device=dev))
        ys.append(torch.randn(B // G, 64, 64, 1, device=dev))
        streams.append(torch.cuda.Stream(dev))
    def step():
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                trs[g].train_step(xs[g], ys[g])
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"G = {G} sub-batches of {B // G}: {1e3 * dt:.3f} ms per full batch of {B} -> {B / dt:.0f} samples/s ({1 / dt:.1f} steps/s equivalent)", flush=True)
    del trs, xs, ys, streams
    torch.cuda.empty_cache()
