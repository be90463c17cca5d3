"""BASELINE config 0 (experiments/torus_li/zongyi/4_layers: FNOZongyi2DBlock modes 12, width 20, 4 layers, 64x64) on one
MI355X: forward latency at the reference's CPU-runnable batch (2) and training steps/s of the 10-step rollout routine at
the config's batch size (20), with the oracle's CPU time for the same forward beside it.  One JSON line."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--cpu", type=int, default=1, help="also time the oracle forward on the host cores")
    a = ap.parse_args()
    from fourierflow_amd.modules import FNOZongyi2DBlock
    from fourierflow_amd.routines import Grid2DRolloutExperiment
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    conv = FNOZongyi2DBlock(modes1=12, modes2=12, width=20, n_layers=4)
    routine = Grid2DRolloutExperiment(conv, n_steps=10, optimizer=dict(lr=2.5e-3, weight_decay=1e-4),
                                      scheduler=dict(step_size=100, gamma=0.5)).to(dev)
    G, B = 64, a.batch
    x2 = torch.randn(2, G, G, 12, device=dev)
    with torch.no_grad():
        for _ in range(5):
            conv(x2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            conv(x2)
        torch.cuda.synchronize()
        ms_fwd = (time.perf_counter() - t0) / 50 * 1e3

    def batch():
        xx = torch.cat([torch.randn(B, G, G, 10, device=dev), routine._positions(B, G, G, dev)], dim=-1)
        return dict(x=xx, y=torch.randn(B, G, G, 10, device=dev))

    bt = batch()
    for _ in range(a.warmup):
        loss = routine.training_step(bt)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = routine.training_step(bt)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cpu = None
    if a.cpu:
        from oracle import ffno_oracle as orc      # checker / CPU baseline only
        sd = {k: v.detach().cpu() for k, v in conv.state_dict().items()}
        xc = x2.cpu()
        with torch.no_grad():
            orc.fno_zongyi_2d(sd, xc, modes=12, n_layers=4)
            t0 = time.perf_counter()
            for _ in range(5):
                ref = orc.fno_zongyi_2d(sd, xc, modes=12, n_layers=4)["forecast"]
            cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
            got = conv(x2)["forecast"].cpu()
        cpu = dict(ms_per_forward=round(cpu_ms, 2), cores=torch.get_num_threads(), kind="port",
                   rel_l2_vs_hip=float((got - ref).norm() / ref.norm()))
    print(json.dumps({"metric": "training-steps/sec, Grid2DRolloutExperiment + FNOZongyi2DBlock (torus_li/zongyi/4_layers)",
                      "value": round(a.steps / dt, 2), "unit": f"steps/s (batch {B}, 10-step rollout with BPTT)",
                      "ms_per_step": round(dt / a.steps * 1e3, 3), "ms_per_forward_b2": round(ms_fwd, 3), "dtype": "f32",
                      "data": "synthetic N(0,1)", "final_loss": round(float(loss.item()), 5), "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
