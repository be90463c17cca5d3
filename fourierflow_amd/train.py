"""`python -m fourierflow_amd.train CONFIG.yaml [overrides...]` -- minimal counterpart of `fourierflow train`
(reference commands/train.py:27-123) for the hot path: build the routine from an experiment config and run
the accumulation epoch + training steps.  Data files are out of scope: batches are synthetic
N(0,1) fields of the configured grid; one JSON line per logged step.
"""
import argparse
import json
import time

import torch

from .config import build_routine, load_config


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("config")
    ap.add_argument("overrides", nargs="*")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--accumulation-batches", type=int, default=4)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--trial", type=int, default=0)
    ap.add_argument("--size", type=int, nargs="+", default=None,
                    help="mesh size for StructuredMeshExperiment configs, e.g. --size 221 51 or --size 101 31 20")
    ap.add_argument("--checkpoint", default=None, help="write a Lightning-layout checkpoint here at the end (last.ckpt)")
    ap.add_argument("--resume", default=None, help="resume weights / optimiser / schedule from this checkpoint")
    args = ap.parse_args(argv)
    cfg = load_config(args.config, args.overrides)
    dev = torch.device("cuda:0")
    torch.manual_seed(7231 + args.trial)
    routine = build_routine(cfg).to(dev)
    B = args.batch_size or int(cfg.get("builder", {}).get("batch_size", 19))
    G = args.grid

    mesh = hasattr(routine, "model")                     # StructuredMeshExperiment: plain x -> y regression on a mesh
    rollout = type(routine).__name__ == "Grid2DRolloutExperiment"    # 10 input frames + n_steps targets (NSZongyiBuilder)
    if rollout:
        def batch():
            xx = torch.cat([torch.randn(B, G, G, 10, device=dev), routine._positions(B, G, G, dev)], dim=-1)
            return dict(x=xx, y=torch.randn(B, G, G, routine.n_steps, device=dev))
    elif mesh:
        size = tuple(args.size or (G, G))
        cin = routine.model.input_dim - len(size)
        cout = getattr(routine.model, "output_dim", 1)

        def batch():
            return dict(x=torch.randn(B, *size, cin, device=dev), y=torch.randn(B, *size, cout, device=dev))
    else:
        def batch():
            return dict(x=torch.randn(B, G, G, 1, device=dev), y=torch.randn(B, G, G, 1, device=dev))

    start = dict(epoch=0, global_step=0)
    if args.resume:
        start = routine.resume_from_checkpoint(args.resume)
    for _ in range(0 if (mesh or rollout or args.resume) else args.accumulation_batches):   # epoch 0: normaliser statistics only
        routine.training_step(batch(), epoch=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(args.steps):
        if rollout:
            loss = routine.training_step(batch(), step)[0]
        else:
            loss = routine.training_step(batch(), step) if mesh else routine.training_step(batch(), epoch=1)
        if step % max(1, args.steps // 5) == 0 or step == args.steps - 1:
            print(json.dumps(dict(step=step, loss=round(float(loss.item()), 6), lr=routine.trainer().current_lr())), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(steps=args.steps, batch=B, grid=G, steps_per_s=round(args.steps / dt, 2),
                          resumed_from_step=start["global_step"])), flush=True)
    if args.checkpoint:
        routine.save_checkpoint(args.checkpoint, epoch=max(1, start["epoch"]))


if __name__ == "__main__":
    main()
