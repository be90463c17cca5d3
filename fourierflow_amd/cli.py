"""`python -m fourierflow_amd {train,test,predict} CONFIG.yaml [overrides...]` -- the command surface of the reference
(`fourierflow train | test | predict`, reference commands/train.py:27-148, commands/test.py:24-90,
commands/predict.py:24-110) for the routines built here, with the same positional arguments and flag names
(``--force --resume --checkpoint-id --trial --debug --no-logging --map-location``) and the same on-disk layout:

    <config_dir>/checkpoints/trial-<trial>-<id>/epoch=<e>-step=<s>-valid_loss=<v>.ckpt   (best, what `test` / `predict` load)
    <config_dir>/checkpoints/trial-<trial>-<id>/last.ckpt                                 (what `--resume` continues from)

What is NOT here is the reference's control plane: Hydra (the loader of fourierflow_amd/config.py resolves the same
interpolations), Lightning (the routines run their own fused step), wandb (one JSON line per logged step on stdout)
and the dataset builders.  Batches therefore come from ``--data FILE.npz`` (arrays named like
the builder's batches: ``x``/``y`` [, ``f``, ``mu``] for the Markov and mesh routines, ``data`` for the rollout routine;
first axis = samples) or, without it, are synthetic N(0,1) fields of the configured geometry.
"""
from __future__ import annotations

import json
import math
import time
from pathlib import Path
from typing import Dict, Iterator, List, Optional

import numpy as np
import torch
from typer import Argument, Option, Typer

from .config import build_routine, load_config

app = Typer(add_completion=False, help=__doc__)
_LAST: Dict[str, object] = {}


def _last_routine():
    """The routine object the last `train` command of this process built (for callers that drive the CLI in-process)."""
    return _LAST.get("routine")


# ------------------------------------------------------------------------------------------------------------------
def _kind(routine) -> str:
    name = type(routine).__name__
    return {"Grid2DMarkovExperiment": "markov", "Grid2DRolloutExperiment": "rollout",
            "StructuredMeshExperiment": "mesh"}[name]


def _device(device: Optional[str]) -> torch.device:
    return torch.device(device or "cuda:0")


def _init_distributed(device: Optional[str]):
    """(rank, world, device).  Started as one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m
    fourierflow_amd train ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) the command joins the job --
    the counterpart of the reference handing Lightning a DDPPlugin (commands/train.py:83-84): weights are broadcast from
    rank 0, every rank draws its own shard of each global batch, the flat gradient buffer is all-reduced once per step
    (FFNOTrainer) and the normaliser statistics are global (Normalizer.sync_across_ranks).  RCCL on GPUs, gloo on CPU tensors."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or "RANK" not in os.environ:
        return 0, 1, _device(device)
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(device) if device else torch.device("cuda", local)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    if not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl" if dev.type == "cuda" else "gloo", rank=rank, world_size=world)
    return rank, world, dev


class _Batches:
    """Batches of the routine's geometry: slices of an .npz file, or synthetic."""

    def __init__(self, routine, cfg, dev, data: Optional[Path], batch_size: Optional[int], grid: int, size: Optional[List[int]],
                 seed: int, rank: int = 0, world: int = 1):
        self.routine, self.dev, self.kind = routine, dev, _kind(routine)
        self.rank, self.world = rank, world      # data parallel: rank r takes every world-th batch of a file / its own stream
        seed = seed + 1000003 * rank
        self.B = batch_size or int(cfg.get("builder", {}).get("batch_size", 19))
        self.grid, self.size = grid, tuple(size) if size else None
        self.gen = torch.Generator().manual_seed(seed)
        self.arrays: Optional[Dict[str, np.ndarray]] = None
        if data is not None:
            with np.load(str(data)) as z:
                self.arrays = {k: z[k].astype(np.float32) for k in z.files}
            need = ("data",) if self.kind == "rollout" else ("x", "y")
            missing = [k for k in need if k not in self.arrays]
            if missing:
                raise ValueError(f"{data}: arrays {missing} missing (found {sorted(self.arrays)})")

    def _rand(self, *shape):
        return torch.randn(*shape, generator=self.gen).to(self.dev)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        while True:
            yield from self.epoch()

    def epoch(self) -> Iterator[Dict[str, torch.Tensor]]:
        if self.arrays is not None:
            n = len(next(iter(self.arrays.values())))
            nb = (n // self.B) // self.world * self.world        # the same number of batches on every rank
            for i in range(self.rank * self.B, nb * self.B, self.world * self.B):
                b = {k: torch.from_numpy(v[i:i + self.B]).to(self.dev) for k, v in self.arrays.items()}
                yield self._finish(b)
            return
        yield self._finish(self._synthetic())

    def _synthetic(self):
        B, G, r = self.B, self.grid, self.routine
        if self.kind == "rollout":
            return dict(data=self._rand(B, G, G, 10 + r.n_steps))
        if self.kind == "mesh":
            size = self.size or (G, G)
            cin = r.model.input_dim - len(size)
            return dict(x=self._rand(B, *size, cin), y=self._rand(B, *size, getattr(r.model, "output_dim", 1)))
        b = dict(x=self._rand(B, G, G, 1), y=self._rand(B, G, G, 1))
        if getattr(r, "append_force", False):
            b["f"] = self._rand(B, G, G)
        if getattr(r, "append_mu", False):
            b["mu"] = torch.rand(B, generator=self.gen).to(self.dev)
        return b

    def _finish(self, b):
        if self.kind == "rollout":      # the reference's forward() splits `data` and appends the positions (:38-50)
            d = b["data"]
            B, X, Y, _ = d.shape
            xx = torch.cat([d[..., :10], self.routine._positions(B, X, Y, d.device)], dim=-1)
            return dict(x=xx, y=d[..., 10:].contiguous())
        return b


def _train_step(routine, kind, batch, epoch, step):
    if kind == "rollout":
        return routine.training_step(batch, step)[0]
    if kind == "mesh":
        return routine.training_step(batch, step)
    return routine.training_step(batch, epoch=epoch)


def _valid_loss(routine, kind, batch) -> float:
    """`valid_loss` of the checkpoint file name: the routine's validation metric on one batch."""
    # the reference validates / tests under model.eval(): the Normalizer must NOT accumulate the validation batch into its
    # running statistics (normalizer.py:45-49) -- and they must not leak into the checkpoints written afterwards
    was_training = routine.training
    routine.eval()
    try:
        with torch.no_grad():
            if kind == "rollout":
                return float(routine.validation_step(batch)["valid_loss"].item())
            if kind == "mesh":
                return float(routine.validation_step(batch).item())
            tr = routine.trainer()
            pred = routine._unshuffle(tr.engine.forward(routine._shuffle(routine._build_features(batch, add_noise=False)), False))
            target = (batch["dy"] if routine.learn_difference else batch["y"]).contiguous()
            return float(tr.loss_and_grad(pred, target, routine._affine_tensor())[0].item())
    finally:
        routine.train(was_training)


def _trial_dir(config_dir: Path, trial: int, checkpoint_id: Optional[str], create: bool) -> Path:
    root = config_dir / "checkpoints"
    found = sorted(root.glob(f"trial-{trial}-*"))
    if checkpoint_id:
        d = root / f"trial-{trial}-{checkpoint_id}"
    elif found:
        d = found[-1]
    else:
        d = root / f"trial-{trial}-{time.strftime('%Y%m%d-%H%M%S')}"
    if create:
        d.mkdir(parents=True, exist_ok=True)
    return d


def _best_checkpoint(config_dir: Path, trial: int, explicit: Optional[str]) -> Path:
    if explicit:
        return Path(explicit)
    paths = sorted((config_dir / "checkpoints").glob(f"trial-{trial}-*/epoch*.ckpt"))
    if len(paths) != 1:      # the reference asserts exactly one (commands/test.py:56-60)
        raise FileNotFoundError(f"expected exactly one checkpoints/trial-{trial}-*/epoch*.ckpt under {config_dir}, "
                                f"found {len(paths)}")
    return paths[0]


# ------------------------------------------------------------------------------------------------------------------
@app.command()
def train(config_path: Path, overrides: Optional[List[str]] = Argument(None), force: bool = False, resume: bool = False,
          checkpoint_id: Optional[str] = None, trial: int = 0, debug: bool = False, no_logging: bool = False,
          steps: int = Option(20, help="optimisation steps to run (the reference runs trainer.max_epochs epochs)"),
          accumulation_batches: int = Option(4, help="epoch-0 batches that only accumulate the normaliser statistics"),
          steps_per_epoch: int = Option(0, help="advance the epoch counter (StepLR, file names) every N steps; 0 = never"),
          data: Optional[Path] = Option(None, help=".npz with the builder's batch arrays; default: synthetic fields"),
          batch_size: Optional[int] = None, grid: int = 64, size: Optional[List[int]] = Option(None, help="mesh size"),
          device: Optional[str] = Option(None, hidden=True)):
    """Train: build the routine from CONFIG (+ `a.b=c` overrides) and run fused optimisation steps."""
    cfg = load_config(str(config_path), overrides or [])
    rank, world, dev = _init_distributed(device)
    torch.manual_seed(int(cfg.get("seed", 7231 + trial)))      # commands/train.py:61-64 (the same initial weights on every rank)
    routine = build_routine(cfg).to(dev)
    _LAST["routine"] = routine
    kind = _kind(routine)
    batches = _Batches(routine, cfg, dev, data, batch_size, grid, size, seed=7231 + trial, rank=rank, world=world)
    # checkpoints and the log lines are rank 0's (every rank holds the same weights and the same global normaliser statistics)
    trial_dir = None if no_logging else _trial_dir(config_path.parent, trial, checkpoint_id, create=rank == 0)
    out_dir = trial_dir if rank == 0 else None      # (only rank 0 writes; every rank reads the checkpoint it resumes from)
    if out_dir is not None and force and not resume:
        for old in out_dir.glob("*.ckpt"):       # delete_old_results (commands/train.py:58)
            old.unlink()
    start = dict(epoch=0, global_step=0)
    if resume:
        if trial_dir is None or not (trial_dir / "last.ckpt").exists():
            raise FileNotFoundError("--resume needs checkpoints/trial-<trial>-*/last.ckpt (commands/train.py:74-80)")
        start = routine.resume_from_checkpoint(str(trial_dir / "last.ckpt"))
    it = iter(batches)
    epoch = start["epoch"]
    if kind == "markov" and not resume and routine.should_normalize:      # epoch 0: statistics only (:376-378)
        for _ in range(accumulation_batches):
            routine.training_step(next(it), epoch=0)
        epoch = max(epoch, 1)
    if hasattr(routine, "current_epoch"):
        routine.current_epoch = epoch
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for step in range(steps):
        loss = _train_step(routine, kind, next(it), epoch, step)
        if steps_per_epoch and (step + 1) % steps_per_epoch == 0:
            epoch += 1
            if hasattr(routine, "on_train_epoch_end"):
                routine.on_train_epoch_end()
            elif hasattr(routine, "current_epoch"):
                routine.current_epoch = epoch
        if step % max(1, steps // 5) == 0 or step == steps - 1:
            lv = float(loss.item())
            if not math.isfinite(lv):
                # activations, spectra and gradients are range-safe by construction (include/ffno.h "Range words"); what is left is
                # data that is non-finite already, a diverged run, or WEIGHTS beyond the half format's range in the split-fp16
                # packs (|W| >= 65504) -- the any-range arithmetic is one switch away
                raise FloatingPointError(
                    f"non-finite training loss at step {start['global_step'] + step}: check the data and the learning rate; if the "
                    f"weights have grown past 6.5e4, run with FFNO_FF_SPLIT=bf16x3 FFNO_X3_MIX_SPLIT=bf16x3 (any fp32 range)")
            if rank == 0:
                print(json.dumps(dict(step=start["global_step"] + step, epoch=epoch, train_loss=round(lv, 6),
                                      lr=routine.trainer().current_lr())), flush=True)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    summary = dict(steps=steps, batch=batches.B, world_size=world, steps_per_s=round(steps / max(dt, 1e-9), 2),
                   resumed_from_step=start["global_step"])
    if out_dir is not None:
        gs = start["global_step"] + steps
        vl = _valid_loss(routine, kind, next(it))
        for old in out_dir.glob("epoch*.ckpt"):      # CustomModelCheckpoint keeps the single best file
            old.unlink()
        best = out_dir / f"epoch={epoch}-step={gs}-valid_loss={vl:.5f}.ckpt"
        routine.save_checkpoint(str(best), epoch=epoch, global_step=gs)
        routine.save_checkpoint(str(out_dir / "last.ckpt"), epoch=epoch, global_step=gs)
        summary.update(valid_loss=round(vl, 6), checkpoint=str(best))
    if rank == 0:
        print(json.dumps(summary), flush=True)
    if world > 1:
        torch.distributed.barrier()


@app.command()
def test(config_path: Path, overrides: Optional[List[str]] = Argument(None), force: bool = False, trial: int = 0,
         map_location: Optional[str] = None, debug: bool = False, no_logging: bool = False,
         batches: int = Option(1, help="test batches to average over"), data: Optional[Path] = None,
         batch_size: Optional[int] = None, grid: int = 64, size: Optional[List[int]] = None,
         device: Optional[str] = Option(None, hidden=True)):
    """Test: load the best checkpoint of the trial (or `checkpoint_path=...` override) and report the test metrics."""
    cfg = load_config(str(config_path), overrides or [])
    dev = _device(device)
    routine = build_routine(cfg).to(dev)
    kind = _kind(routine)
    ckpt = _best_checkpoint(config_path.parent, trial, cfg.get("checkpoint_path"))
    routine.load_lightning_model_state(str(ckpt), map_location)
    routine.to(dev)
    routine.eval()       # trainer.test / predict run under eval(): no statistics accumulation (commands/test.py, normalizer.py:48)
    src = _Batches(routine, cfg, dev, data, batch_size, grid, size, seed=7231 + trial)
    it = iter(src)
    acc: Dict[str, float] = {}
    for _ in range(batches):
        b = next(it)
        if kind == "rollout":
            m = {k: v for k, v in routine.test_step(b).items() if k in ("test_loss", "test_loss_avg", "test_time_until")}
        else:
            m = {"test_loss": _valid_loss(routine, kind, b)}
        for k, v in m.items():
            acc[k] = acc.get(k, 0.0) + float(v) / batches
    print(json.dumps(dict(checkpoint=str(ckpt), **{k: round(v, 6) for k, v in acc.items()})), flush=True)


@app.command()
def predict(config_path: Path, overrides: Optional[List[str]] = Argument(None), trial: int = 0,
            map_location: Optional[str] = None, debug: bool = False,
            n_steps: Optional[int] = Option(None, help="rollout length (default: the routine's n_steps)"),
            output: Optional[Path] = Option(None, help="write the predictions here (.npz); default: <trial dir>/predictions.npz"),
            data: Optional[Path] = None, batch_size: Optional[int] = 1, grid: int = 64, size: Optional[List[int]] = None,
            device: Optional[str] = Option(None, hidden=True)):
    """Predict: load the best checkpoint, run the model autoregressively (grid routines) or once (mesh routine), save the
    predictions and report the time per model step (the reference's `inference_time`, commands/train.py:132-148)."""
    cfg = load_config(str(config_path), overrides or [])
    dev = _device(device)
    routine = build_routine(cfg).to(dev)
    kind = _kind(routine)
    ckpt = _best_checkpoint(config_path.parent, trial, cfg.get("checkpoint_path"))
    routine.load_lightning_model_state(str(ckpt), map_location)
    routine.to(dev)
    routine.eval()       # trainer.test / predict run under eval(): no statistics accumulation (commands/test.py, normalizer.py:48)
    b = next(iter(_Batches(routine, cfg, dev, data, batch_size, grid, size, seed=7231 + trial)))

    def run():
        with torch.no_grad():
            if kind == "markov":
                return routine.rollout(b["x"], n_steps, b.get("f"), b.get("mu"))
            if kind == "rollout":
                routine.eval()
                return routine._learning_step(b)[2]
            return routine.trainer().predict(b["x"])

    run()       # warm-up (routine.warmup() in the reference)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    preds = run()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    steps = (n_steps or getattr(routine, "n_steps", None) or 1) if kind != "mesh" else 1
    out = output or (ckpt.parent / "predictions.npz")
    np.savez(str(out), preds=preds.detach().cpu().numpy())
    print(json.dumps(dict(checkpoint=str(ckpt), predictions=str(out), shape=list(preds.shape),
                          inference_time_ms_per_step=round(1e3 * elapsed / steps, 4))), flush=True)


def main():
    app()


if __name__ == "__main__":
    main()
