"""`python -m fourierflow_amd {train,test,predict}` (the reference's Typer commands, commands/train.py:27-148,
test.py:24-90, predict.py:24-110): positional config + overrides, the reference's flag names, and its checkpoint layout
(checkpoints/trial-<n>-<id>/epoch=..-step=..-valid_loss=...ckpt + last.ckpt)."""
import json
import os

import numpy as np
import pytest
from typer.testing import CliRunner

from backend_util import host_device  # noqa: F401

MARKOV = """
routine:
  _target_: fourierflow.routines.Grid2DMarkovExperiment
  conv:
    _target_: fourierflow.modules.FNOFactorized2DBlock
    modes: 4
    width: 32
    n_layers: 2
    input_dim: 3
    share_weight: true
    factor: 4
    ff_weight_norm: true
    gain: 0.1
  n_steps: 3
  max_accumulations: 100
  noise_std: 0.01
  optimizer:
    _target_: functools.partial
    _args_: ["${get_method: torch.optim.AdamW}"]
    lr: 0.0025
    weight_decay: 0.0001
  scheduler:
    scheduler:
      _target_: functools.partial
      _args_: ["${get_method: fourierflow.schedulers.CosineWithWarmupScheduler}"]
      num_warmup_steps: 5
      num_training_steps: 100
      num_cycles: 0.5
builder:
  batch_size: 2
"""

ROLLOUT = """
routine:
  _target_: fourierflow.routines.Grid2DRolloutExperiment
  conv:
    _target_: fourierflow.modules.FNOZongyi2DBlock
    modes1: 3
    modes2: 3
    width: 20
    n_layers: 2
  n_steps: 2
  optimizer:
    _target_: functools.partial
    _args_: ["${get_method: torch.optim.AdamW}"]
    lr: 0.0025
    weight_decay: 0.0001
  scheduler:
    scheduler:
      _target_: functools.partial
      _args_: ["${get_method: torch.optim.lr_scheduler.StepLR}"]
      step_size: 1
      gamma: 0.5
builder:
  batch_size: 2
"""


def _run(args, device):
    from fourierflow_amd.cli import app
    res = CliRunner().invoke(app, [*args, "--device", device])
    assert res.exit_code == 0, (res.output, res.exception)
    return [json.loads(l) for l in res.output.splitlines() if l.startswith("{")]


def test_cli_train_resume_test_predict_markov(tmp_path, host_device):
    cfg = tmp_path / "config.yaml"
    cfg.write_text(MARKOV)
    out = _run(["train", str(cfg), "routine.noise_std=0.0", "--steps", "3", "--grid", "8", "--accumulation-batches", "1"],
               host_device)
    assert [o["step"] for o in out[:-1]] == [0, 1, 2] and out[-1]["resumed_from_step"] == 0
    tdirs = os.listdir(tmp_path / "checkpoints")
    assert len(tdirs) == 1 and tdirs[0].startswith("trial-0-")
    files = sorted(os.listdir(tmp_path / "checkpoints" / tdirs[0]))
    assert files[1] == "last.ckpt" and files[0].startswith("epoch=1-step=3-valid_loss=") and files[0].endswith(".ckpt")
    # --resume continues the step count and the schedule from last.ckpt; the single best file is replaced
    out = _run(["train", str(cfg), "--steps", "2", "--grid", "8", "--resume"], host_device)
    assert out[0]["step"] == 3 and out[-1]["resumed_from_step"] == 3
    assert abs(out[0]["lr"] - 0.0025 * 4 / 5) < 1e-12            # warm-up factor of step 3 (next step = 4 of 5)
    files = sorted(os.listdir(tmp_path / "checkpoints" / tdirs[0]))
    assert len(files) == 2 and files[0].startswith("epoch=1-step=5-")
    t = _run(["test", str(cfg), "--grid", "8", "--batches", "2"], host_device)[-1]
    assert t["checkpoint"].endswith(files[0]) and 0.5 < t["test_loss"] < 1.5
    p = _run(["predict", str(cfg), "--grid", "8"], host_device)[-1]
    preds = np.load(p["predictions"])["preds"]
    assert preds.shape == (1, 8, 8, 3) and np.isfinite(preds).all() and p["inference_time_ms_per_step"] > 0
    # a second trial directory makes `test` ambiguous only for its own trial number
    with pytest.raises(AssertionError):
        _run(["test", str(cfg), "--grid", "8", "--trial", "1"], host_device)


def test_cli_rollout_routine_with_data_file(tmp_path, host_device):
    cfg = tmp_path / "config.yaml"
    cfg.write_text(ROLLOUT)
    rs = np.random.RandomState(0)
    np.savez(tmp_path / "data.npz", data=rs.standard_normal((4, 8, 8, 12)).astype(np.float32))
    out = _run(["train", str(cfg), "--steps", "2", "--data", str(tmp_path / "data.npz"), "--steps-per-epoch", "1",
                "--checkpoint-id", "abc"], host_device)
    assert os.path.isdir(tmp_path / "checkpoints" / "trial-0-abc")
    assert [o["lr"] for o in out[:-1]] == [0.00125, 0.000625]          # StepLR(step_size=1, gamma=0.5) after each epoch
    t = _run(["test", str(cfg), "--data", str(tmp_path / "data.npz")], host_device)[-1]
    assert set(t) >= {"test_loss", "test_loss_avg", "test_time_until"}
    p = _run(["predict", str(cfg), "--data", str(tmp_path / "data.npz"), "--batch-size", "2"], host_device)[-1]
    assert p["shape"] == [2, 8, 8, 2]
    bad = tmp_path / "bad.npz"
    np.savez(bad, x=np.zeros((2, 4, 4, 1), np.float32))
    from fourierflow_amd.cli import app
    res = CliRunner().invoke(app, ["train", str(cfg), "--data", str(bad), "--device", host_device])
    assert res.exit_code != 0 and isinstance(res.exception, ValueError)


def _cli_ddp_worker(rank, world, port, cfg_path, data_path):
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), FFNO_ALLOW_TEST_BACKEND="1")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    from backend_util import emu_lib
    from fourierflow_amd import _lib
    _lib._install_test_backend(emu_lib())
    from fourierflow_amd.cli import app
    res = CliRunner().invoke(app, ["train", cfg_path, "routine.noise_std=0.0", "--steps", "3", "--data", data_path,
                                   "--accumulation-batches", "1", "--device", "cpu"])
    assert res.exit_code == 0, (res.output, res.exception)
    lines = [json.loads(l) for l in res.output.splitlines() if l.startswith("{")]
    assert (len(lines) > 0) == (rank == 0)            # the log lines and the checkpoints are rank 0's
    # every rank ends with the same weights and the same (global) normaliser statistics
    from fourierflow_amd.cli import _last_routine
    sd = {k: v.detach().cpu().numpy() for k, v in _last_routine().state_dict().items()}
    np.savez(os.path.join(os.path.dirname(cfg_path), f"rank{rank}.npz"), **sd)
    torch.distributed.destroy_process_group()


def test_cli_train_joins_a_two_rank_job(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m fourierflow_amd train ...` (here: two spawned processes, gloo, the CPU
    emulator): the command initialises the process group from RANK / WORLD_SIZE, rank r takes every second batch of the data
    file, gradients are all-reduced per step, only rank 0 logs and writes checkpoints -- and both ranks end bit-identical.
    (The reference's switch: Lightning's DDPPlugin, commands/train.py:83-84.)"""
    import torch.multiprocessing as mp
    cfg = tmp_path / "config.yaml"
    cfg.write_text(MARKOV)
    rs = np.random.RandomState(1)
    np.savez(tmp_path / "data.npz", x=rs.standard_normal((8, 8, 8, 1)).astype(np.float32),
             y=rs.standard_normal((8, 8, 8, 1)).astype(np.float32))
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_cli_ddp_worker, args=(2, port, str(cfg), str(tmp_path / "data.npz")), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert set(r0.files) == set(r1.files) and len(r0.files) > 10
    for k in r0.files:
        np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
    tdirs = os.listdir(tmp_path / "checkpoints")
    assert len(tdirs) == 1 and "last.ckpt" in os.listdir(tmp_path / "checkpoints" / tdirs[0])
