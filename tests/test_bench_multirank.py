"""`python bench.py --gpus 2` end to end on ONE device (FFNO_BENCH_ONE_DEVICE=1: both ranks on cuda:0 over gloo -- a dry run of the
N > 1 control flow, not a measurement): the self-launch under torch.distributed.run, the per-step gradient all-reduce, the
max-over-ranks timing and the JSON line.  Rounds 1-2 shipped a bench whose rank 0 issued one more gradient all-reduce after the
other ranks had left (a hang under RCCL) and no one-GPU box could show it; this test can.  The RCCL path itself needs two devices
(tests/test_trainer.py runs it whenever two are visible)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("gpus,extra,scaling,gbatch", [(2, [], "weak", 16), (8, ["--global-batch", "256"], "strong", 256)])
def test_bench_multi_rank_dry_run(gpus, extra, scaling, gbatch):
    """2 ranks (weak / strong) and the shape of BASELINE config 2 itself: 8 ranks, global batch 256 -- so the first
    run on an 8-GPU node only changes the backend."""
    env = dict(os.environ, FFNO_BENCH_ONE_DEVICE="1")
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--batch", "8",
           "--cpu-steps", "0", "--no-secondary", "--n1-steps-per-s", "100"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    # the N > 1 line decomposes itself -- every rank's step times, the collective's own time, efficiency vs N = 1
    dd = d["distributed"]
    assert len(dd["per_rank"]) == gpus and [p["rank"] for p in dd["per_rank"]] == list(range(gpus))
    assert all(0 < p["step_ms_min"] <= p["step_ms_median"] <= p["step_ms_max"] and p["region_wall_ms"] > 0 for p in dd["per_rank"])
    assert dd["allreduce_us"]["samples"] == 2 and 0 < dd["allreduce_us"]["min"] <= dd["allreduce_us"]["max"]
    assert abs(dd["efficiency_vs_n1"] - d["value"] / (gpus * 100.0)) < 1e-3
    assert d["n_gpus"] == gpus and d["scaling"] == scaling and d["steps"] == 2
    assert d["distributed"]["world_size"] == gpus and d["distributed"]["world_size_counted_by_all_reduce"] == gpus
    assert d["distributed"]["backend"] == "gloo"      # (the dry run; under RCCL the line must carry rccl_version: bench.py refuses else)
    assert d["config"]["global_batch"] == gbatch and d["value"] > 0
    for k in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in d


@pytest.mark.gpu
def test_bench_headline_line_carries_every_variant_leg():
    """the all-bf16x3 leg of the headline line raised for a whole round (a stale workspace) and the line only
    said so inside a nested dict.  A short headline run: both arithmetic variants and the bf16 storage variant must be numbers."""
    env = dict(os.environ)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--cpu-steps", "0", "--no-secondary"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    v = d["arithmetic_variants_steps_per_s"]
    assert "error" not in v and len(v) == 2 and all(isinstance(x, (int, float)) and x > 0 for x in v.values()), v
    b = d["bf16_storage_variant"]
    assert b is not None and "error" not in b, b
    assert d["roofline"]["frac"] > 0 and d["config"]["per_gpu_batch"] == 32
