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
@pytest.mark.parametrize("extra,scaling", [([], "weak"), (["--global-batch", "16"], "strong")])
def test_bench_two_ranks_dry_run(extra, scaling):
    env = dict(os.environ, FFNO_BENCH_ONE_DEVICE="1")
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--cpu-steps", "0", "--no-secondary"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 2
    assert d["distributed"]["world_size"] == 2 and d["distributed"]["world_size_counted_by_all_reduce"] == 2
    assert d["config"]["global_batch"] == 16 and d["value"] > 0
    for k in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in d
