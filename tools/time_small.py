"""What every NON-hot launch of one training step costs ("26 small launches, ~0.2 ms").

Builds the headline trainer (markov/24, batch 32, 64 x 64), captures (entry point, arguments) of every C-ABI call of one step
through the engine's timer hook, and replays each distinct non-hot call back to back between one pair of HIP events.  The
loss / optimiser calls of the trainer are timed the same way.  Replays only re-run a launch on the state the step left behind:
every one of these calls is idempotent on its outputs (the reduces and packs overwrite, AdamW is NOT replayed on the live buffers).

    python tools/time_small.py [batch] [replays]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fourierflow_amd import _capi, _lib  # noqa: E402
from fourierflow_amd.engine import _p  # noqa: E402
from fourierflow_amd.modules import FNOFactorized2DBlock  # noqa: E402
from fourierflow_amd.trainer import FFNOTrainer  # noqa: E402

HOT = {"spectral_fused", "spectral_fused(adj)", "ff_fwd", "ff_bwd_data", "ff_bwd_weights_partial", "fw_grad_partial"}
MARKOV24 = dict(modes=16, width=64, n_layers=24, input_dim=3, share_weight=True, factor=4, ff_weight_norm=True,
                gain=0.1, dropout=0.0, in_dropout=0.0)


class Capture:
    def __init__(self):
        self.calls = []
        self.on = False

    def seen(self, name, fn, args):
        if self.on:
            self.calls.append((name, fn, args))

    def want(self, name):
        return False


def replay(fn, args, n):
    for _ in range(3):
        fn(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn(*args)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    blk = FNOFactorized2DBlock(**MARKOV24).to(dev)
    tr = FFNOTrainer(blk, lr=2.5e-3, weight_decay=1e-4, num_warmup_steps=500, num_training_steps=100000)
    x = torch.randn(B, 64, 64, 3, device=dev)
    y = torch.randn(B, 64, 64, 1, device=dev)
    for _ in range(3):
        tr.train_step(x, y)
    cap = Capture()
    tr.engine.timer = cap
    cap.on = True
    tr.train_step(x, y)
    cap.on = False
    tr.engine.timer = None
    torch.cuda.synchronize()
    lib = _lib.get_lib()
    st = _lib.current_stream(dev)
    rows, seen = [], set()
    total_small = 0.0
    for name, fn, args in cap.calls:
        key = (name, getattr(fn, "__name__", "?"))
        if name in HOT or key in seen:
            continue
        seen.add(key)
        cnt = sum(1 for c in cap.calls if (c[0], getattr(c[1], "__name__", "?")) == key)
        us = replay(fn, args, n)
        rows.append((name, key[1], cnt, us))
        total_small += cnt * us
    # the trainer's own calls
    pred = tr.engine.forward(blk.prepare_input(x), True)
    loss, gy = tr.loss_and_grad(pred, y)
    tr.engine.backward(gy)
    torch.cuda.synchronize()
    Bn = pred.shape[0]
    npx = pred.numel() // Bn
    us = replay(lib.ffno_lploss_fwd_bwd, (_p(pred), _p(y.contiguous()), _p(tr.loss), _p(tr._gy), _p(tr._tmp), Bn, npx, 1.0, None, st), n)
    rows.append(("lploss (2 kernels)", "ffno_lploss_fwd_bwd", 1, us))
    total_small += us
    # AdamW on scratch copies of the flat buffers (same sizes)
    p2, g2, m2, v2 = (torch.zeros_like(tr.pflat) for _ in range(4))
    us = replay(lib.ffno_adamw_flat, (_p(p2), _p(g2), _p(m2), _p(v2), p2.numel(), 1e-3, 0.9, 0.999, 1e-8, 1e-4, 1, 1.0, st), n)
    rows.append(("adamw", "ffno_adamw_flat", 1, us))
    total_small += us
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        loss.clone()
    b.record()
    torch.cuda.synchronize()
    us = 1e3 * a.elapsed_time(b) / n
    rows.append(("loss.clone()", "torch", 1, us))
    total_small += us
    print(f"non-hot launches of one markov/24 training step at batch {B} (replayed {n} x back to back):")
    for name, sym, cnt, us in rows:
        print(f"  {name:28s} {sym:40s} x{cnt:<3d} {us:8.2f} us")
    print(f"  sum over the step: {total_small:.1f} us in {sum(r[2] for r in rows)} calls")


if __name__ == "__main__":
    main()
