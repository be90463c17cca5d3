"""Same-box, same-process A/B of the forward-only pass (trainer.predict, markov/24, batch 32) with the inference layers self-ranged
(FFNO_BRANCH_SELF_RANGE: no range words) and on the tensor's range words; interleaved rounds.   python tools/ab_forward.py [rounds]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourierflow_amd.modules import FNOFactorized2DBlock  # noqa: E402
from fourierflow_amd.trainer import FFNOTrainer  # noqa: E402

MARKOV24 = dict(modes=16, width=64, n_layers=24, input_dim=3, share_weight=True, factor=4, ff_weight_norm=True,
                gain=0.1, dropout=0.0, in_dropout=0.0)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    blk = FNOFactorized2DBlock(**MARKOV24).to(dev)
    tr = FFNOTrainer(blk)
    x = torch.randn(32, 64, 64, 3, device=dev)

    def ms(n=50):
        for _ in range(5):
            tr.predict(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.predict(x)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    for r in range(rounds):
        out = []
        for sr in (True, False):
            tr.engine.infer_self_range = sr
            out.append((sr, tr.engine.infer_self_ranged_last if False else None, ms()))
        print(f"round {r}: self-ranged {out[0][2]:.3f} ms   range words {out[1][2]:.3f} ms")


if __name__ == "__main__":
    main()
