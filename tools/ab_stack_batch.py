"""Same-box, same-process A/B of the forward-only pass (trainer.predict, markov/24) over batch sizes: the persistent stack
(ffno_infer_stack: CUs / 8 groups of 8 workgroups, a group per image, groups idle below 32 images and walking several above) vs the
per-layer launches (ffno_layer_infer: row tiles sized to fill the chip); interleaved rounds.
python tools/ab_stack_batch.py [rounds] [batch ...]"""
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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    batches = [int(a) for a in sys.argv[2:]] or [9, 12, 16, 19, 24, 32, 40, 48, 64, 96]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    blk = FNOFactorized2DBlock(**MARKOV24).to(dev)
    tr = FFNOTrainer(blk)
    eng = tr.engine

    def ms(x, n=40):
        for _ in range(4):
            tr.predict(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.predict(x)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    print("# batch: ms per forward (persistent stack | per-layer launches), per round; rel-L2 between the two results")
    for B in batches:
        x = torch.randn(B, 64, 64, 3, device=dev)
        eng.use_infer_stack = True
        ys = tr.predict(x).clone()
        took = bool(eng.infer_stack_last)
        eng.use_infer_stack = False
        yl = tr.predict(x).clone()
        err = float((ys - yl).norm() / yl.norm())
        rows = []
        for _ in range(rounds):
            pair = []
            for on in (True, False):
                eng.use_infer_stack = on
                pair.append(ms(x))
            rows.append(pair)
        eng.use_infer_stack = True
        best = [min(r[i] for r in rows) for i in (0, 1)]
        print(f"B {B:3d}  stack taken {took}  " + "  ".join(f"{a:.3f}|{b:.3f}" for a, b in rows) +
              f"   best {best[0]:.3f} | {best[1]:.3f}  ({100 * (best[0] / best[1] - 1):+.1f} %)   per image {1e3 * best[0] / B:.1f} | {1e3 * best[1] / B:.1f} us   rel-L2 {err:.2e}",
              flush=True)


if __name__ == "__main__":
    main()
