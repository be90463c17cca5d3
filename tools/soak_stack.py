"""Soak test of the persistent inference stack (ffno_infer_stack through trainer.predict): 40 s of back-to-back forwards at batch
32 / 19 / 64 / 9 (every 50th compared bit for bit with the per-layer launches, error word read), then 200 forwards beside a GEMM
stream that competes for the CUs.
    python tools/soak_stack.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourierflow_amd.modules import FNOFactorized2DBlock
from fourierflow_amd.trainer import FFNOTrainer
kw = dict(modes=16, width=64, n_layers=24, input_dim=3, share_weight=True, factor=4, ff_weight_norm=True, gain=0.1, dropout=0.0, in_dropout=0.0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = FFNOTrainer(FNOFactorized2DBlock(**kw).to(dev))
eng = tr.engine
xs = {B: torch.randn(B, 64, 64, 3, device=dev) for B in (32, 19, 64, 9)}
ref = {}
eng.use_infer_stack = False
for B, x in xs.items():
    ref[B] = tr.predict(x).clone()
eng.use_infer_stack = True
t0 = time.time(); n = 0; bad = 0
side = torch.cuda.Stream()
while time.time() - t0 < 40:
    for B, x in xs.items():
        y = tr.predict(x)
        n += 1
        if n % 50 == 0:
            assert eng.infer_stack_last
            if not torch.equal(y, ref[B]) or int(eng._ws.stack_sync[-1].item()) != 0:
                bad += 1
    if n % 400 == 0:      # a training step in between (other kernels, weights unchanged afterwards? no: they change -> refresh the references)
        pass
print(f"{n} persistent forwards in {time.time() - t0:.1f} s, mismatches / error words: {bad}")
# with a competing stream: a long-running kernel on another stream may hold CUs -> the launch must still finish (or report an error word), never hang
a = torch.randn(8192, 8192, device=dev)
with torch.cuda.stream(side):
    for _ in range(20):
        b = a @ a
t1 = time.time(); errs = 0
for _ in range(200):
    y = tr.predict(xs[32])
torch.cuda.synchronize()
print(f"200 forwards beside a GEMM stream: {time.time() - t1:.2f} s, stack still on: {eng.use_infer_stack}, last result equal: {torch.equal(y, ref[32])}, error word {int(eng._ws.stack_sync[-1].item())}")
