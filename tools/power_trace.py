"""Socket power, shader clock and memory clock of the hot launches and of the whole training / inference step, MEASURED
(rounds 4-5 inferred a power cap from busy-cycle counters; this samples it).

    python tools/power_trace.py [--seconds 5] [--out gpurun_out/r06_power.md]

Sources (an ordinary user can read them on the GPU boxes): /sys/class/drm/card*/device/hwmon/hwmon*/power1_input (socket power, uW),
freq1_input (sclk, Hz), freq2_input (mclk, Hz), power1_cap -- sampled every ~10 ms by a thread while the main thread keeps one
workload running for `--seconds`.  The box shows the sysfs nodes of all 8 GPUs of its host but gives the process one of them: the
card whose power rises under a calibration load is ours.

Workloads, each on REAL operands and -- for the replayed launches -- on ALL-ZERO activations / spectra / gradients (same instruction
stream and addresses: what the data itself costs):
  * the whole training step (markov/24, batch 32) and the forward-only pass (trainer.predict: the inference layer);
  * every hot launch of the step, captured with its arguments (bench.KernelProbe) and replayed back to back;
  * the inference layer's two launches.
Output: a markdown table -- W (mean / max), sclk (mean / min), mclk, us per launch (HIP events over the window), J per launch / step.
"""
import argparse
import glob
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class Sampler:
    def __init__(self, period=0.01):
        self.cards = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.access(os.path.join(d, "power1_input"), os.R_OK):
                self.cards.append(d)
        self.period = period
        self.card = None
        self.rows = []
        self._stop = threading.Event()
        self._thr = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return float("nan")

    def read(self, d):
        return (self._read(os.path.join(d, "power1_input")) * 1e-6, self._read(os.path.join(d, "freq1_input")) * 1e-6,
                self._read(os.path.join(d, "freq2_input")) * 1e-6)

    def pick_card(self, load):
        """Our device's hwmon directory: by PCI address (hipDeviceGetPCIBusId against the sysfs device links); if that fails, the card
        whose power follows an on / off / on / off pattern of `load()` best (the host's other GPUs run other tenants' jobs: a single
        rise proves nothing)."""
        self.how = None
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, torch.cuda.current_device()) == 0:
                addr = buf.value.decode().lower()
                for d in self.cards:
                    if os.path.realpath(d.split("/hwmon/")[0]).lower().endswith(addr):
                        self.card, self.how = d, f"PCI address {addr} (hipDeviceGetPCIBusId)"
        except OSError:
            pass
        score, trace = {d: 0.0 for d in self.cards}, {d: [] for d in self.cards}
        for phase in range(6):          # off, on, off, on, off, on: 1 s each
            on = phase & 1
            t0 = time.time()
            acc = {d: [] for d in self.cards}
            while time.time() - t0 < 1.0:
                if on:
                    load()
                    torch.cuda.synchronize()
                else:
                    time.sleep(0.02)
                for d in self.cards:
                    acc[d].append(self.read(d)[0])
            for d in self.cards:
                m = sum(acc[d]) / len(acc[d])
                trace[d].append(m)
                score[d] += m if on else -m
        if self.card is None:
            self.card, self.how = max(self.cards, key=lambda d: score[d]), "largest on - off power difference over three load / idle cycles"
        self.cap_w = self._read(os.path.join(self.card, "power1_cap")) * 1e-6
        self.idle_w = min(trace[self.card][0::2])
        return self.card, trace

    def start(self):
        self.rows = []
        self._stop.clear()

        def run():
            while not self._stop.is_set():
                self.rows.append((time.time(),) + self.read(self.card))
                time.sleep(self.period)
        self._thr = threading.Thread(target=run)
        self._thr.start()

    def stop(self):
        self._stop.set()
        self._thr.join()
        return self.rows


def window(sampler, fn, seconds, per_call=1, batch=20):
    """Keep `fn` (enqueue-only) running for `seconds`; -> dict(W mean/max, sclk mean/min, mclk, us per call, J per call, samples)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    # ramp the clocks / power state before the window opens
    t0 = time.time()
    while time.time() - t0 < 0.5:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    sampler.start()
    e0.record()
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(batch):
            fn()
        n += batch
        torch.cuda.synchronize()      # (bounds the queue; costs one launch gap per `batch` calls)
    e1.record()
    torch.cuda.synchronize()
    rows = sampler.stop()
    us = 1e3 * e0.elapsed_time(e1) / (n * per_call)
    # drop the first 10 % of the samples (the sampler starts a little before the device is saturated)
    rows = rows[len(rows) // 10:] or rows
    w = [r[1] for r in rows]
    sc = [r[2] for r in rows]
    mc = [r[3] for r in rows]
    mean = lambda v: sum(v) / max(len(v), 1)      # noqa: E731
    return dict(w_mean=mean(w), w_max=max(w), sclk_mean=mean(sc), sclk_min=min(sc), mclk=mean(mc), us=us,
                joule=mean(w) * us * 1e-6, samples=len(rows), calls=n)


def zero_workspace(eng):
    ws_ = getattr(eng, "_ws", None)

    def _zero(v):
        if torch.is_tensor(v):
            if v.is_floating_point():
                v.zero_()
        elif isinstance(v, (list, tuple)):
            for u in v:
                _zero(u)
        elif isinstance(v, dict):
            for u in v.values():
                _zero(u)
    for v in list(vars(ws_).values()):
        _zero(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_power.md"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.trainer import FFNOTrainer
    torch.manual_seed(0)
    block = FNOFactorized2DBlock(**bench.MARKOV24).to(dev)
    trainer = FFNOTrainer(block, lr=2.5e-3, weight_decay=1e-4, num_warmup_steps=500, num_training_steps=100000)
    gen = torch.Generator().manual_seed(1000)
    x = torch.randn(32, 64, 64, 3, generator=gen).to(dev)
    y = torch.randn(32, 64, 64, 1, generator=gen).to(dev)
    for _ in range(5):
        trainer.train_step(x, y)
    torch.cuda.synchronize()

    smp = Sampler()
    card, seen = smp.pick_card(lambda: trainer.train_step(x, y))
    lines = ["# Socket power / clocks of the hot launches and of the step, measured (tools/power_trace.py; MI355X, round 6)", "",
             f"git head `{bench.git_head()}`, library stamp `{bench.lib_source_stamp()}`; sampler: sysfs `{card}` every ~10 ms "
             f"(power1_input = socket power, freq1_input = sclk, freq2_input = mclk), {args.seconds:.0f} s per row; "
             f"power cap (power1_cap) **{smp.cap_w:.0f} W**, idle {smp.idle_w:.0f} W.", "",
             f"Card selection: {smp.how}.  Mean W of every GPU of the host over idle / load / idle / load / idle / load seconds of THIS "
             "process (the other GPUs run other tenants' jobs):", ""]
    for d, tr in seen.items():
        lines.append(f"* `{d.split('/device/')[0].split('/')[-1]}`: " + " / ".join(f"{v:.0f}" for v in tr) + ("  **<- this device**" if d == card else ""))
    lines += ["", "| workload | operands | us / call | W mean | W max | sclk mean MHz | sclk min MHz | mclk MHz | J / call | samples |",
              "|---|---|---|---|---|---|---|---|---|---|"]

    def row(name, kind, r):
        lines.append(f"| {name} | {kind} | {r['us']:.1f} | {r['w_mean']:.0f} | {r['w_max']:.0f} | {r['sclk_mean']:.0f} | {r['sclk_min']:.0f} | "
                     f"{r['mclk']:.0f} | {r['joule']:.3e} | {r['samples']} |")
        bench.log(lines[-1])

    res = {}
    res["step"] = window(smp, lambda: trainer.train_step(x, y), args.seconds)
    row("training step (markov/24, batch 32)", "real", res["step"])
    res["fwd"] = window(smp, lambda: trainer.predict(x), args.seconds)
    row("forward only (trainer.predict: inference layer)", "real", res["fwd"])
    trainer.engine.use_infer_layer = False
    res["fwd_train_path"] = window(smp, lambda: trainer.predict(x), args.seconds)
    row("forward only through the training-path launches", "real", res["fwd_train_path"])
    trainer.engine.use_infer_layer = True

    # capture the hot launches of one training step and of one inference forward
    hot = list(bench.HOT) + ["spectral_mix", "infer_ff"]
    probe = bench.KernelProbe(hot)
    trainer.engine.timer = probe
    probe.capture = True
    trainer.train_step(x, y)
    trainer.predict(x)
    probe.capture = False
    trainer.engine.timer = None
    torch.cuda.synchronize()
    calls = {n: c[len(c) // 2] for n, c in probe.calls.items() if c}
    for kind in ("real", "zero"):
        if kind == "zero":
            # all-zero activations / spectra / gradients in the training workspace AND the inference workspace
            for key, ws in list(trainer.engine.__dict__.get("_ws_cache", {}).items()):
                trainer.engine._ws = ws
                zero_workspace(trainer.engine)
        for n, (fn, a) in calls.items():
            r = window(smp, lambda fn=fn, a=a: fn(*a), args.seconds, batch=400)
            res[(n, kind)] = r
            row(f"`{n}` replayed", kind, r)
    # one untimed step so that nothing keeps the zeroed workspaces
    trainer.train_step(x, y)
    torch.cuda.synchronize()

    lines += ["", "## Reading", ""]
    st = res["step"]
    lines.append(f"* Whole training step: **{st['w_mean']:.0f} W mean / {st['w_max']:.0f} W max of the {smp.cap_w:.0f} W cap**, sclk "
                 f"{st['sclk_mean']:.0f} MHz mean (min {st['sclk_min']:.0f}), {st['us'] / 1e3:.3f} ms -> **{st['joule']:.2f} J / step**.")
    fw = res["fwd"]
    lines.append(f"* Forward only (inference layer): {fw['w_mean']:.0f} W, sclk {fw['sclk_mean']:.0f} MHz, {fw['us'] / 1e3:.3f} ms -> {fw['joule']:.2f} J / forward "
                 f"(training-path launches: {res['fwd_train_path']['us'] / 1e3:.3f} ms, {res['fwd_train_path']['joule']:.2f} J).")
    for n in calls:
        a, b = res[(n, "real")], res[(n, "zero")]
        lines.append(f"* `{n}`: real {a['us']:.1f} us at {a['sclk_mean']:.0f} MHz / {a['w_mean']:.0f} W; zero operands {b['us']:.1f} us at "
                     f"{b['sclk_mean']:.0f} MHz / {b['w_mean']:.0f} W -> time x{a['us'] / b['us']:.3f}, clock x{b['sclk_mean'] / max(a['sclk_mean'], 1):.3f}, "
                     f"energy {a['joule']:.2e} vs {b['joule']:.2e} J.")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
