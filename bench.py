#!/usr/bin/env python3
"""Benchmark of the F-FNO hot path on MI355X: training steps/s of torus_li/markov/24_layers.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One *step* = forward + relative-L2 loss + backward + (gradient all-reduce over RCCL when N > 1) +
fused AdamW/cosine update on one batch of synthetic N(0,1) inputs already resident in HBM -- the
reference's `Grid2DMarkovExperiment._training_step` + `optimize_manually`
(routines/grid_2d_markov.py:172-193, routines/base.py:27-52) for the model of
experiments/torus_li/markov/24_layers/config.yaml (modes 16, width 64, 24 layers, shared Fourier
weights, factor 4, weight-norm), fp32, per-GPU batch 32 (weak scaling: global batch = 32 * N).

Prints ONE JSON line (rank 0).  `value` = whole-node training steps/s = N * K / T, where every rank
runs K steps of per-GPU batch 32 and T is the max over ranks of the barrier-bracketed wall time.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.perf_counter()

MARKOV24 = dict(modes=16, width=64, n_layers=24, input_dim=3, share_weight=True, factor=4, ff_weight_norm=True,
                gain=0.1, dropout=0.0, in_dropout=0.0)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
# split-bf16 kernels (ffx.hip): one fp32-accurate multiply-add = 6 bf16 MFMA multiply-adds, so their matrix-core ceiling in
# ALGORITHMIC (fp32-equivalent) FLOP/s is the dense bf16 peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s) / 6
BF16X3_PEAK_TFLOPS = 2500.0 / 6.0
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak


class KernelTimer:
    """HIP-event timing of selected launches on the stream they are enqueued on (torch's current stream)."""

    def __init__(self, names, every=1):
        self.names = set(names)
        self.pairs = {n: [] for n in names}
        self._open = {}
        self.enabled = False
        self.count = {n: 0 for n in names}
        self.every = every

    def want(self, name):
        if not self.enabled or name not in self.names:
            return False
        self.count[name] += 1
        return self.count[name] % self.every == 0

    def start(self, name, stream=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream) if stream is not None else ev.record()
        self._open[name] = ev

    def stop(self, name, stream=None):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream) if stream is not None else ev.record()
        self.pairs[name].append((self._open.pop(name), ev))

    def calibrate(self, n=50):
        """Elapsed time of an EMPTY start/stop pair on the same stream: what the event bracket itself adds to a launch
        (subtracted in summary(), so the per-launch times line up with rocprofv3's kernel-trace durations)."""
        prs = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            b.record()
            prs.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in prs)
        # a bracket around a real kernel hides part of that latency behind the kernel: 0.6 x the empty-pair time is what
        # reproduces rocprofv3's kernel-trace averages (profiles/r01_v6_kernel_stats.md: 38.6 us vs 42.0 us raw, pair 5.3 us)
        self.overhead_us = 0.6 * 1e3 * ms[len(ms) // 2]
        return self.overhead_us

    def summary(self):
        out = {}
        ov = getattr(self, "overhead_us", 0.0)
        for n, prs in self.pairs.items():
            if prs:
                ms = [a.elapsed_time(b) for a, b in prs]
                out[n] = dict(avg_us=max(1e3 * sum(ms) / len(ms) - ov, 0.0), samples=len(ms), launches_per_sample=self.every,
                              event_overhead_us=round(ov, 2))
        return out


FFX_KERNELS = ("ff_fwd", "ff_bwd_data", "ff_bwd_weights_partial")   # on the split-bf16 path when engine._ffx()
X3_ALWAYS = ("fw_grad_partial",)                                   # split-bf16 unconditionally


def algorithmic_work(P, C, H, K, B, M, N, L, paired=True):
    """Algorithmic FLOPs / HBM bytes per launch of the timed kernels and launches per train step.
    fp32: 4 B per element.  R = lines per axis; spectra are K*R*2C floats."""
    R = B * M
    spec = 4.0 * K * R * 2 * C
    act = 4.0 * P * C
    dft = 2.0 * R * (2 * K) * N * C            # truncated DFT (or its inverse) of R lines as a [2K x N].[N x C] product
    mix = 8.0 * R * K * C * C                   # complex per-mode channel mix
    if paired:
        # both axes of a layer in one launch (engine "concurrent_branches"): read x once, write the two branch outputs,
        # save the two spectra (training); the adjoint also reads the residual gradient
        sf = dict(flops=2 * (2 * dft + mix), bytes=3 * act + 2 * spec, per_step=L, bound="hbm")
        sa = dict(flops=2 * (2 * dft + mix), bytes=4 * act + 2 * spec, per_step=L, bound="hbm")
    else:
        # one spectral branch: read x, write s (+ read s when accumulating the 2nd branch), + save the spectrum (training)
        sf = dict(flops=2 * dft + mix, bytes=(act + act + spec) + 0.5 * act, per_step=2 * L, bound="hbm")
        sa = dict(flops=2 * dft + mix, bytes=(act + act + act + spec), per_step=2 * L, bound="hbm")
    return {
        "spectral_fused": sf,
        "spectral_fused(adj)": sa,
        # the same two branches through the three paired STAGE launches (K > 16: 256 x 256 grids); one "launch" = the three
        # kernels of a ffno_spectral_staged_pair call, algorithmic work as above (the spectra that travel are overhead)
        "spectral_staged_pair": sf,
        "spectral_staged_pair(adj)": sa,
        # split-bf16 feed-forward (ffx.hip): no hidden activations in HBM, only the ReLU sign bits (P*H/8 bytes); the
        # weight-gradient kernel recomputes h and dh, so its algorithmic FLOPs are 4 GEMMs (2 recomputed + 2 gradients)
        # (+ with paired branches: the second branch buffer is read and the sum written back while staging)
        "ff_fwd": dict(flops=4.0 * P * C * H, bytes=(5 if paired else 3) * act + P * H / 8, per_step=L, bound="mfma"),
        "ff_bwd_data": dict(flops=4.0 * P * C * H, bytes=(4 if paired else 2) * act + P * H / 8, per_step=L, bound="mfma"),
        "ff_bwd_weights_partial": dict(flops=8.0 * P * C * H, bytes=2 * act, per_step=L, bound="mfma"),
        "fw_grad_partial": dict(flops=L * mix, bytes=2 * L * spec, per_step=2, bound="hbm"),
    }


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(batch, grid, steps, kw):
    """The oracle (op-for-op CPU torch restatement of the reference, certified against its golden
    vectors) timed on this box's host cores: same model, batch and synthetic data distribution.
    The intra-op thread count is calibrated first (all cores is NOT the fastest on a 256-core box)."""
    from oracle import ffno_oracle as orc
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sd = orc.init_block_state_dict(modes=kw["modes"], width=kw["width"], input_dim=kw["input_dim"],
                                   n_layers=kw["n_layers"], share_weight=kw["share_weight"], factor=kw["factor"],
                                   ff_weight_norm=kw["ff_weight_norm"], gain=kw["gain"], seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, grid, grid, kw["input_dim"], generator=g)
    y = torch.randn(batch, grid, grid, 1, generator=g)

    def fwd(xb):
        with torch.no_grad():
            t0 = time.perf_counter()
            orc.ffno2d_block(sd, xb, modes=kw["modes"], n_layers=kw["n_layers"])
            return time.perf_counter() - t0

    best_t, best_n = None, None
    for nthr in sorted({n for n in (8, 16, 32, 64, 128, ncpu) if n <= ncpu}):
        torch.set_num_threads(nthr)
        fwd(x[:4])
        t = fwd(x[:4])
        log(f"cpu baseline calibration: {nthr} threads -> {1e3 * t:.0f} ms / forward(B=4)")
        if best_t is None or t < best_t:
            best_t, best_n = t, nthr
        if t > 3 * best_t:
            break
    torch.set_num_threads(best_n)
    uniq, seen = [], set()
    for k, v in sd.items():
        if id(v) not in seen:
            seen.add(id(v))
            uniq.append(v.requires_grad_(True))
    opt = torch.optim.AdamW(uniq, lr=2.5e-3, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: orc.cosine_warmup_factor(s, 500, 100000, 0.5))

    def step():
        t0 = time.perf_counter()
        opt.zero_grad()
        out = orc.ffno2d_block(sd, x, modes=kw["modes"], n_layers=kw["n_layers"])["forecast"]
        loss = orc.lp_rel_loss(out, y)
        loss.backward()
        opt.step()
        sch.step()
        return time.perf_counter() - t0

    t_warm = step()
    log(f"cpu baseline warm-up step: {t_warm:.2f} s ({best_n} threads)")
    times = []
    for _ in range(steps):
        times.append(step())
        if sum(times) > 60:
            break
    dt = sum(times) / len(times)
    f = fwd(x)
    return dict(value=round(1.0 / dt, 4), unit="steps/s", cores=best_n, kind="port", host_cores_available=ncpu,
                s_per_step=round(dt, 3), ms_per_forward=round(1e3 * f, 1),
                sample=f"oracle (CPU torch restatement of the reference op sequence, pinned to its golden vectors) "
                       f"train step, same model/batch ({batch}, {grid}x{grid}, fp32): 1 warm-up + {len(times)} timed "
                       f"steps on {best_n} threads (best of a thread-count calibration)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--modes", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed CPU-baseline steps (0 disables)")
    ap.add_argument("--no-x3", action="store_true", help="spectral branches on the fp32-MFMA kernel instead of the split-bf16 one")
    ap.add_argument("--x3-interleave", type=int, default=1, help="bit 0: workgroup->branch interleave; bits 8..: start skew / 256 cycles")
    ap.add_argument("--ffx-schedule", type=int, default=1, help="bit mask: 1 forward, 2 backward-data, 4 weight gradients on the role-split schedule (default 1)")
    ap.add_argument("--plus", action="store_true", help="FNOPlus2DBlock (non-factorized ablation) instead of the F-FNO block; "
                                                       "no roofline / CPU baseline for this secondary workload")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from fourierflow_amd.modules import FNOFactorized2DBlock, FNOPlus2DBlock
    from fourierflow_amd.trainer import FFNOTrainer

    kw = dict(MARKOV24, n_layers=args.layers, modes=args.modes)
    torch.manual_seed(0)  # same initial weights on every rank (and broadcast from rank 0 anyway)
    block = (FNOPlus2DBlock if args.plus else FNOFactorized2DBlock)(**kw).to(dev)
    trainer = FFNOTrainer(block, lr=2.5e-3, weight_decay=1e-4, num_warmup_steps=500, num_training_steps=100000)
    from fourierflow_amd import _lib as _fl
    _fl.get_lib().ffno_ffx_set_schedule(args.ffx_schedule)
    trainer.engine.use_x3 = not args.no_x3
    trainer.engine.x3_interleave = args.x3_interleave
    B, G = args.batch, args.grid
    gen = torch.Generator().manual_seed(1000 + rank)  # rank r draws its own shard of the global batch
    x = torch.randn(B, G, G, kw["input_dim"], generator=gen).to(dev)
    y = torch.randn(B, G, G, 1, generator=gen).to(dev)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log(f"model + trainer ready on {dev}; warm-up {args.warmup} steps")
    for _ in range(args.warmup):
        trainer.train_step(x, y)
    torch.cuda.synchronize()
    if rank == 0:
        log("warm-up done; timing")
    names = ["spectral_fused", "spectral_fused(adj)", "spectral_staged_pair", "spectral_staged_pair(adj)", "ff_fwd", "ff_bwd_data", "ff_bwd_weights_partial", "fw_grad_partial"]
    timer = KernelTimer(names, every=7) if rank == 0 else None   # sample 1 launch in 7 (odd: both branches get sampled)
    trainer.engine.timer = timer
    if timer:
        timer.enabled = True
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(x, y)
    sync()
    elapsed = time.perf_counter() - t0
    if timer:
        timer.enabled = False
        timer.calibrate()
    trainer.engine.timer = None
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(tmax.item())
    if rank == 0:
        log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    loss_val = float(loss.item())

    # forward-only latency (the reference's `infer` path), same batch
    for _ in range(2):
        trainer.predict(x)
    sync()
    t1 = time.perf_counter()
    nf = max(5, args.steps)
    for _ in range(nf):
        trainer.predict(x)
    sync()
    ms_fwd = 1e3 * (time.perf_counter() - t1) / nf
    # ... and at batch 1 ("also report B=1 latency"; what an autoregressive rollout pays per model step)
    ms_fwd_b1 = None
    if rank == 0:
        x1 = x[:1].contiguous()
        for _ in range(3):
            trainer.predict(x1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            trainer.predict(x1)
        torch.cuda.synchronize()
        ms_fwd_b1 = 1e3 * (time.perf_counter() - t1) / 20
        trainer.predict(x)      # back to the timed geometry (keeps `paired_last` describing the timed workload)

    if rank == 0:
        log(f"forward-only: {ms_fwd:.3f} ms")
        P = B * G * G
        C, H, K = kw["width"], kw["width"] * kw["factor"], kw["modes"]
        paired = bool(getattr(trainer.engine, "paired_last", False))
        work = algorithmic_work(P, C, H, K, B, G, G, args.layers, paired)
        ksum = timer.summary()
        pmc = {}
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if B == 32 and G == 64 and args.layers == 24 and K == 16:
                pmc = pj["kernels"]
        except Exception:  # noqa: BLE001 - the PMC summary is optional evidence
            pass
        kernels = {}
        for n, srow in ksum.items():
            w = work[n]
            us = srow["avg_us"]
            mpeak = BF16X3_PEAK_TFLOPS if ((n in FFX_KERNELS and trainer.engine._ffx()) or n in X3_ALWAYS) else FP32_MFMA_PEAK_TFLOPS
            kernels[n] = dict(avg_us=round(us, 2), per_step=w["per_step"], ms_per_step=round(us * w["per_step"] * 1e-3, 3),
                              tflops=round(w["flops"] / us * 1e-6, 2), gbs=round(w["bytes"] / us * 1e-3, 1),
                              mfma_peak_tflops=round(mpeak, 1), frac_mfma=round(w["flops"] / us * 1e-6 / mpeak, 3),
                              frac_hbm=round(w["bytes"] / us * 1e-3 / HBM_PEAK_GBS, 3), samples=srow["samples"])
        # dominant kernel = the kernel symbol with the largest share of the step (spectral_fused fwd+adj are one symbol)
        share = {}
        for n, kr in kernels.items():
            share[n.split("(")[0]] = share.get(n.split("(")[0], 0.0) + kr["ms_per_step"]
        dom_sym = max(share, key=share.get) if share else None
        roofline = None
        if dom_sym:
            members = [n for n in kernels if n.split("(")[0] == dom_sym]
            us = sum(kernels[n]["avg_us"] * kernels[n]["per_step"] for n in members) / sum(kernels[n]["per_step"] for n in members)
            fl = sum(work[n]["flops"] * work[n]["per_step"] for n in members) / sum(work[n]["per_step"] for n in members)
            by = sum(work[n]["bytes"] * work[n]["per_step"] for n in members) / sum(work[n]["per_step"] for n in members)
            bound = work[members[0]]["bound"]
            if bound == "hbm":
                ach, peak, unit = by / us * 1e-3, HBM_PEAK_GBS, "GB/s"
            else:
                ach, peak, unit = fl / us * 1e-6, kernels[members[0]]["mfma_peak_tflops"], "TFLOP/s"
            traffic = pmc.get(dom_sym, {}).get("hbm_bytes_per_launch")
            roofline = dict(kernel=dom_sym, bound=bound, achieved=round(ach, 2), peak=peak, unit=unit,
                            frac=round(ach / peak, 4), traffic=traffic, avg_launch_us=round(us, 2),
                            share_of_step=round(share[dom_sym] / (1e3 * elapsed / args.steps), 3),
                            algorithmic_bytes_per_launch=int(by), algorithmic_flops_per_launch=int(fl),
                            event_overhead_us=round(timer.overhead_us, 2),
                            note="achieved = algorithmic bytes (or FLOPs) per launch / mean HIP-event launch time in the timed "
                                 "region (minus 0.6 x the measured time of an empty event pair, the share a bracket adds around a kernel); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/pmc_traffic.json "
                                 "(separate rocprofv3 --pmc passes, gfx950 x2 correction on FETCH_SIZE)")
        cpu = None
        if args.plus:
            roofline = None
        if world == 1 and args.cpu_steps > 0 and not args.plus:
            cpu = cpu_baseline(B, G, args.cpu_steps, kw)
        steps_per_s = world * args.steps / elapsed
        out = {
            "metric": ("training-steps/sec (whole node), FNOPlus2DBlock %dL %dx%d modes %d" % (args.layers, G, G, K) if args.plus else
                       "training-steps/sec (whole node), F-FNO 24L 64x64 torus (torus_li/markov/24_layers)"
                       if (G, args.layers, K) == (64, 24, 16) else
                       "training-steps/sec (whole node), F-FNO %dL %dx%d modes %d" % (args.layers, G, G, K)),
            "value": round(steps_per_s, 3), "unit": "steps/s (per-GPU batch %d, summed over ranks)" % B,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic N(0,1) inputs/targets, reference-init weights",
            "config": {"workload": "%s train step: FNOFactorized2DBlock(modes=%d,width=64,"
                                   "n_layers=%d,input_dim=3,share_weight,factor=4,weight_norm) %dx%d, fp32"
                                   % ("FNOPlus2DBlock (non-factorized ablation)" if args.plus else
                                      "torus_li/markov/24_layers" if (G, args.layers, K) == (64, 24, 16) else "F-FNO",
                                      K, args.layers, G, G),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "optimizer": "AdamW(lr 2.5e-3, wd 1e-4) + cosine warm-up, fused flat kernel",
                       "collective": "1 x all_reduce(flat fp32 grads, %d floats) / step" % trainer.pflat.numel()},
            "samples_per_s": round(steps_per_s * B, 1), "ms_per_forward": round(ms_fwd, 3),
            "ms_per_forward_batch1": round(ms_fwd_b1, 3),
            "final_loss": round(loss_val, 5),
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(steps_per_s / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
