"""Training step (flat AdamW + cosine warm-up + relative-L2 loss) against the reference's golden
training run, and the data-parallel path (world_size 2, gloo on CPU through the emulator backend)."""
import os
import sys

import numpy as np
import pytest
import torch

import golden_util as gu
from backend_util import emu_lib, host_device, rel_l2  # noqa: F401


def make_trainer(kw, seed, device, **tkw):
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.trainer import FFNOTrainer
    blk = FNOFactorized2DBlock(**kw)
    blk.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in gu.make_block_state_dict(kw, seed).items()})
    blk = blk.to(device)
    return blk, FFNOTrainer(blk, lr=2.5e-3, weight_decay=1e-4, num_warmup_steps=2, num_training_steps=10,
                            num_cycles=0.5, **tkw)


def test_train_steps_match_reference_golden(host_device):
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, steps = [int(v) for v in g["meta"]]
    blk, tr = make_trainer(kw, seed, host_device)
    for s in range(steps):
        x_np, t_np = gu.make_block_io(kw, seed + 1 + s, B, M, N)
        assert abs(tr.current_lr() - float(g["lrs"][s])) < 1e-12
        loss = tr.train_step(torch.from_numpy(x_np).to(host_device), torch.from_numpy(t_np).to(host_device))
        assert abs(loss.item() - float(g["losses"][s])) < 2e-5 * max(1.0, float(g["losses"][s]))
    named = dict(blk.named_parameters())
    for n in [k for k in gu.packed_names(g) if k.startswith("final.")]:
        err = gu.compare_packed(g, n, named[n[6:]].detach().cpu().numpy(), 1e-5)
        assert err < 2e-4, (n, err)   # 2 AdamW updates amplify fp32 rounding of tiny gradients (m/sqrt(v))
    # module parameters alias the flat buffer (one fused optimiser launch updates all of them)
    assert all(p.data_ptr() >= tr.pflat.data_ptr() and
               p.data_ptr() < tr.pflat.data_ptr() + tr.pflat.numel() * 4 for p in blk.parameters())


@pytest.mark.parametrize("deferred,lazy,sched,layer_calls", [(False, "0", 0, True), (True, "0", 3, True), (True, "s", 2, True),
                                                             (True, "s", 3, False), (True, "s", 0, False)])
def test_train_steps_match_golden_under_every_backward_schedule(host_device, deferred, lazy, sched, layer_calls):
    """The variants of the backward pass that round 4 added -- feed-forward weight gradients per layer or of ALL layers in one launch
    after the loop (engine.ff_wgrad_deferred), the chain launches writing their input sums or leaving the forward's ("s")
    to that launch (engine.ff_lazy_sums), shared-tile or wave-tile chain kernels (engine.ff_schedule), one C call per layer
    or one per kernel -- all reproduce the reference's golden training steps (losses 2e-5, final weights 2e-4)."""
    if host_device == "cpu" and (deferred, lazy, sched, layer_calls) in ((True, "0", 3, True), (True, "s", 3, False)):
        pytest.skip("emulator time budget (the GPU run covers all five)")
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, steps = [int(v) for v in g["meta"]]
    blk, tr = make_trainer(kw, seed, host_device)
    eng = tr.engine
    eng.ff_wgrad_deferred, eng.ff_lazy_sums, eng.ff_schedule, eng.use_layer_calls = deferred, lazy, sched, layer_calls
    for s in range(steps):
        x_np, t_np = gu.make_block_io(kw, seed + 1 + s, B, M, N)
        loss = tr.train_step(torch.from_numpy(x_np).to(host_device), torch.from_numpy(t_np).to(host_device))
        assert abs(loss.item() - float(g["losses"][s])) < 2e-5 * max(1.0, float(g["losses"][s]))
    ws = eng._ws
    assert bool(getattr(ws, "defer_wgrad", False)) == deferred and (getattr(ws, "lazy_sums", "") or "0") == (lazy if deferred else "0")
    assert bool(getattr(eng, "_saved_lazy", False)) == (deferred and lazy != "0")
    assert (ws.wg_jobs is not None and len(ws.wg_jobs) == kw["n_layers"]) == deferred
    named = dict(blk.named_parameters())
    for n in [k for k in gu.packed_names(g) if k.startswith("final.")]:
        err = gu.compare_packed(g, n, named[n[6:]].detach().cpu().numpy(), 1e-5)
        assert err < 2e-4, (n, err)


def test_arithmetic_switched_on_a_live_trainer_equals_a_fresh_engine(host_device):
    """ADVICE r04 (high) / VERDICT r04 weak #2: `ff_split`, `x3_mix_split`, `ff_wgrad_deferred`, `ff_lazy_sums` are plain attributes,
    and the workspace used to freeze what it derived from them at creation -- switching `ff_split` to "bf16x3" on a trainer that had
    already stepped sent bf16x3 packs into the deferred fp16x2 weight-gradient launch (wrong gradients, no error), switching
    `x3_mix_split` as well raised AttributeError (bench.py's variant leg did exactly that).  They are part of the workspace key now:
    train two steps, switch both to bf16x3, step, switch back, step == the same sequence on engines that were never switched."""
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, _ = [int(v) for v in g["meta"]]
    data = [gu.make_block_io(kw, seed + 1 + s, B, M, N) for s in range(4)]
    to = lambda a: torch.from_numpy(a).to(host_device)      # noqa: E731

    h, b = ("fp16x2", "fp16x2"), ("bf16x3", "bf16x3")
    schedule = [h, h, b, h]
    blk, tr = make_trainer(kw, seed, host_device)
    snaps, losses, grads = [], [], []
    for (ff, mix), (x_np, t_np) in zip(schedule, data):
        snaps.append((tr.pflat.clone(), tr.m.clone(), tr.v.clone(), tr.step_count, tr.opt_step))
        tr.engine.ff_split, tr.engine.x3_mix_split = ff, mix
        losses.append(tr.train_step(to(x_np), to(t_np)).item())
        grads.append(tr.engine.gflat.detach().cpu().numpy().copy())
    assert tr.engine._ws.defer_wgrad                  # back on the deferred fp16x2 launch after the round trip
    for k in (2, 3):
        # the reference for step k: an engine that has never run anything but step k's arithmetic, started from the live
        # trainer's state before that step
        fresh_blk, fresh = make_trainer(kw, seed, host_device)
        pf, m, v, sc, oc = snaps[k]
        fresh.pflat.copy_(pf)
        fresh.m.copy_(m)
        fresh.v.copy_(v)
        fresh.step_count, fresh.opt_step = sc, oc
        fresh.engine.weights_changed()
        fresh.engine.ff_split, fresh.engine.x3_mix_split = schedule[k]
        loss = fresh.train_step(to(data[k][0]), to(data[k][1])).item()
        assert loss == losses[k], (k, loss, losses[k])
        np.testing.assert_array_equal(fresh.engine.gflat.detach().cpu().numpy(), grads[k])
    assert bool(fresh.engine._ws.defer_wgrad) and len(tr.engine._ws_cache) >= 2      # (one workspace per arithmetic)


def test_backward_uses_the_workspace_its_forward_filled(host_device):
    """the workspace key carries the schedule attributes, and backward() used to re-derive its workspace from
    the CURRENT attributes -- a switch between forward(save_for_backward=True) and backward() handed it a freshly allocated
    workspace (uninitialised saved tensors, sign bits, range words: garbage gradients, no error).  The pass now keeps the
    workspace object it filled: an unchanged engine gets exactly that object (even after another geometry went through the
    cache in between), a changed one is refused."""
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, _ = [int(v) for v in g["meta"]]
    blk, tr = make_trainer(kw, seed, host_device)
    eng = tr.engine
    x_np, t_np = gu.make_block_io(kw, seed + 1, B, M, N)
    x = torch.from_numpy(x_np).to(host_device)
    out = eng.forward(x, True)
    ws = eng._saved_ws
    assert ws is eng._ws
    gy = torch.ones_like(out)
    ref = eng.backward(gy).clone()
    # a validation-sized forward in between moves the engine's current workspace; the saved one is still the forward's
    out = eng.forward(x, True)
    ws = eng._saved_ws
    eng._workspace(1, (M, N), False)
    assert eng._ws is not ws
    again = eng.backward(gy)
    assert eng._saved_ws is ws
    np.testing.assert_array_equal(again.cpu().numpy(), ref.cpu().numpy())
    # an arithmetic switch between the two passes: refused (it used to differentiate uninitialised memory)
    eng.forward(x, True)
    eng.ff_split = "bf16x3"
    with pytest.raises(RuntimeError, match="changed between forward"):
        eng.backward(gy)
    eng.ff_split = "fp16x2"
    np.testing.assert_array_equal(eng.backward(gy).cpu().numpy(), ref.cpu().numpy())


def test_deferred_launch_falls_back_beyond_its_memory_limit(host_device, monkeypatch):
    """the all-layers weight-gradient launch keeps one gradient buffer per layer (and, with lazy sums, both branch
    outputs of every layer).  Beyond FFNO_FF_DEFER_MAX_BYTES the engine keeps the ping-pong pair and per-layer launches -- same
    golden losses."""
    monkeypatch.setenv("FFNO_FF_DEFER_MAX_BYTES", "1")
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, steps = [int(v) for v in g["meta"]]
    blk, tr = make_trainer(kw, seed, host_device)
    for s in range(steps):
        x_np, t_np = gu.make_block_io(kw, seed + 1 + s, B, M, N)
        loss = tr.train_step(torch.from_numpy(x_np).to(host_device), torch.from_numpy(t_np).to(host_device))
        assert abs(loss.item() - float(g["losses"][s])) < 2e-5 * max(1.0, float(g["losses"][s]))
    ws = tr.engine._ws
    assert not ws.defer_wgrad and len(ws.G) == 2 and ws.wg_jobs is None and not getattr(ws, "lazy_sums", "")


def test_load_state_dict_on_a_trainer_bound_module_is_seen(host_device):
    """FFNOTrainer re-points the module's parameters at its flat buffer (``p.data = view``), which keeps every
    Parameter's own version counter -- ``load_state_dict`` / ``p.copy_()`` bump that counter only, and an engine bound to the
    views kept the weight-norm products, packs and folded head of the OLD weights (rel. error ~0.8).  The engine now binds
    aliases that share the Parameters' counters: train -> validate -> load best.ckpt -> test in one process is right."""
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, _ = [int(v) for v in g["meta"]]
    blk, tr = make_trainer(kw, seed, host_device)
    x_np, t_np = gu.make_block_io(kw, seed + 1, B, M, N)
    x, t = torch.from_numpy(x_np).to(host_device), torch.from_numpy(t_np).to(host_device)
    tr.train_step(x, t)
    before = tr.predict(x).cpu().numpy()
    other = {k: torch.from_numpy(v.copy()) for k, v in gu.make_block_state_dict(kw, seed + 5).items()}
    blk.load_state_dict(other)                                      # in place, through the Parameters
    after = tr.predict(x).cpu().numpy()
    after_module = blk(x)["forecast"].detach().cpu().numpy()
    fresh_blk, fresh_tr = make_trainer(kw, seed + 5, host_device)
    want = fresh_tr.predict(x).cpu().numpy()
    assert rel_l2(before, want) > 1e-2                               # (the two weight sets really differ)
    np.testing.assert_array_equal(after, want)
    np.testing.assert_array_equal(after_module, want)
    # and a single parameter overwritten in place
    with torch.no_grad():
        dict(blk.named_parameters())["out.1.bias"].add_(0.5)
    np.testing.assert_allclose(tr.predict(x).cpu().numpy(), want + 0.5, rtol=0, atol=1e-6)
    # the next training step still starts from the loaded weights (its forward re-derives the operands)
    l1 = tr.train_step(x, t).item()
    fresh_dict = dict(fresh_blk.named_parameters())
    with torch.no_grad():
        fresh_dict["out.1.bias"].add_(0.5)
    assert abs(l1 - fresh_tr.train_step(x, t).item()) < 1e-6


def test_parameters_made_under_inference_mode_still_run(host_device):
    """inference tensors have no version counter (``t._version`` raises); the derived operands are then simply
    rebuilt by every forward."""
    from fourierflow_amd.modules import FNOFactorized2DBlock
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, _ = [int(v) for v in g["meta"]]
    sd = {k: torch.from_numpy(v.copy()) for k, v in gu.make_block_state_dict(kw, seed).items()}
    x = torch.from_numpy(gu.make_block_io(kw, seed + 1, B, M, N)[0]).to(host_device)
    ref = FNOFactorized2DBlock(**kw)
    ref.load_state_dict(sd)
    with torch.no_grad():
        want = ref.to(host_device)(x)["forecast"].cpu().numpy()
    with torch.inference_mode():
        blk = FNOFactorized2DBlock(**kw)
        blk.load_state_dict(sd)
        blk = blk.to(host_device)
        got = blk(x)["forecast"].cpu().numpy()
        got2 = blk(x)["forecast"].cpu().numpy()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got2, want)


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from fourierflow_amd import _lib
    _lib._install_test_backend(emu_lib())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, steps = [int(v) for v in g["meta"]]
    # rank r deliberately starts from DIFFERENT weights: the rank-0 broadcast must fix that
    blk, tr = make_trainer(kw, seed + 7 * rank, "cpu")
    losses = []
    for s in range(steps):
        x_np, t_np = gu.make_block_io(kw, seed + 1 + s, B, M, N)
        sl = slice(rank * B // world, (rank + 1) * B // world)
        loss = tr.train_step(torch.from_numpy(x_np[sl].copy()), torch.from_numpy(t_np[sl].copy()))
        losses.append(loss.item())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pflat=tr.pflat.numpy(), losses=np.array(losses))
    dist.destroy_process_group()


@pytest.mark.emu
def test_ddp_two_ranks_gloo_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["pflat"], r1["pflat"])          # replicas stay bit-identical
    g = gu.load_golden("train_c64_2l")
    # mean of the two per-rank losses == the reference's full-batch loss (LpLoss.rel is a per-sample mean)
    np.testing.assert_allclose((r0["losses"] + r1["losses"]) / 2, g["losses"], rtol=2e-5)
    # and the final weights equal the reference's single-process run on the full batch
    from fourierflow_amd import _lib
    _lib._install_test_backend(emu_lib())
    try:
        kw = gu.golden_kwargs(g)
        blk, tr = make_trainer(kw, int(g["meta"][3]), "cpu")
        names = [n for n, _ in blk.engine_parameters()]
        off = 0
        for n in names:
            shape = blk.engine().param_shapes[n]
            cnt = int(np.prod(shape))
            got = r0["pflat"][off:off + cnt].reshape(shape)
            off += cnt
            assert gu.compare_packed(g, "final." + n, got, 1e-5) < 2e-4, n
    finally:
        _lib._install_test_backend(None)


@pytest.mark.gpu
def test_training_learns_a_spectral_operator_on_gpu():
    """End-to-end sanity beyond per-step parity: 150 fused train steps of a 4-layer F-FNO on a learnable target (a fixed
    low-pass + shift of the input field) must cut the relative-L2 loss by more than half, with finite weights."""
    from fourierflow_amd.modules import FNOFactorized2DBlock
    from fourierflow_amd.trainer import FFNOTrainer
    torch.manual_seed(0)
    dev = "cuda:0"
    blk = FNOFactorized2DBlock(modes=8, width=32, n_layers=4, input_dim=3, share_weight=True, factor=4,
                               ff_weight_norm=True, gain=0.1).to(dev)
    tr = FFNOTrainer(blk, lr=2.5e-3, weight_decay=1e-4, num_warmup_steps=10, num_training_steps=1000)
    B, G = 8, 32
    g = torch.Generator().manual_seed(1)
    ticks = torch.linspace(0, 1, G)
    pos = torch.stack(torch.meshgrid(ticks, ticks, indexing="ij"), dim=-1)[None].expand(B, G, G, 2)

    def batch():
        w = torch.randn(B, G, G, 1, generator=g)
        wf = torch.fft.rfft2(w[..., 0])
        wf[:, 5:-4] = 0
        wf[:, :, 5:] = 0
        y = torch.roll(torch.fft.irfft2(wf, s=(G, G)), shifts=(2, -1), dims=(1, 2))[..., None]
        return torch.cat([w, pos], dim=-1).contiguous().to(dev), y.contiguous().to(dev)

    first = tr.train_step(*batch()).item()
    for _ in range(150):
        last = tr.train_step(*batch()).item()
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    assert bool(torch.isfinite(tr.pflat).all())


def _nccl_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    g = gu.load_golden("train_c64_2l")
    kw = gu.golden_kwargs(g)
    B, M, N, seed, steps = [int(v) for v in g["meta"]]
    blk, tr = make_trainer(kw, seed + 7 * rank, dev)       # different start weights per rank: the rank-0 broadcast fixes it
    losses = []
    for s in range(steps):
        x_np, t_np = gu.make_block_io(kw, seed + 1 + s, B, M, N)
        sl = slice(rank * B // world, (rank + 1) * B // world)
        loss = tr.train_step(torch.from_numpy(x_np[sl].copy()).to(dev), torch.from_numpy(t_np[sl].copy()).to(dev))
        losses.append(loss.item())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pflat=tr.pflat.cpu().numpy(), losses=np.array(losses))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_two_ranks_rccl_equals_reference_full_batch(tmp_path):
    """The data-parallel step over RCCL on two MI355X: one process per GPU, one all-reduce of the flat gradient buffer per
    step.  Replicas stay bit-identical and end at the reference's single-process full-batch weights.  Needs two GPUs:
    skips itself on a one-GPU box (the world-size-2 gloo test covers the same logic on the CPU emulator)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_nccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["pflat"], r1["pflat"])
    g = gu.load_golden("train_c64_2l")
    np.testing.assert_allclose((r0["losses"] + r1["losses"]) / 2, g["losses"], rtol=2e-5)
    kw = gu.golden_kwargs(g)
    blk, _ = make_trainer(kw, int(g["meta"][3]), "cuda:0")
    off = 0
    for n in [n for n, _ in blk.engine_parameters()]:
        shape = blk.engine().param_shapes[n]
        cnt = int(np.prod(shape))
        assert gu.compare_packed(g, "final." + n, r0["pflat"][off:off + cnt].reshape(shape), 1e-5) < 2e-4, n
        off += cnt


def _normalizer_sync_worker(rank, world, port, out_dir):
    """Each rank accumulates its own shard; statistics are synced; the module is checkpointed, re-loaded (resume) and
    accumulates again.  Writes what every rank holds at the end."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fourierflow_amd.modules.normalizer import Normalizer

    def accumulate(nz, x):      # what ffno_markov_features does to the state (normalizer.py:45-55)
        nz.sum += x.sum(0)
        nz.sum_squared += (x * x).sum(0)
        nz.count += x.shape[0]
        nz.n_accumulations += 1

    g = torch.Generator().manual_seed(100 + rank)
    a, b = torch.randn(7, 3, generator=g) + rank, torch.randn(5, 3, generator=g) * 2
    nz = Normalizer([3])
    accumulate(nz, a)
    nz.sync_across_ranks()
    nz.sync_across_ranks()                       # idempotent: nothing new since the last sync
    ckpt = {k: v.clone() for k, v in nz.state_dict().items()}
    resumed = Normalizer([3])
    resumed.load_state_dict(ckpt)                # every rank loads the GLOBAL statistics
    resumed = resumed.to("cpu")                  # (a device move keeps the sync baseline with the buffers)
    loaded_count = float(resumed.count)
    accumulate(resumed, b)
    resumed.sync_across_ranks()
    torch.save(dict(count=resumed.count, sum=resumed.sum, sum_squared=resumed.sum_squared, loaded_count=loaded_count,
                    a=a, b=b), os.path.join(out_dir, f"nz{rank}.pt"))
    dist.destroy_process_group()


def test_normalizer_sync_survives_checkpoint_resume_under_ddp(tmp_path):
    """after resuming a checkpoint under DDP every rank already holds the global statistics -- the first sync must
    all-reduce only what was accumulated SINCE the load, not the loaded history once per rank.  World size 2, gloo."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_normalizer_sync_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"nz{k}.pt") for k in range(2)]
    every = torch.cat([r[0]["a"], r[1]["a"], r[0]["b"], r[1]["b"]]).double()
    for k in range(2):
        assert r[k]["loaded_count"] == 14.0                                   # 7 + 7: the first sync
        assert float(r[k]["count"]) == 24.0                                   # + 5 + 5, NOT 2 x 14 + 10
        np.testing.assert_allclose(r[k]["sum"].double().numpy(), every.sum(0).numpy(), rtol=1e-5)
        np.testing.assert_allclose(r[k]["sum_squared"].double().numpy(), (every * every).sum(0).numpy(), rtol=1e-5)
