"""FNOZongyi2DBlock -- MI355X-native mirror of ``fourierflow.modules.zongyi_fno.grid_2d`` (reference grid_2d.py:16-129),
the original FNO baseline of BASELINE config 0 (experiments/torus_li/zongyi/4_layers: modes 12, width 20, 4 layers).

Same constructor signature, parameter names / shapes / registration order (state_dict compatible) and forward contract
(``{'forecast': [B, M, N, 1]}``); underneath is the HIP kernel sequence of
:class:`fourierflow_amd.engine_zongyi.ZongyiEngine`.  HIP only: CPU tensors raise.  The autograd node also returns the
INPUT gradient and several forward passes may be alive at once, which is what the 10-step rollout training of
Grid2DRolloutExperiment needs.
"""
import torch
import torch.nn as nn

from ... import _lib
from ...engine_zongyi import ZongyiEngine


class SpectralConv2d(nn.Module):
    """Parameter container of one layer (grid_2d.py:17-30): ``linear`` then the two corner-block weights."""

    def __init__(self, in_dim, out_dim, n_modes, resdiual=True, dropout=0.1):
        super().__init__()
        self.in_dim, self.out_dim, self.n_modes, self.residual = in_dim, out_dim, n_modes, resdiual
        self.linear = nn.Linear(in_dim, out_dim)
        self.fourier_weight = nn.ParameterList(
            [nn.Parameter(torch.empty(in_dim, out_dim, n_modes, n_modes, 2)) for _ in range(2)])
        for param in self.fourier_weight:
            nn.init.xavier_normal_(param, gain=1 / (in_dim * out_dim))      # grid_2d.py:29-30

    def forward(self, x):
        raise RuntimeError("layers of FNOZongyi2DBlock run inside the block's fused HIP pass; call the block")


class _ZongyiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module._engine_for(params)
        need_grad = any(ctx.needs_input_grad)
        slot = 0
        if need_grad:
            slot = module._next_slot()
        y = eng.forward(x, need_grad, slot=slot, n_slots=module.max_live_passes, weights_ready=module.weights_frozen)
        ctx.module, ctx.slot, ctx.ticket = module, slot, module._tickets[slot] if need_grad else None
        return y

    @staticmethod
    def backward(ctx, gy):
        module = ctx.module
        if module._tickets[ctx.slot] != ctx.ticket:
            raise RuntimeError(f"FNOZongyi2DBlock: more than max_live_passes={module.max_live_passes} forward passes were "
                               "alive at once; raise module.max_live_passes")
        eng = module._engine
        need_dx = ctx.needs_input_grad[0]
        if module.fused_grad_accumulation:      # the routine reads eng.gflat after the whole graph ran (eng.zero_grad() first)
            res = eng.backward(gy.contiguous(), slot=ctx.slot, need_dx=need_dx, accumulate=True)
            return (res[1] if need_dx else None, None, *([None] * len(eng.param_names)))
        flat, dx = eng.backward(gy.contiguous(), slot=ctx.slot, need_dx=True)
        flat = flat.clone()
        grads, off = [], 0
        for n in eng.param_names:
            cnt = 1
            for s in eng.param_shapes[n]:
                cnt *= s
            grads.append(flat[off:off + cnt].view(eng.param_shapes[n]))
            off += cnt
        return (dx if ctx.needs_input_grad[0] else None, None, *grads)


class FNOZongyi2DBlock(nn.Module):
    def __init__(self, modes1, modes2, width, input_dim=12, dropout=0.1, n_layers=4, residual=False, conv_residual=True):
        super().__init__()
        if modes1 != modes2:
            raise NotImplementedError("modes1 != modes2: the reference itself only uses modes1 (grid_2d.py:111)")
        self.modes1, self.modes2, self.width, self.input_dim = modes1, modes2, width, input_dim
        self.n_layers, self.residual, self.conv_residual = n_layers, residual, conv_residual
        self.in_proj = nn.Linear(input_dim, width)
        self.spectral_layers = nn.ModuleList([
            SpectralConv2d(in_dim=width, out_dim=width, n_modes=modes1, resdiual=conv_residual, dropout=dropout)
            for _ in range(n_layers)])
        self.feedforward = nn.Sequential(nn.Linear(width, 128), nn.ReLU(inplace=True), nn.Linear(128, 1))
        self._engine = None
        self.weights_frozen = False              # True: the padded / packed weights of the last pass are still valid
        self.fused_grad_accumulation = False     # True: parameter gradients accumulate in engine().gflat, .grad stays None
        self.max_live_passes = 16       # forward passes that may await their backward at once (rollout: n_steps)
        self._tickets, self._cursor, self._ticket = {}, 0, 0

    # -- engine plumbing ----------------------------------------------------------------------------
    def engine(self) -> ZongyiEngine:
        if self._engine is None:
            self._engine = ZongyiEngine(modes=self.modes1, width=self.width, input_dim=self.input_dim, n_layers=self.n_layers,
                                        residual=self.residual, conv_residual=self.conv_residual)
        return self._engine

    def engine_parameters(self):
        named = dict(self.named_parameters())
        return [(n, named[n]) for n in self.engine().param_names]

    def prepare_input(self, x):
        return x

    def _engine_for(self, params):
        eng = self.engine()
        eng.bind({n: p.detach() for n, p in zip(eng.param_names, params)})
        return eng

    def _next_slot(self):
        slot = self._cursor % self.max_live_passes
        self._cursor += 1
        self._ticket += 1
        self._tickets[slot] = self._ticket
        return slot

    def forward(self, x, **kwargs):
        # x.shape == [n_batches, *dim_sizes, input_size]
        _lib.require_device_tensor(x, "FNOZongyi2DBlock input")
        params = [p for _, p in self.engine_parameters()]
        return {'forecast': _ZongyiFn.apply(x, self, *params)}
