"""F-FNO 2-D regular-grid operator -- MI355X-native mirror of
``fourierflow.modules.factorized_fno.grid_2d`` (reference grid_2d.py:10-177).

Same classes, constructor signatures, parameter names/shapes (state_dict compatible, including the
duplicated shared-weight keys) and forward contracts; everything underneath is the HIP kernel
sequence driven by :class:`fourierflow_amd.engine.FFNO2DEngine`.  HIP only: CPU tensors raise.
"""
import torch
import torch.nn as nn

from ... import _lib
from ...engine import FFNO2DEngine
from ..feedforward import FeedForward
from ..linear import WNLinear


def _fourier_weights(in_dim, out_dim, n_modes, gain=1.0):
    plist = nn.ParameterList([])
    for _ in range(2):
        param = nn.Parameter(torch.empty(in_dim, out_dim, n_modes, 2))
        nn.init.xavier_normal_(param, gain=gain)  # fan_in = O*K*2, fan_out = I*K*2 (grid_2d.py:26-28)
        plist.append(param)
    return plist


class SpectralConv2d(nn.Module):
    """One factorized spectral layer: ``forward(x) -> (backcast, forecast|None)`` (grid_2d.py:42-49)."""

    def __init__(self, in_dim, out_dim, n_modes, forecast_ff, backcast_ff, fourier_weight, factor, ff_weight_norm,
                 n_ff_layers, layer_norm, use_fork, dropout, mode):
        super().__init__()
        if in_dim != out_dim:
            raise NotImplementedError("in_dim != out_dim is not used by the F-FNO block")
        self.in_dim, self.out_dim, self.n_modes, self.mode, self.use_fork = in_dim, out_dim, n_modes, mode, use_fork
        self.fourier_weight = fourier_weight
        if not self.fourier_weight:
            self.fourier_weight = _fourier_weights(in_dim, out_dim, n_modes)
        if use_fork:
            self.forecast_ff = forecast_ff
            if not self.forecast_ff:
                self.forecast_ff = FeedForward(out_dim, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout, general_ok=True)
        self.backcast_ff = backcast_ff
        if not self.backcast_ff:
            self.backcast_ff = FeedForward(out_dim, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout, general_ok=True)

    def forward_fourier(self, x):
        from ...ops import spectral_conv2d
        return spectral_conv2d(x, self.fourier_weight[0], self.fourier_weight[1], self.n_modes, self.mode)

    def forward(self, x):
        if self.mode != 'no-fourier':
            x = self.forward_fourier(x)
        b = self.backcast_ff(x)
        f = self.forecast_ff(x) if self.use_fork else None
        return b, f


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module._engine_for(params)
        need_grad = ctx.needs_input_grad[0] or any(ctx.needs_input_grad[2:])
        y = eng.forward(x, need_grad, training=module.training)      # (dropout follows nn.Module.training like the reference)
        module._generation += 1
        ctx.module, ctx.gen, ctx.n = module, module._generation, len(params)
        return y

    @staticmethod
    def backward(ctx, gy):
        module = ctx.module
        if module._generation != ctx.gen:
            raise RuntimeError("FNOFactorized2DBlock: only the most recent forward pass can be back-propagated "
                               "(activations live in one pre-allocated workspace)")
        eng = module._engine
        # the reference modules are ordinary autograd modules: dL/dx goes to whatever produced the input (grid_2d.py:154-177)
        flat = eng.backward(gy.contiguous(), need_dx=ctx.needs_input_grad[0]).clone()
        grads = []
        off = 0
        for n in eng.param_names:
            cnt = 1
            for s in eng.param_shapes[n]:
                cnt *= s
            grads.append(flat[off:off + cnt].view(eng.param_shapes[n]))
            off += cnt
        return (eng.dx if ctx.needs_input_grad[0] else None, None, *grads)


class FNOFactorized2DBlock(nn.Module):
    def __init__(self, modes, width, input_dim=12, dropout=0.0, in_dropout=0.0, n_layers=4,
                 share_weight: bool = False, share_fork=False, factor=2, ff_weight_norm=False, n_ff_layers=2,
                 gain=1, layer_norm=False, use_fork=False, mode='full'):
        super().__init__()
        self.modes, self.width, self.input_dim = modes, width, input_dim
        self.n_layers, self.use_fork, self.mode = n_layers, use_fork, mode
        self.share_weight, self.share_fork = share_weight, share_fork
        self.factor, self.ff_weight_norm, self.layer_norm = factor, ff_weight_norm, bool(layer_norm)
        self.n_ff_layers, self.dropout, self.in_dropout = n_ff_layers, float(dropout), float(in_dropout)
        self.in_proj = WNLinear(input_dim, width, wnorm=ff_weight_norm)
        self.drop = nn.Dropout(in_dropout)      # (no parameters; the mask is drawn inside the engine: ffno_dropout)

        self.forecast_ff = self.backcast_ff = None
        if share_fork:
            if use_fork:
                self.forecast_ff = FeedForward(width, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout, general_ok=True)
            self.backcast_ff = FeedForward(width, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout, general_ok=True)

        self.fourier_weight = None
        if share_weight:
            self.fourier_weight = _fourier_weights(width, width, modes, gain)

        self.spectral_layers = nn.ModuleList([])
        for _ in range(n_layers):
            self.spectral_layers.append(SpectralConv2d(
                in_dim=width, out_dim=width, n_modes=modes, forecast_ff=self.forecast_ff,
                backcast_ff=self.backcast_ff, fourier_weight=self.fourier_weight, factor=factor,
                ff_weight_norm=ff_weight_norm, n_ff_layers=n_ff_layers, layer_norm=layer_norm, use_fork=use_fork,
                dropout=dropout, mode=mode))

        self.out = nn.Sequential(WNLinear(width, 128, wnorm=ff_weight_norm), WNLinear(128, 1, wnorm=ff_weight_norm))
        self._engine = None
        self._generation = 0

    # -- engine plumbing ----------------------------------------------------------------------------
    def engine(self) -> FFNO2DEngine:
        if self._engine is None:
            self._engine = FFNO2DEngine(modes=self.modes, width=self.width, input_dim=self.input_dim,
                                        n_layers=self.n_layers, factor=self.factor, share_weight=self.share_weight,
                                        share_fork=self.share_fork, ff_weight_norm=self.ff_weight_norm, mode=self.mode,
                                        use_fork=self.use_fork, layer_norm=self.layer_norm, n_ff_layers=self.n_ff_layers,
                                        dropout=self.dropout, in_dropout=self.in_dropout)
        return self._engine

    def engine_parameters(self):
        """Unique parameters in the engine's flat-buffer order (reference named_parameters() names)."""
        eng = self.engine()
        slots = self.__dict__.get("_param_slots")
        if slots is None:
            # where each engine parameter lives: (owning submodule, attribute).  The module tree is fixed after construction, so
            # the walk over named_parameters() (0.3 ms of a 0.7 ms batch-1 forward) is done once; the Parameter objects are
            # re-read from their owners on every call (a replaced or re-registered Parameter is seen).
            owners = {}
            for prefix, mod in self.named_modules():
                for attr in mod._parameters:
                    owners.setdefault(prefix + ("." if prefix else "") + attr, (mod, attr))
            slots = [(n,) + owners[n] for n in eng.param_names]
            self.__dict__["_param_slots"] = slots
        return [(n, mod._parameters[attr]) for n, mod, attr in slots]

    def prepare_input(self, x):
        return x

    def _engine_for(self, params):
        eng = self.engine()
        eng.bind({n: p.detach() for n, p in zip(eng.param_names, params)})
        return eng

    def forward(self, x, **kwargs):
        # x.shape == [n_batches, *dim_sizes, input_size]; kwargs (global_step) are ignored like the reference
        _lib.require_device_tensor(x, "FNOFactorized2DBlock input")
        params = [p for _, p in self.engine_parameters()]
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            forecast = _BlockFn.apply(x, self, *params)
        else:      # inference: no autograd node (a Function.apply over ~100 tensors costs as much as a third of the kernels)
            forecast = self._engine_for(params).forward(x, False, training=self.training)
            self._generation += 1
        # forecast_list (per-layer head outputs, use_fork) is returned for logging like the reference does; gradients
        # flow through 'forecast' only
        flist = list(getattr(self._engine, "forecast_list", [])) if self.use_fork else []
        return {'forecast': forecast, 'forecast_list': flist}
