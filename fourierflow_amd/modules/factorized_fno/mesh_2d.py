"""F-FNO 2-D structured-mesh operator -- MI355X-native mirror of
``fourierflow.modules.factorized_fno.mesh_2d`` (reference mesh_2d.py:14-175; the airfoil / pipe / elasticity experiments).

Same classes, constructor signatures, parameter names/shapes and forward contract
(``forward(x[B,X,Y,input_dim-2]) -> [B,X,Y,1]``).  Differences from the torus block that matter here: ``modes_x`` and
``modes_y`` are independent, ``fourier_weight[0]`` mixes the FIRST spatial axis (mesh_2d.py:92-96) and ``[1]`` the last
(mesh_2d.py:71-75), the lifted features are zero-padded by 8 on both axes and the last backcast is cropped before the
head (mesh_2d.py:150,158).  Pad and crop are index maps inside the lift / head kernels.  HIP only: CPU tensors raise.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ...engine import FFNOEngine
from ..feedforward import FeedForward
from ..linear import WNLinear


class SpectralConv2d(nn.Module):
    """Parameter container of one layer (mesh_2d.py:14-46); the arithmetic runs in the engine."""

    def __init__(self, in_dim, out_dim, modes_x, modes_y, forecast_ff, backcast_ff, fourier_weight, factor,
                 ff_weight_norm, n_ff_layers, layer_norm, use_fork, dropout, mode):
        super().__init__()
        if use_fork:
            raise NotImplementedError("use_fork is not used by FNOFactorizedMesh2D")
        if mode != "full":
            raise NotImplementedError("FNOFactorizedMesh2D always runs mode='full' (mesh_2d.py:140)")
        self.in_dim, self.out_dim, self.modes_x, self.modes_y = in_dim, out_dim, modes_x, modes_y
        self.mode, self.use_fork = mode, use_fork
        self.fourier_weight = fourier_weight
        if not self.fourier_weight:
            self.fourier_weight = nn.ParameterList([])
            for n_modes in [modes_x, modes_y]:
                param = nn.Parameter(torch.empty(in_dim, out_dim, n_modes, 2))
                nn.init.xavier_normal_(param)
                self.fourier_weight.append(param)
        self.backcast_ff = backcast_ff
        if not self.backcast_ff:
            self.backcast_ff = FeedForward(out_dim, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout, general_ok=True)


class _Mesh2DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module._engine_for(params)
        y = eng.forward(x, ctx.needs_input_grad[0] or any(ctx.needs_input_grad[2:]))
        module._generation += 1
        ctx.module, ctx.gen = module, module._generation
        return y

    @staticmethod
    def backward(ctx, gy):
        module = ctx.module
        if module._generation != ctx.gen:
            raise RuntimeError("FNOFactorizedMesh2D: only the most recent forward pass can be back-propagated")
        eng = module._engine
        # the reference modules are ordinary autograd modules: dL/dx goes to whatever produced the input (grid_2d.py:154-177)
        flat = eng.backward(gy.contiguous(), need_dx=ctx.needs_input_grad[0]).clone()
        grads, off = [], 0
        for n in eng.param_names:
            cnt = int(np.prod(eng.param_shapes[n]))
            grads.append(flat[off:off + cnt].view(eng.param_shapes[n]))
            off += cnt
        return (eng.dx if ctx.needs_input_grad[0] else None, None, *grads)


class FNOFactorizedMesh2D(nn.Module):
    def __init__(self, modes_x, modes_y, width, input_dim, n_layers, share_weight, factor, ff_weight_norm, n_ff_layers,
                 layer_norm):
        super().__init__()
        self.padding = 8  # pad the domain if input is non-periodic (mesh_2d.py:113)
        self.modes_x, self.modes_y = modes_x, modes_y
        self.width, self.input_dim, self.n_layers = width, input_dim, n_layers
        self.share_weight, self.factor, self.ff_weight_norm = share_weight, factor, ff_weight_norm
        self.in_proj = WNLinear(input_dim, self.width, wnorm=ff_weight_norm)
        self.fourier_weight = None
        if share_weight:
            self.fourier_weight = nn.ParameterList([])
            for n_modes in [modes_x, modes_y]:
                param = nn.Parameter(torch.empty(width, width, n_modes, 2))
                nn.init.xavier_normal_(param)
                self.fourier_weight.append(param)
        self.spectral_layers = nn.ModuleList([])
        for _ in range(n_layers):
            self.spectral_layers.append(SpectralConv2d(
                in_dim=width, out_dim=width, modes_x=modes_x, modes_y=modes_y, forecast_ff=None, backcast_ff=None,
                fourier_weight=self.fourier_weight, factor=factor, ff_weight_norm=ff_weight_norm, n_ff_layers=n_ff_layers,
                layer_norm=layer_norm, use_fork=False, dropout=0.0, mode='full'))
        self.out = nn.Sequential(WNLinear(self.width, 128, wnorm=ff_weight_norm), WNLinear(128, 1, wnorm=ff_weight_norm))
        self.layer_norm = bool(layer_norm)
        self.n_ff_layers = n_ff_layers
        self._engine = None
        self._generation = 0

    def engine(self) -> FFNOEngine:
        if self._engine is None:
            self._engine = FFNOEngine(modes=(self.modes_x, self.modes_y), width=self.width, input_dim=self.input_dim,
                                      n_layers=self.n_layers, factor=self.factor, share_weight=self.share_weight,
                                      share_fork=False, ff_weight_norm=self.ff_weight_norm, mode="full", spatial_dims=2,
                                      padding=self.padding, output_dim=1, first_axis_first=True, layer_norm=self.layer_norm, n_ff_layers=self.n_ff_layers)
        return self._engine

    def engine_parameters(self):
        named = dict(self.named_parameters())
        return [(n, named[n]) for n in self.engine().param_names]

    def _engine_for(self, params):
        eng = self.engine()
        eng.bind({n: p.detach() for n, p in zip(eng.param_names, params)})
        return eng

    def get_grid(self, shape, device):
        B, X, Y = shape[0], shape[1], shape[2]
        gx = torch.tensor(np.linspace(0, 1, X), dtype=torch.float).reshape(1, X, 1, 1).repeat([B, 1, Y, 1])
        gy = torch.tensor(np.linspace(0, 1, Y), dtype=torch.float).reshape(1, 1, Y, 1).repeat([B, X, 1, 1])
        return torch.cat((gx, gy), dim=-1).to(device)

    def prepare_input(self, x):
        """Append the linspace coordinate channels (mesh_2d.py:147-148): [B, X, Y, input_dim - 2] -> [..., input_dim]."""
        key = (tuple(x.shape[:3]), x.device)
        if getattr(self, "_grid_key", None) != key:     # the grid only depends on the shape: build it once
            self._grid, self._grid_key = self.get_grid(x.shape, x.device), key
        return torch.cat((x, self._grid), dim=-1)

    def forward(self, x):
        _lib.require_device_tensor(x, "FNOFactorizedMesh2D input")
        x = self.prepare_input(x)   # [B, X, Y, input_dim]
        params = [p for _, p in self.engine_parameters()]
        return _Mesh2DFn.apply(x, self, *params)
