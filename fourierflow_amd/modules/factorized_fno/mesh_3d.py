"""F-FNO 3-D structured-mesh operator -- MI355X-native mirror of
``fourierflow.modules.factorized_fno.mesh_3d`` (reference mesh_3d.py:14-189, the plasticity experiments).

Same classes, constructor signatures, parameter names/shapes and forward contract
(``forward(x[B,X,Y,Z,input_dim-3]) -> [B,X,Y,Z,output_dim]``).  The three per-axis branches run on the same
HIP spectral kernels as the 2-D block through reshaped views; the +8 zero padding and the final crop are
index maps inside the lift / head kernels (no pad or crop copies).  HIP only: CPU tensors raise.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ...engine import FFNOEngine
from ..feedforward import FeedForward
from ..linear import WNLinear


class SpectralConv2d(nn.Module):
    """3-axis factorized spectral layer (the reference keeps the 2d name, mesh_3d.py:14)."""

    def __init__(self, in_dim, out_dim, modes_x, modes_y, modes_z, forecast_ff, backcast_ff, fourier_weight, factor,
                 ff_weight_norm, n_ff_layers, layer_norm, use_fork, dropout):
        super().__init__()
        if use_fork:
            raise NotImplementedError("use_fork is not used by FNOFactorizedMesh3D")
        self.in_dim, self.out_dim = in_dim, out_dim
        self.modes_x, self.modes_y, self.modes_z, self.use_fork = modes_x, modes_y, modes_z, use_fork
        self.fourier_weight = fourier_weight
        if not self.fourier_weight:
            self.fourier_weight = nn.ParameterList([])
            for n_modes in [modes_x, modes_y, modes_z]:
                param = nn.Parameter(torch.empty(in_dim, out_dim, n_modes, 2))
                nn.init.xavier_normal_(param)
                self.fourier_weight.append(param)
        self.backcast_ff = backcast_ff
        if not self.backcast_ff:
            self.backcast_ff = FeedForward(out_dim, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout, general_ok=True)


class _Mesh3DFn(torch.autograd.Function):
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
            raise RuntimeError("FNOFactorizedMesh3D: only the most recent forward pass can be back-propagated")
        eng = module._engine
        # the reference modules are ordinary autograd modules: dL/dx goes to whatever produced the input (grid_2d.py:154-177)
        flat = eng.backward(gy.contiguous(), need_dx=ctx.needs_input_grad[0]).clone()
        grads, off = [], 0
        for n in eng.param_names:
            cnt = int(np.prod(eng.param_shapes[n]))
            grads.append(flat[off:off + cnt].view(eng.param_shapes[n]))
            off += cnt
        return (eng.dx if ctx.needs_input_grad[0] else None, None, *grads)


class FNOFactorizedMesh3D(nn.Module):
    def __init__(self, modes_x, modes_y, modes_z, width, input_dim, output_dim, n_layers, share_weight, factor,
                 ff_weight_norm, n_ff_layers, layer_norm):
        super().__init__()
        self.padding = 8  # pad the domain if input is non-periodic (mesh_3d.py:120)
        self.modes_x, self.modes_y, self.modes_z = modes_x, modes_y, modes_z
        self.width, self.input_dim, self.output_dim, self.n_layers = width, input_dim, output_dim, n_layers
        self.share_weight, self.factor, self.ff_weight_norm = share_weight, factor, ff_weight_norm
        self.in_proj = WNLinear(input_dim, self.width, wnorm=ff_weight_norm)
        self.fourier_weight = None
        if share_weight:
            self.fourier_weight = nn.ParameterList([])
            for n_modes in [modes_x, modes_y, modes_z]:
                param = nn.Parameter(torch.empty(width, width, n_modes, 2))
                nn.init.xavier_normal_(param)
                self.fourier_weight.append(param)
        self.spectral_layers = nn.ModuleList([])
        for _ in range(n_layers):
            self.spectral_layers.append(SpectralConv2d(
                in_dim=width, out_dim=width, modes_x=modes_x, modes_y=modes_y, modes_z=modes_z, forecast_ff=None,
                backcast_ff=None, fourier_weight=self.fourier_weight, factor=factor, ff_weight_norm=ff_weight_norm,
                n_ff_layers=n_ff_layers, layer_norm=layer_norm, use_fork=False, dropout=0.0))
        self.out = nn.Sequential(WNLinear(self.width, 128, wnorm=ff_weight_norm),
                                 WNLinear(128, output_dim, wnorm=ff_weight_norm))
        self.layer_norm = bool(layer_norm)
        self.n_ff_layers = n_ff_layers
        self._engine = None
        self._generation = 0

    def engine(self) -> FFNOEngine:
        if self._engine is None:
            self._engine = FFNOEngine(modes=(self.modes_x, self.modes_y, self.modes_z), width=self.width,
                                      input_dim=self.input_dim, n_layers=self.n_layers, factor=self.factor,
                                      share_weight=self.share_weight, share_fork=False,
                                      ff_weight_norm=self.ff_weight_norm, mode="full", spatial_dims=3,
                                      padding=self.padding, output_dim=self.output_dim, layer_norm=self.layer_norm, n_ff_layers=self.n_ff_layers)
        return self._engine

    def engine_parameters(self):
        named = dict(self.named_parameters())
        return [(n, named[n]) for n in self.engine().param_names]

    def _engine_for(self, params):
        eng = self.engine()
        eng.bind({n: p.detach() for n, p in zip(eng.param_names, params)})
        return eng

    def get_grid(self, shape, device):
        B, X, Y, Z = shape[0], shape[1], shape[2], shape[3]
        gx = torch.tensor(np.linspace(0, 1, X), dtype=torch.float).reshape(1, X, 1, 1, 1).repeat([B, 1, Y, Z, 1])
        gy = torch.tensor(np.linspace(0, 1, Y), dtype=torch.float).reshape(1, 1, Y, 1, 1).repeat([B, X, 1, Z, 1])
        gz = torch.tensor(np.linspace(0, 1, Z), dtype=torch.float).reshape(1, 1, 1, Z, 1).repeat([B, X, Y, 1, 1])
        return torch.cat((gx, gy, gz), dim=-1).to(device)

    def prepare_input(self, x):
        """Append the linspace coordinate channels (mesh_3d.py:161-162): [B, X, Y, Z, input_dim - 3] -> [..., input_dim]."""
        key = (tuple(x.shape[:4]), x.device)
        if getattr(self, "_grid_key", None) != key:     # the grid only depends on the shape: build it once
            self._grid, self._grid_key = self.get_grid(x.shape, x.device), key
        return torch.cat((x, self._grid), dim=-1)

    def forward(self, x):
        _lib.require_device_tensor(x, "FNOFactorizedMesh3D input")
        x = self.prepare_input(x)   # [B, X, Y, Z, input_dim]
        params = [p for _, p in self.engine_parameters()]
        return _Mesh3DFn.apply(x, self, *params)
