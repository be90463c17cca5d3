"""FNOMesh2D -- MI355X-native mirror of ``fourierflow.modules.zongyi_fno.mesh_2d`` (reference mesh_2d.py:14-106), the
geo-FNO baseline of the airfoil / pipe experiments (experiments/{airfoil,pipe}/geo-fno*).

Same constructor, parameter names / shapes / dtypes and registration order -- ``convs.{i}.weights1/2`` are complex64
``[in, out, modes1, modes2]`` in the state_dict exactly like the reference's (they are STORED as their ``view_as_real``
twins so that one flat fp32 buffer can hold every parameter for the fused optimiser) -- and the same forward contract
(``x [B, X, Y, 2] -> [B, X, Y, 1]``).  Underneath: :class:`fourierflow_amd.engine_geofno.GeoFNO2DEngine`.  HIP only.
"""
import torch
import torch.nn as nn

from ... import _lib
from ...engine_geofno import GeoFNO2DEngine


class _ComplexCornerWeights(nn.Module):
    """Parameter container of one Fourier layer: ``weights1..n``, complex64 ``[in, out, *modes]`` in the state_dict, stored as
    their ``view_as_real`` twins."""

    def __init__(self, in_channels, out_channels, modes, n_weights):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.scale = 1 / (in_channels * out_channels)
        self._names = tuple(f"weights{j}" for j in range(1, n_weights + 1))
        for name in self._names:       # scale * torch.rand(..., dtype=cfloat): real and imaginary parts U[0, 1)
            setattr(self, name, nn.Parameter(self.scale * torch.rand(in_channels, out_channels, *modes, 2)))
        self._register_state_dict_hook(self._complex_out)
        self._register_load_state_dict_pre_hook(self._complex_in)

    @staticmethod
    def _complex_out(module, state_dict, prefix, local_metadata):
        for name in module._names:
            state_dict[prefix + name] = torch.view_as_complex(state_dict[prefix + name].contiguous())

    def _complex_in(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name in self._names:
            t = state_dict.get(prefix + name)
            if t is not None and t.is_complex():
                state_dict[prefix + name] = torch.view_as_real(t.to(torch.complex64))

    def forward(self, x):
        raise RuntimeError("Fourier layers of FNOMesh2D / FNOMesh3D run inside the model's fused HIP pass; call the model")


class SpectralConv2d(_ComplexCornerWeights):
    """mesh_2d.py:15-32: two corner-block weight tensors [in, out, modes1, modes2]."""

    def __init__(self, in_channels, out_channels, modes1, modes2):
        super().__init__(in_channels, out_channels, (modes1, modes2), 2)
        self.modes1, self.modes2 = modes1, modes2


class _MeshFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module._engine_for(params)
        need_grad = any(ctx.needs_input_grad[2:])
        y = eng.forward(x, need_grad)
        module._generation += 1
        ctx.module, ctx.gen = module, module._generation
        return y

    @staticmethod
    def backward(ctx, gy):
        module = ctx.module
        if ctx.needs_input_grad[0]:
            # the reference modules are ordinary autograd modules and would propagate dL/dx; this operator's kernels stop at
            # the lift's parameter gradients -- fail loudly instead of silently training upstream parameters without that term
            raise RuntimeError(f"{type(module).__name__}: the gradient with respect to the INPUT tensor is not computed by the "
                               "gfx950 kernel set (feed it a tensor that does not require grad)")
        if module._generation != ctx.gen:
            raise RuntimeError("FNOMesh2D / FNOMesh3D: only the most recent forward pass can be back-propagated")
        eng = module._engine
        flat = eng.backward(gy.contiguous()).clone()
        grads, off = [], 0
        for n in eng.param_names:
            cnt = 1
            for s in eng.param_shapes[n]:
                cnt *= s
            grads.append(flat[off:off + cnt].view(eng.param_shapes[n]))
            off += cnt
        return (None, None, *grads)


class FNOMesh2D(nn.Module):
    def __init__(self, modes1, modes2, width, n_layers):
        super().__init__()
        self.modes1, self.modes2, self.width, self.n_layers = modes1, modes2, width, n_layers
        self.padding = 8
        self.input_dim = 4                               # (mesh x, mesh y) + the two grid channels (mesh_2d.py:62,79-80)
        self.fc0 = nn.Linear(4, width)
        self.convs = nn.ModuleList([SpectralConv2d(width, width, modes1, modes2) for _ in range(n_layers)])
        self.ws = nn.ModuleList([nn.Conv2d(width, width, 1) for _ in range(n_layers)])
        self.fc1 = nn.Linear(width, 128)
        self.fc2 = nn.Linear(128, 1)
        self._engine = None
        self._generation = 0

    # -- engine plumbing ----------------------------------------------------------------------------
    def engine(self) -> GeoFNO2DEngine:
        if self._engine is None:
            self._engine = GeoFNO2DEngine(modes1=self.modes1, modes2=self.modes2, width=self.width, n_layers=self.n_layers)
        return self._engine

    def engine_parameters(self):
        named = dict(self.named_parameters())
        return [(n, named[n]) for n in self.engine().param_names]

    def get_grid(self, shape, device):
        B, X, Y = shape[0], shape[1], shape[2]
        gx = torch.linspace(0, 1, X, device=device).reshape(1, X, 1, 1).expand(B, X, Y, 1)
        gy = torch.linspace(0, 1, Y, device=device).reshape(1, 1, Y, 1).expand(B, X, Y, 1)
        return torch.cat((gx, gy), dim=-1)

    def prepare_input(self, x):
        return torch.cat((x, self.get_grid(x.shape, x.device)), dim=-1).contiguous()

    def _engine_for(self, params):
        eng = self.engine()
        eng.bind({n: p.detach() for n, p in zip(eng.param_names, params)})
        return eng

    def forward(self, x):
        _lib.require_device_tensor(x, "FNOMesh2D input")
        params = [p for _, p in self.engine_parameters()]
        return _MeshFn.apply(self.prepare_input(x), self, *params)
