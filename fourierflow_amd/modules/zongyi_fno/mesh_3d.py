"""FNOMesh3D -- MI355X-native mirror of ``fourierflow.modules.zongyi_fno.mesh_3d`` (reference mesh_3d.py:7-113), the
geo-FNO baseline of the plasticity experiments (experiments/plasticity/geo-fno*): ``x [B, X, Y, Z, 1] -> [B, X, Y, Z, 4]``.

Same constructor, parameter names / shapes / dtypes (``convs.{i}.weights1..4`` complex64 ``[in, out, m1, m2, m3]``) and
registration order (per layer: conv then w, mesh_3d.py:77-82 -- but ``convs`` and ``ws`` are separate ModuleLists, so the
state_dict lists all convs first).  Underneath: :class:`fourierflow_amd.engine_geofno.GeoFNO3DEngine`.  HIP only.
"""
import torch
import torch.nn as nn

from ... import _lib
from ...engine_geofno import GeoFNO3DEngine
from .mesh_2d import _ComplexCornerWeights, _MeshFn


class SpectralConv3d(_ComplexCornerWeights):
    """mesh_3d.py:8-31: four corner-block weight tensors [in, out, modes1, modes2, modes3]."""

    def __init__(self, in_channels, out_channels, modes1, modes2, modes3):
        super().__init__(in_channels, out_channels, (modes1, modes2, modes3), 4)
        self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3


class FNOMesh3D(nn.Module):
    def __init__(self, modes1, modes2, modes3, width, n_layers=4):
        super().__init__()
        self.modes1, self.modes2, self.modes3, self.width, self.n_layers = modes1, modes2, modes3, width, n_layers
        self.padding = 5
        self.input_dim, self.output_dim = 4, 4            # value + three grid channels (mesh_3d.py:68,87-88); fc2: 128 -> 4
        self.fc0 = nn.Linear(4, width)
        self.convs = nn.ModuleList([SpectralConv3d(width, width, modes1, modes2, modes3) for _ in range(n_layers)])
        self.ws = nn.ModuleList([nn.Conv3d(width, width, 1) for _ in range(n_layers)])
        self.fc1 = nn.Linear(width, 128)
        self.fc2 = nn.Linear(128, 4)
        self._engine = None
        self._generation = 0

    # -- engine plumbing ----------------------------------------------------------------------------
    def engine(self) -> GeoFNO3DEngine:
        if self._engine is None:
            self._engine = GeoFNO3DEngine(modes1=self.modes1, modes2=self.modes2, modes3=self.modes3, width=self.width,
                                          n_layers=self.n_layers)
        return self._engine

    def engine_parameters(self):
        named = dict(self.named_parameters())
        return [(n, named[n]) for n in self.engine().param_names]

    def get_grid(self, shape, device):
        B, X, Y, Z = shape[0], shape[1], shape[2], shape[3]
        gx = torch.linspace(0, 1, X, device=device).reshape(1, X, 1, 1, 1).expand(B, X, Y, Z, 1)
        gy = torch.linspace(0, 1, Y, device=device).reshape(1, 1, Y, 1, 1).expand(B, X, Y, Z, 1)
        gz = torch.linspace(0, 1, Z, device=device).reshape(1, 1, 1, Z, 1).expand(B, X, Y, Z, 1)
        return torch.cat((gx, gy, gz), dim=-1)

    def prepare_input(self, x):
        return torch.cat((x, self.get_grid(x.shape, x.device)), dim=-1).contiguous()

    def _engine_for(self, params):
        eng = self.engine()
        eng.bind({n: p.detach() for n, p in zip(eng.param_names, params)})
        return eng

    def forward(self, x):
        _lib.require_device_tensor(x, "FNOMesh3D input")
        params = [p for _, p in self.engine_parameters()]
        return _MeshFn.apply(self.prepare_input(x), self, *params)
