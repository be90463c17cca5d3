"""CNOFactorized2DBlock / CNOFactorizedMesh2D / CNOFactorizedMesh3D -- MI355X-native mirrors of
``fourierflow.modules.factorized_cno`` (reference factorized_cno/grid_2d.py:98-172, mesh_2d.py:103-170, mesh_3d.py:120-194):
the F-FNO operators with an orthonormal DCT-II along every axis instead of the rFFT (modules/dct.py) and REAL per-mode
weights ``[in, out, modes]``.  Constructor signatures, parameter names / shapes, state_dict layout and forward contracts are
the reference's; everything else (feed-forward, weight norm, padding, grid channels, heads) is inherited from the F-FNO
mirrors, and the engine runs the DCT branch (``ffno_dct_branch``: the truncated real-DFT kernels over the L samples of a
length-2L transform plus a phase rotation).  HIP only.
"""
import torch
import torch.nn as nn

from ...engine import FFNOEngine
from ..factorized_fno.grid_2d import FNOFactorized2DBlock
from ..factorized_fno.mesh_2d import FNOFactorizedMesh2D
from ..factorized_fno.mesh_3d import FNOFactorizedMesh3D


def _make_real(module: nn.Module, shared_gain: float = 1.0):
    """Replace every Fourier weight [in, out, modes, 2] by a real [in, out, modes] one, in place in its ParameterList
    (shared lists stay shared), initialised like the reference (xavier_normal_; the shared list with ``gain``)."""
    seen = set()
    lists = [(module.fourier_weight, shared_gain)] if getattr(module, "fourier_weight", None) else []
    lists += [(layer.fourier_weight, 1.0) for layer in module.spectral_layers]
    for plist, gain in lists:
        if id(plist) in seen:
            continue
        seen.add(id(plist))
        for i, p in enumerate(plist):
            w = nn.Parameter(torch.empty(*p.shape[:-1]))
            nn.init.xavier_normal_(w, gain=gain)
            plist[i] = w


class CNOFactorized2DBlock(FNOFactorized2DBlock):
    def __init__(self, modes, width, input_dim=12, dropout=0.0, in_dropout=0.0, n_layers=4, share_weight: bool = False,
                 share_fork=False, factor=2, ff_weight_norm=False, n_ff_layers=2, gain=1, layer_norm=False, use_fork=False,
                 mode='full'):
        if mode != 'full':
            raise NotImplementedError("CNOFactorized2DBlock ignores `mode` in the reference (grid_2d.py:51-96 has no switch)")
        super().__init__(modes, width, input_dim=input_dim, dropout=dropout, in_dropout=in_dropout, n_layers=n_layers,
                         share_weight=share_weight, share_fork=share_fork, factor=factor, ff_weight_norm=ff_weight_norm,
                         n_ff_layers=n_ff_layers, gain=gain, layer_norm=layer_norm, use_fork=use_fork, mode='full')
        _make_real(self, gain)

    def engine(self) -> FFNOEngine:
        if self._engine is None:
            self._engine = FFNOEngine(modes=self.modes, width=self.width, input_dim=self.input_dim, n_layers=self.n_layers,
                                      factor=self.factor, share_weight=self.share_weight, share_fork=self.share_fork,
                                      ff_weight_norm=self.ff_weight_norm, mode="full", spatial_dims=2, padding=0, output_dim=1,
                                      use_fork=self.use_fork, spectral="dct")
        return self._engine


class CNOFactorizedMesh2D(FNOFactorizedMesh2D):
    def __init__(self, modes_x, modes_y, width, input_dim, n_layers, share_weight, factor, ff_weight_norm, n_ff_layers,
                 layer_norm):
        super().__init__(modes_x, modes_y, width, input_dim, n_layers, share_weight, factor, ff_weight_norm, n_ff_layers,
                         layer_norm)
        _make_real(self)

    def engine(self) -> FFNOEngine:
        if self._engine is None:
            self._engine = FFNOEngine(modes=(self.modes_x, self.modes_y), width=self.width, input_dim=self.input_dim,
                                      n_layers=self.n_layers, factor=self.factor, share_weight=self.share_weight,
                                      share_fork=False, ff_weight_norm=self.ff_weight_norm, mode="full", spatial_dims=2,
                                      padding=self.padding, output_dim=1, first_axis_first=True, spectral="dct")
        return self._engine


class CNOFactorizedMesh3D(FNOFactorizedMesh3D):
    def __init__(self, modes_x, modes_y, modes_z, width, input_dim, output_dim, n_layers, share_weight, factor, ff_weight_norm,
                 n_ff_layers, layer_norm):
        super().__init__(modes_x, modes_y, modes_z, width, input_dim, output_dim, n_layers, share_weight, factor, ff_weight_norm,
                         n_ff_layers, layer_norm)
        _make_real(self)

    def engine(self) -> FFNOEngine:
        if self._engine is None:
            self._engine = FFNOEngine(modes=(self.modes_x, self.modes_y, self.modes_z), width=self.width,
                                      input_dim=self.input_dim, n_layers=self.n_layers, factor=self.factor,
                                      share_weight=self.share_weight, share_fork=False, ff_weight_norm=self.ff_weight_norm,
                                      mode="full", spatial_dims=3, padding=self.padding, output_dim=self.output_dim,
                                      spectral="dct")
        return self._engine
