"""FNOPlus2DBlock -- MI355X-native mirror of ``fourierflow.modules.zongyi_fno.grid_plus_2d`` (reference
grid_plus_2d.py:9-161): the F-FNO block with its factorized spectral layer replaced by the classic non-factorized
2-D spectral convolution (rfft2, two K x K corner blocks with weights [I, O, K, K, 2], irfft2) -- the
"no factorization" / "fno++" ablation of the paper.  Same constructor, parameter names / shapes and forward contract as
the reference; everything around the spectral conv (lift, feed-forward, residual, head, fork heads) is the block of
``factorized_fno.grid_2d``, and the arithmetic runs in the same engine with ``spectral='plus'``.
"""
import torch
import torch.nn as nn

from ...engine import FFNOEngine
from ..factorized_fno.grid_2d import FNOFactorized2DBlock
from ..feedforward import FeedForward


def _fourier_weights_2d(in_dim, out_dim, n_modes, gain=1.0):
    plist = nn.ParameterList([])
    for _ in range(2):
        param = nn.Parameter(torch.empty(in_dim, out_dim, n_modes, n_modes, 2))
        nn.init.xavier_normal_(param, gain=gain)      # grid_plus_2d.py:25-29, :116-120
        plist.append(param)
    return plist


class SpectralConv2d(nn.Module):
    """Parameter container of one layer (grid_plus_2d.py:9-38); the arithmetic runs in the engine."""

    def __init__(self, in_dim, out_dim, n_modes, forecast_ff, backcast_ff, fourier_weight, factor, ff_weight_norm,
                 n_ff_layers, layer_norm, use_fork, dropout, mode):
        super().__init__()
        if in_dim != out_dim:
            raise NotImplementedError("in_dim != out_dim is not used by the block")
        if mode == 'low-pass':
            raise NotImplementedError("the reference raises for mode='low-pass' (grid_plus_2d.py:74-75)")
        self.in_dim, self.out_dim, self.n_modes, self.mode, self.use_fork = in_dim, out_dim, n_modes, mode, use_fork
        self.fourier_weight = fourier_weight
        if not self.fourier_weight:
            self.fourier_weight = _fourier_weights_2d(in_dim, out_dim, n_modes)
        if use_fork:
            self.forecast_ff = forecast_ff
            if not self.forecast_ff:
                self.forecast_ff = FeedForward(out_dim, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout)
        self.backcast_ff = backcast_ff
        if not self.backcast_ff:
            self.backcast_ff = FeedForward(out_dim, factor, ff_weight_norm, n_ff_layers, layer_norm, dropout)


class FNOPlus2DBlock(FNOFactorized2DBlock):
    def __init__(self, modes, width, input_dim=12, dropout=0.0, in_dropout=0.0, n_layers=4, share_weight: bool = False,
                 share_fork=False, factor=2, ff_weight_norm=False, n_ff_layers=2, gain=1, layer_norm=False, use_fork=False,
                 mode='full'):
        # build the surrounding block without spectral layers, then attach the non-factorized ones under the same names
        super().__init__(modes, width, input_dim=input_dim, dropout=dropout, in_dropout=in_dropout, n_layers=0,
                         share_weight=False, share_fork=share_fork, factor=factor, ff_weight_norm=ff_weight_norm,
                         n_ff_layers=n_ff_layers, gain=gain, layer_norm=layer_norm, use_fork=use_fork, mode=mode)
        self.n_layers, self.share_weight = n_layers, share_weight
        out = self.out
        del self.out, self.spectral_layers            # reference registration order: ..., fourier_weight, spectral_layers, out
        self.fourier_weight = _fourier_weights_2d(width, width, modes, gain) if share_weight else None
        self.spectral_layers = nn.ModuleList([])
        for _ in range(n_layers):
            self.spectral_layers.append(SpectralConv2d(
                in_dim=width, out_dim=width, n_modes=modes, forecast_ff=self.forecast_ff, backcast_ff=self.backcast_ff,
                fourier_weight=self.fourier_weight, factor=factor, ff_weight_norm=ff_weight_norm, n_ff_layers=n_ff_layers,
                layer_norm=layer_norm, use_fork=use_fork, dropout=dropout, mode=mode))
        self.out = out

    def engine(self) -> FFNOEngine:
        if self._engine is None:
            self._engine = FFNOEngine(modes=self.modes, width=self.width, input_dim=self.input_dim, n_layers=self.n_layers,
                                      factor=self.factor, share_weight=self.share_weight, share_fork=self.share_fork,
                                      ff_weight_norm=self.ff_weight_norm, mode=self.mode, spatial_dims=2, padding=0,
                                      output_dim=1, use_fork=self.use_fork, spectral="plus")
        return self._engine
