"""FeedForward -- mirror of ``fourierflow.modules.feedforward.FeedForward``
(fourierflow/modules/feedforward.py:6-24).  Same constructor and state-dict keys
(``layers.{i}.0.{weight_g,weight_v,bias}``); forward = one fused HIP kernel
(GEMM1 + bias + ReLU + GEMM2 + bias, ffno_ff_fwd).

Supported by the HIP kernel set: n_layers == 2, dropout == 0, layer_norm == False -- the only
combination any shipped experiment config uses.  Anything else raises at construction.
"""
import torch.nn as nn

from .linear import WNLinear


class FeedForward(nn.Module):
    def __init__(self, dim, factor, ff_weight_norm, n_layers, layer_norm, dropout):
        super().__init__()
        if n_layers != 2 or layer_norm or dropout:
            raise NotImplementedError(
                f"FeedForward(n_layers={n_layers}, layer_norm={layer_norm}, dropout={dropout}): the gfx950 kernel set "
                f"implements the 2-layer, no-LayerNorm, dropout=0 feed-forward used by every fourierflow experiment")
        self.dim, self.factor, self.wnorm = dim, factor, ff_weight_norm
        self.layers = nn.ModuleList([])
        for i in range(n_layers):
            in_dim = dim if i == 0 else dim * factor
            out_dim = dim if i == n_layers - 1 else dim * factor
            # indices 1..3 keep the reference's Sequential slots (Dropout / ReLU / LayerNorm) for key parity
            self.layers.append(nn.Sequential(WNLinear(in_dim, out_dim, wnorm=ff_weight_norm), nn.Identity(),
                                             nn.Identity(), nn.Identity()))

    def forward(self, x):
        from ..ops import feedforward
        l0, l1 = self.layers[0][0], self.layers[1][0]
        return feedforward(x, None, l0, l1)
