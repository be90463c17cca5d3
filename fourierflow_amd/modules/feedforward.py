"""FeedForward -- mirror of ``fourierflow.modules.feedforward.FeedForward``
(fourierflow/modules/feedforward.py:6-24).  Same constructor and state-dict keys
(``layers.{i}.0.{weight_g,weight_v,bias}``); forward = one fused HIP kernel
(GEMM1 + bias + ReLU + GEMM2 + bias, ffno_ff_fwd).

Supported by the HIP kernel set: n_layers == 2, dropout == 0, with or without the final LayerNorm (every shipped
experiment config uses n_layers = 2, dropout = 0, layer_norm = false).  Anything else raises at construction.
"""
import torch.nn as nn

from .linear import WNLinear


class FeedForward(nn.Module):
    def __init__(self, dim, factor, ff_weight_norm, n_layers, layer_norm, dropout):
        super().__init__()
        if n_layers != 2 or dropout:
            raise NotImplementedError(
                f"FeedForward(n_layers={n_layers}, dropout={dropout}): the gfx950 kernel set implements the 2-layer, "
                f"dropout=0 feed-forward (with or without the final LayerNorm) used by every fourierflow experiment")
        self.dim, self.factor, self.wnorm, self.layer_norm = dim, factor, ff_weight_norm, bool(layer_norm)
        self.layers = nn.ModuleList([])
        for i in range(n_layers):
            in_dim = dim if i == 0 else dim * factor
            out_dim = dim if i == n_layers - 1 else dim * factor
            # indices 1..3 keep the reference's Sequential slots (Dropout / ReLU / LayerNorm) for key parity
            self.layers.append(nn.Sequential(WNLinear(in_dim, out_dim, wnorm=ff_weight_norm), nn.Identity(), nn.Identity(),
                                             nn.LayerNorm(out_dim) if layer_norm and i == n_layers - 1 else nn.Identity()))

    def forward(self, x):
        from ..ops import feedforward
        l0, l1 = self.layers[0][0], self.layers[1][0]
        y = feedforward(x, None, l0, l1)
        if self.layer_norm:
            from ..ops import layer_norm
            y = layer_norm(y, self.layers[1][3])
        return y
