"""FeedForward -- mirror of ``fourierflow.modules.feedforward.FeedForward``
(fourierflow/modules/feedforward.py:6-24).  Same constructor and state-dict keys
(``layers.{i}.0.{weight_g,weight_v,bias}``); forward = one fused HIP kernel
(GEMM1 + bias + ReLU + GEMM2 + bias, ffno_ff_fwd).

The fused kernels implement n_layers == 2, dropout == 0 (every shipped experiment config); a block that owns an engine with the
GENERAL feed-forward path (one glin kernel per linear layer, regenerated dropout masks: csrc/glin.hip) also takes n_layers >= 2 and
dropout in [0, 1) -- it says so through ``general_ok``.  Called on its own, ``forward`` covers the fused shape only.
"""
import torch.nn as nn

from .linear import WNLinear


class FeedForward(nn.Module):
    def __init__(self, dim, factor, ff_weight_norm, n_layers, layer_norm, dropout, general_ok: bool = False):
        super().__init__()
        if (n_layers != 2 or dropout) and not general_ok:
            raise NotImplementedError(
                f"FeedForward(n_layers={n_layers}, dropout={dropout}): this operator's engine only drives the fused 2-layer, "
                f"dropout=0 feed-forward (the F-FNO grid / mesh blocks take the general shapes)")
        if n_layers < 2 or not (0.0 <= float(dropout) < 1.0):
            raise NotImplementedError(f"FeedForward(n_layers={n_layers}, dropout={dropout}) is not built")
        self.n_layers, self.p_drop = n_layers, float(dropout)
        self.dim, self.factor, self.wnorm, self.layer_norm = dim, factor, ff_weight_norm, bool(layer_norm)
        self.layers = nn.ModuleList([])
        for i in range(n_layers):
            in_dim = dim if i == 0 else dim * factor
            out_dim = dim if i == n_layers - 1 else dim * factor
            # indices 1..3 keep the reference's Sequential slots (Dropout / ReLU / LayerNorm) for key parity
            self.layers.append(nn.Sequential(WNLinear(in_dim, out_dim, wnorm=ff_weight_norm), nn.Dropout(dropout), nn.Identity(),
                                             nn.LayerNorm(out_dim) if layer_norm and i == n_layers - 1 else nn.Identity()))

    def forward(self, x):
        from ..ops import feedforward
        if self.n_layers != 2 or self.p_drop:
            raise NotImplementedError("stand-alone FeedForward.forward covers n_layers = 2, dropout = 0; inside an F-FNO block "
                                      "the engine runs the general path")
        l0, l1 = self.layers[0][0], self.layers[1][0]
        y = feedforward(x, None, l0, l1)
        if self.layer_norm:
            from ..ops import layer_norm
            y = layer_norm(y, self.layers[1][3])
        return y
