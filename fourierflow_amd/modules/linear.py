"""WNLinear -- parameter container mirroring the reference's ``fourierflow.modules.WNLinear``
(fourierflow/modules/linear.py:41-52: nn.Linear + torch.nn.utils.weight_norm, dim=0).

Same constructor, same parameter names (``weight_g`` [out,1], ``weight_v`` [out,in], ``bias`` -- or
``weight``/``bias`` without weight-norm) and the same initialisation (nn.Linear's kaiming-uniform,
g = ||v|| per row), so reference checkpoints load with strict=True.  Inside a block the arithmetic
(W = g v/||v||, y = x W^T + b and its gradients) runs in the fused HIP kernels of the owner; called on
its own, ``forward`` runs the weight-norm + pointwise-linear kernels (ops.wn_linear).
"""
import math

import torch
import torch.nn as nn


class WNLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 wnorm: bool = False):
        super().__init__()
        self.in_features, self.out_features, self.wnorm = in_features, out_features, wnorm
        w = torch.empty(out_features, in_features, device=device, dtype=dtype or torch.float32)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        if not bias:
            raise NotImplementedError("bias=False is not part of the F-FNO hot path")
        bound = 1.0 / math.sqrt(in_features) if in_features > 0 else 0.0
        b = nn.Parameter(torch.empty(out_features, device=device, dtype=w.dtype).uniform_(-bound, bound))
        # registration order = the reference's parameters() order (a torch.optim state_dict indexes parameters by it):
        # nn.Linear registers weight, bias; weight_norm then removes `weight` and appends weight_g, weight_v AFTER the bias
        if wnorm:
            self.bias = b
            self.weight_g = nn.Parameter(w.norm(2, dim=1, keepdim=True))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)
            self.bias = b

    def effective_weight(self) -> torch.Tensor:
        """W = g * v / ||v||_row (plain torch; diagnostics only -- the hot path uses ffno_weightnorm_fwd)."""
        if not self.wnorm:
            return self.weight
        return self.weight_v * (self.weight_g / self.weight_v.norm(2, dim=1, keepdim=True))

    def forward(self, x):
        """Stand-alone evaluation (reference linear.py:41-52): weight-norm kernel + pointwise-linear kernel, with autograd.
        Inside the F-FNO block the same arithmetic runs in the fused lift / feed-forward / head kernels instead."""
        from ..ops import wn_linear
        return wn_linear(x, self)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, wnorm={self.wnorm}"
