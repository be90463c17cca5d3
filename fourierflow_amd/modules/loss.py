"""``LpLoss`` -- mirror of ``fourierflow.modules.loss.LpLoss`` (reference loss.py:4-46) for what the routines use: the
relative L2 loss averaged over the batch, computed (with its gradient) by one fused HIP pass
(:func:`fourierflow_amd.ops.lp_rel_loss`).  Other norms / reductions are not used by any routine and raise."""
from ..ops import lp_rel_loss


class LpLoss:
    def __init__(self, d=2, p=2, size_average=True, reduction=True):
        assert d > 0 and p > 0
        if p != 2 or not size_average or not reduction:
            raise NotImplementedError("the HIP loss kernel implements LpLoss(p=2, size_average=True, reduction=True), the only "
                                      "form the reference's routines construct")
        self.d, self.p, self.reduction, self.size_average = d, p, reduction, size_average

    def rel(self, x, y):
        n = x.shape[0]
        return lp_rel_loss(x.reshape(n, -1), y.reshape(n, -1))

    def abs(self, x, y):
        raise NotImplementedError("LpLoss.abs is not used by any routine of the reference")

    def __call__(self, x, y):
        return self.rel(x, y)
