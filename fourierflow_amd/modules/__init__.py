"""Host-side mirror of ``fourierflow.modules`` for the F-FNO hot path."""
from .factorized_cno import CNOFactorized2DBlock, CNOFactorizedMesh2D, CNOFactorizedMesh3D  # noqa: F401
from .factorized_fno import FNOFactorized2DBlock, FNOFactorizedMesh2D, FNOFactorizedMesh3D  # noqa: F401
from .feedforward import FeedForward  # noqa: F401
from .linear import WNLinear  # noqa: F401
from .normalizer import Normalizer  # noqa: F401
from .position import fourier_encode  # noqa: F401
from .zongyi_fno import FNOMesh2D, FNOMesh3D, FNOPlus2DBlock, FNOZongyi2DBlock  # noqa: F401
